#include "hip-graphics-canvas.h"

#include <cassert>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <vector>

#include "hip-context.h"

namespace timg {

// "ID unique enough for our purposes" (src/kitty-canvas.cc:47-52): start from the time, count up.
static uint32_t CreateKittyId() {
    static const uint32_t kStart = (uint32_t)time(nullptr) << 7;
    static uint32_t counter      = 0;
    return kStart + ++counter;
}

static int CellHeight(int pixels, int cell_y_px) {  // src/kitty-canvas.cc:236-239, src/iterm2-canvas.cc:91-95
    assert(pixels <= 0);
    return -((-pixels + cell_y_px - 1) / cell_y_px);
}

// Room for what TerminalCanvas queues in front of a frame (cursor moves, clear screen, the
// --title line: at most a terminal line of UTF-8); its length cannot be asked for.  (The
// encoded size is exact, so without it a long title would not fit.)
static constexpr size_t kPrefixBudget = 16 * 1024;

namespace {
// What both Sends share after the cursor prefix has been consumed on the calling thread: a copy
// of the frame (it is only valid during the call), the device encode on the encoder pool, one
// future for the sequencer.
template <class Encode>
void SendAsync(timg_hip_ctx *ctx, ThreadPool *pool, BufferedWriteSequencer *ws, const Framebuffer &fb,
               char *buffer, char *offset, size_t cap, SeqType seq_type, Duration end_of_frame, const char *what,
               Encode encode) {
    const int w = fb.width(), h = fb.height();
    auto pixels = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4);
    memcpy(pixels->data(), fb.begin(), pixels->size());
    const std::function<OutBuffer()> encode_fun = [=]() {
        size_t len = 0;
        if (HipCall(ctx, [&]() { return encode(pixels->data(), w, h, offset, cap - (size_t)(offset - buffer), &len); }) !=
            TIMG_HIP_OK)
            HipFatal(ctx, what);
        HipCountFrames(kHipTwinGraphics, true);
        return OutBuffer(buffer, (size_t)(offset - buffer) + len);
    };
    ws->WriteBuffer(pool->ExecAsync(encode_fun), seq_type, end_of_frame);
}
}  // namespace

// ---- kitty -----------------------------------------------------------------------------
HipKittyGraphicsCanvas::HipKittyGraphicsCanvas(BufferedWriteSequencer *ws, ThreadPool *thread_pool,
                                               bool tmux_passthrough_needed, const DisplayOptions &opts)
    : TerminalCanvas(ws), options_(opts), executor_(thread_pool), ctx_(SharedHipContext()) {
    if (!ctx_ || !Supports(tmux_passthrough_needed, opts)) HipFatal(ctx_, "HipKittyGraphicsCanvas");
}

int HipKittyGraphicsCanvas::cell_height_for_pixels(int pixels) const {
    return CellHeight(pixels, options_.cell_y_px);
}

void HipKittyGraphicsCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                  Duration end_of_frame) {
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    MoveCursorDX(x / options_.cell_x_px);
    // one id per image, two alternating ones per animation (src/kitty-canvas.cc:139-165)
    uint32_t id = 0;
    switch (seq_type) {
    case SeqType::FrameImmediate: id = CreateKittyId(); break;
    case SeqType::StartOfAnimation:
        id = CreateKittyId();
        CreateKittyId();
        animation_id_ = id;
        flip_buffer_  = 0;
        break;
    case SeqType::AnimationFrame:
        ++flip_buffer_;
        id = animation_id_ + (flip_buffer_ % 2);
        break;
    case SeqType::ControlWrite: break;
    }
    timg_hip_ctx *ctx  = ctx_;
    const int flags    = options_.local_alpha_handling ? TIMG_HIP_GFX_RGB24 : 0;
    const size_t cap   = kPrefixBudget + timg_hip_gfx_max_bytes(fb.width(), fb.height());
    char *const buffer = new char[cap];
    char *const offset = AppendPrefixToBuffer(buffer);  // must happen on this thread
    SendAsync(ctx, executor_, write_sequencer_, fb, buffer, offset, cap, seq_type, end_of_frame,
              "timg_hip_kitty_encode",
              [ctx, flags, id](const uint8_t *px, int w, int h, char *out, size_t out_cap, size_t *len) {
                  return timg_hip_kitty_encode(ctx, px, w, h, 0, 0, 0, 1, flags, &id, out, out_cap, 0, len, nullptr);
              });
}

// ---- iTerm2 ----------------------------------------------------------------------------
HipITerm2GraphicsCanvas::HipITerm2GraphicsCanvas(BufferedWriteSequencer *ws, ThreadPool *thread_pool,
                                                 const DisplayOptions &opts)
    : TerminalCanvas(ws), options_(opts), executor_(thread_pool), ctx_(SharedHipContext()) {
    if (!ctx_ || !Supports(opts)) HipFatal(ctx_, "HipITerm2GraphicsCanvas");
}

int HipITerm2GraphicsCanvas::cell_height_for_pixels(int pixels) const {
    return CellHeight(pixels, options_.cell_y_px);
}

void HipITerm2GraphicsCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                   Duration end_of_frame) {
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    MoveCursorDX(x / options_.cell_x_px);
    timg_hip_ctx *ctx  = ctx_;
    const int flags    = options_.local_alpha_handling ? TIMG_HIP_GFX_RGB24 : 0;
    const size_t cap   = kPrefixBudget + timg_hip_gfx_max_bytes(fb.width(), fb.height());
    char *const buffer = new char[cap];
    char *const offset = AppendPrefixToBuffer(buffer);  // must happen on this thread
    SendAsync(ctx, executor_, write_sequencer_, fb, buffer, offset, cap, seq_type, end_of_frame,
              "timg_hip_iterm2_encode",
              [ctx, flags](const uint8_t *px, int w, int h, char *out, size_t out_cap, size_t *len) {
                  return timg_hip_iterm2_encode(ctx, px, w, h, 0, 0, 0, 1, flags, out, out_cap, 0, len, nullptr);
              });
}

}  // namespace timg
