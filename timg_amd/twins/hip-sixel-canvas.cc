#include "hip-sixel-canvas.h"

#include <cassert>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "hip-context.h"

namespace timg {

static inline int round_to_sixel(int pixels) {  // src/sixel-canvas.cc:91-94
    pixels += 5;
    return pixels - pixels % 6;
}

HipSixelCanvas::HipSixelCanvas(BufferedWriteSequencer *ws, ThreadPool *thread_pool,
                               const SixelOptions &sixel_options,
                               const DisplayOptions &display_opts)
    : TerminalCanvas(ws), options_(display_opts), full_cell_jump_(sixel_options.full_cell_jump),
      broken_cursor_(sixel_options.known_broken_cursor_placement), executor_(thread_pool),
      ctx_(SharedHipContext()) {
    if (!ctx_) HipFatal(ctx_, "HipSixelCanvas");
}

int HipSixelCanvas::cell_height_for_pixels(int pixels) const {  // src/sixel-canvas.cc:157-172
    assert(pixels <= 0);
    pixels = -pixels;
    if (full_cell_jump_) return -((round_to_sixel(pixels) - 6) / options_.cell_y_px + 1);
    return -((round_to_sixel(pixels) + options_.cell_y_px - 1) / options_.cell_y_px);
}

void HipSixelCanvas::Send(int x, int dy, const Framebuffer &fb_orig, SeqType seq_type,
                          Duration end_of_frame) {
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    MoveCursorDX(x / options_.cell_x_px);

    const int w = fb_orig.width(), h = fb_orig.height();
    // The framebuffer is only valid during this call: copy before going async.
    auto pixels = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4);
    memcpy(pixels->data(), fb_orig.begin(), pixels->size());

    // Background for the pad rows (src/sixel-canvas.cc:115-118): the getter is
    // only consulted when there are pad rows, which are fully transparent.
    timg_hip_blend pad;
    memset(&pad, 0, sizeof(pad));
    if (round_to_sixel(h) != h && options_.bgcolor_getter) {
        const rgba_t bg = options_.bgcolor_getter();
        pad.enabled     = 1;
        memcpy(&pad.bg, &bg, 4);
        memcpy(&pad.pattern, &options_.bg_pattern_color, 4);
        pad.pattern_w = options_.pattern_size * options_.cell_x_px;
        pad.pattern_h = options_.pattern_size * options_.cell_y_px / 2;
        pad.start_row = h;
    }
    const size_t cap     = 1024 + timg_hip_sixel_max_bytes(w, h) * 2;
    char *const buffer   = new char[cap];
    char *const offset   = AppendPrefixToBuffer(buffer);  // must happen on this thread
    timg_hip_ctx *ctx    = ctx_;
    const int flags      = broken_cursor_ ? TIMG_HIP_SIXEL_BROKEN_CURSOR : 0;
    const std::function<OutBuffer()> encode_fun = [=]() {
        OutBuffer out(buffer, offset - buffer);
        size_t len = 0;
        if (timg_hip_sixel_encode(ctx, pixels->data(), w, h, 0, 0, 0, 1, flags, &pad, offset,
                                  cap - (size_t)(offset - buffer), 0, &len, nullptr) != TIMG_HIP_OK)
            HipFatal(ctx, "timg_hip_sixel_encode");
        out.size += len;
        return out;
    };
    write_sequencer_->WriteBuffer(executor_->ExecAsync(encode_fun), seq_type, end_of_frame);
}

}  // namespace timg
