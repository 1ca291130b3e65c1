#include "hip-sixel-canvas.h"

#include <cassert>
#include <cstring>
#include <functional>
#include <future>
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

HipSixelCanvas::~HipSixelCanvas() { Flush(); }

void HipSixelCanvas::SetGridColumns(int columns) {
    Flush();
    grid_columns_ = columns;
}

// The held-back row: one batched encode on the encoder pool, one future per Send.
void HipSixelCanvas::Flush() {
    if (queue_.empty()) return;
    const size_t n = queue_.size();
    const std::vector<Pending> items(queue_);
    const std::shared_ptr<std::vector<uint8_t>> pixels = queued_pixels_;
    const int w = queued_w_, h = queued_h_;
    const timg_hip_blend pad = queued_pad_;
    timg_hip_ctx *ctx = ctx_;
    const int flags   = broken_cursor_ ? TIMG_HIP_SIXEL_BROKEN_CURSOR : 0;
    // frames 1.. get their buffers through promises the one task fulfils
    auto later = std::make_shared<std::vector<std::promise<OutBuffer>>>(n - 1);
    std::vector<std::future<OutBuffer>> futures;
    const std::function<OutBuffer()> encode_fun = [=]() {
        const size_t slot = timg_hip_sixel_max_bytes(w, h) * 2;
        std::vector<char> bytes(slot * n);
        std::vector<size_t> lens(n);
        if (timg_hip_sixel_encode(ctx, pixels->data(), w, h, 0, 0, 0, (int)n, flags, &pad, bytes.data(), slot, 0,
                                  lens.data(), nullptr) != TIMG_HIP_OK)
            HipFatal(ctx, "timg_hip_sixel_encode");
        size_t first_size = 0;
        for (size_t i = 0; i < n; ++i) {
            const Pending &p    = items[i];
            const size_t prefix = (size_t)(p.offset - p.buffer);
            if (prefix + lens[i] > p.cap) HipFatal(ctx, "sixel frame larger than its buffer");
            memcpy(p.offset, bytes.data() + i * slot, lens[i]);
            if (i == 0)
                first_size = prefix + lens[i];
            else
                (*later)[i - 1].set_value(OutBuffer(p.buffer, prefix + lens[i]));
        }
        return OutBuffer(items[0].buffer, first_size);
    };
    futures.push_back(executor_->ExecAsync(encode_fun));
    for (size_t i = 1; i < n; ++i) futures.push_back((*later)[i - 1].get_future());
    for (size_t i = 0; i < n; ++i)
        write_sequencer_->WriteBuffer(std::move(futures[i]), items[i].seq_type, items[i].end_of_frame);
    queue_.clear();
    queued_pixels_.reset();
}

void HipSixelCanvas::Send(int x, int dy, const Framebuffer &fb_orig, SeqType seq_type,
                          Duration end_of_frame) {
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    MoveCursorDX(x / options_.cell_x_px);

    const int w = fb_orig.width(), h = fb_orig.height();
    const bool may_hold = grid_columns_ > 1 && seq_type == SeqType::FrameImmediate && !(have_last_x_ && x == last_x_);
    if (!queue_.empty() && (!may_hold || w != queued_w_ || h != queued_h_)) Flush();
    have_last_x_ = true;
    last_x_      = x;

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
    if (may_hold) {
        if (queue_.empty()) queued_pixels_ = std::make_shared<std::vector<uint8_t>>();
        queued_w_   = w;
        queued_h_   = h;
        queued_pad_ = pad;
        const uint8_t *src = (const uint8_t *)fb_orig.begin();
        queued_pixels_->insert(queued_pixels_->end(), src, src + (size_t)w * h * 4);
        queue_.push_back(Pending{buffer, offset, cap, seq_type, end_of_frame});
        if ((int)queue_.size() >= grid_columns_) Flush();
        return;
    }
    // The framebuffer is only valid during this call: copy before going async.
    auto pixels = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4);
    memcpy(pixels->data(), fb_orig.begin(), pixels->size());
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
