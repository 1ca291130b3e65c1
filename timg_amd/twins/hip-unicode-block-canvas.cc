#include "hip-unicode-block-canvas.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hip-context.h"

namespace timg {

HipUnicodeBlockCanvas::HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                                             bool use_upper_half_block, bool use_256_color)
    : TerminalCanvas(ws),
      ctx_(SharedHipContext()),
      flags_((use_quarter ? TIMG_HIP_BLOCK_QUARTER : 0) | (use_upper_half_block ? TIMG_HIP_BLOCK_UPPER : 0) |
             (use_256_color ? TIMG_HIP_BLOCK_COLOR256 : 0)) {
    if (!ctx_ || timg_hip_block_canvas_create(ctx_, flags_, &canvas_) != TIMG_HIP_OK)
        HipFatal(ctx_, "HipUnicodeBlockCanvas");
}

HipUnicodeBlockCanvas::~HipUnicodeBlockCanvas() {
    Flush();  // (before ~TerminalCanvas writes a left-over cursor prefix, src/terminal-canvas.cc:45-51)
    timg_hip_block_canvas_destroy(canvas_);
}

void HipUnicodeBlockCanvas::SetGridColumns(int columns) {
    Flush();
    grid_columns_ = columns;
}

// One Send through the stateful device canvas (which decides about the frame difference the
// way Send does, src/unicode-block-canvas.cc:343-346) and on to the sequencer.
void HipUnicodeBlockCanvas::SendNow(Pending p, const uint8_t *pixels, int width, int height) {
    size_t len = 0;
    if (timg_hip_block_canvas_send(canvas_, p.x, p.dy, pixels, width, height, 0, 0, p.buffer + p.prefix,
                                   p.cap - p.prefix, &len, nullptr) != TIMG_HIP_OK)
        HipFatal(ctx_, "timg_hip_block_canvas_send");
    // nothing emitted: the reference keeps the buffer size zero, dropping the
    // cursor jump as well (:390-395)
    write_sequencer_->WriteBuffer(OutBuffer(p.buffer, len ? p.prefix + len : 0), p.seq_type, p.end_of_frame);
}

void HipUnicodeBlockCanvas::Flush() {
    if (queue_.empty()) return;
    const size_t n = queue_.size(), frame_bytes = (size_t)queued_w_ * 4 * queued_h_;
    // all but the last frame of the row: one launch.  They are full encodes by construction
    // (every one of them follows a Send at another x).
    if (n > 1) {
        const size_t slot = timg_hip_block_max_bytes(queued_w_, queued_h_);
        std::vector<char> bytes(slot * (n - 1));
        std::vector<size_t> lens(n - 1);
        std::vector<int> xs(n - 1);
        for (size_t i = 0; i + 1 < n; ++i) xs[i] = queue_[i].x;
        if (timg_hip_block_encode_grid(ctx_, queued_pixels_.data(), queued_w_, queued_h_, 0, 0, 0, (int)(n - 1),
                                       flags_, xs.data(), bytes.data(), slot, 0, lens.data(),
                                       nullptr) != TIMG_HIP_OK)
            HipFatal(ctx_, "timg_hip_block_encode_grid");
        for (size_t i = 0; i + 1 < n; ++i) {
            Pending &p = queue_[i];
            memcpy(p.buffer + p.prefix, bytes.data() + i * slot, lens[i]);
            write_sequencer_->WriteBuffer(OutBuffer(p.buffer, lens[i] ? p.prefix + lens[i] : 0), p.seq_type,
                                          p.end_of_frame);
        }
    }
    // the last one through the stateful canvas, so that an animation that continues at its
    // place finds it as the previous frame; what the canvas remembers from before the row
    // is not the previous Send any more
    timg_hip_block_canvas_forget(canvas_);
    SendNow(queue_[n - 1], queued_pixels_.data() + (n - 1) * frame_bytes, queued_w_, queued_h_);
    queue_.clear();
    queued_pixels_.clear();
}

// src/unicode-block-canvas.cc:323-403.  What stays on the host is what has to:
// the cursor prefix (queued by the renderer on this thread) and the hand-over
// of exactly one buffer per Send to the write sequencer.
void HipUnicodeBlockCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                 Duration end_of_frame) {
    const int width = fb.width(), height = fb.height();
    // a Send at the previous position may be a frame difference: it needs its predecessor encoded
    const bool may_hold = grid_columns_ > 1 && !(have_last_x_ && x == last_x_);
    if (!queue_.empty() && (!may_hold || width != queued_w_ || height != queued_h_)) Flush();
    have_last_x_ = true;
    last_x_      = x;

    // RequestBuffers (:405-424) leaves room for the prefix in front of the frame
    Pending p;
    p.cap    = timg_hip_block_max_bytes(width, height) + 64;
    p.buffer = new char[p.cap];
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    p.prefix       = (size_t)(AppendPrefixToBuffer(p.buffer) - p.buffer);
    p.x            = x;
    p.dy           = dy;
    p.seq_type     = seq_type;
    p.end_of_frame = end_of_frame;
    if (!may_hold) {
        SendNow(p, (const uint8_t *)fb.begin(), width, height);
        return;
    }
    queued_w_ = width;
    queued_h_ = height;
    queue_.push_back(p);
    const uint8_t *pixels = (const uint8_t *)fb.begin();
    queued_pixels_.insert(queued_pixels_.end(), pixels, pixels + (size_t)width * 4 * height);
    if ((int)queue_.size() >= grid_columns_) Flush();
}

}  // namespace timg
