#include "hip-unicode-block-canvas.h"

#include <cstdio>
#include <cstdlib>

#include "hip-context.h"

namespace timg {

HipUnicodeBlockCanvas::HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                                             bool use_upper_half_block, bool use_256_color)
    : TerminalCanvas(ws), ctx_(SharedHipContext()) {
    const int flags = (use_quarter ? TIMG_HIP_BLOCK_QUARTER : 0) |
                      (use_upper_half_block ? TIMG_HIP_BLOCK_UPPER : 0) |
                      (use_256_color ? TIMG_HIP_BLOCK_COLOR256 : 0);
    if (!ctx_ || timg_hip_block_canvas_create(ctx_, flags, &canvas_) != TIMG_HIP_OK)
        HipFatal(ctx_, "HipUnicodeBlockCanvas");
}

HipUnicodeBlockCanvas::~HipUnicodeBlockCanvas() { timg_hip_block_canvas_destroy(canvas_); }

// src/unicode-block-canvas.cc:323-403.  What stays on the host is what has to:
// the cursor prefix (queued by the renderer on this thread) and the hand-over
// of exactly one buffer per Send to the write sequencer.
void HipUnicodeBlockCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                 Duration end_of_frame) {
    const int width = fb.width(), height = fb.height();
    // RequestBuffers (:405-424) leaves room for the prefix in front of the frame
    const size_t cap = timg_hip_block_max_bytes(width, height) + 64;
    OutBuffer out(new char[cap], 0);
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    char *const pos     = AppendPrefixToBuffer(out.data);
    const size_t prefix = (size_t)(pos - out.data);
    size_t len          = 0;
    if (timg_hip_block_canvas_send(canvas_, x, dy, (const uint8_t *)fb.begin(), width, height, 0, 0, pos,
                                   cap - prefix, &len, nullptr) != TIMG_HIP_OK)
        HipFatal(ctx_, "timg_hip_block_canvas_send");
    // nothing emitted: the reference keeps the buffer size zero, dropping the
    // cursor jump as well (:390-395)
    out.size = len ? prefix + len : 0;
    write_sequencer_->WriteBuffer(std::move(out), seq_type, end_of_frame);
}

}  // namespace timg
