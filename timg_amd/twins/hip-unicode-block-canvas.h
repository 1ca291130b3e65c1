// timg_amd/twins/hip-unicode-block-canvas.h -- GPU twin of
// timg::UnicodeBlockCanvas (src/unicode-block-canvas.h:31-80): same
// constructor, same TerminalCanvas interface, byte-identical output.
//
// Frames that the reference would encode completely (first frame of an image,
// every frame in --grid mode: emit_difference false,
// src/unicode-block-canvas.cc:344-346) are encoded on the GPU.  Frames of an
// animation that the reference encodes as a difference to its backing store go
// through the wrapped reference canvas, which is kept in sync by also showing
// it every GPU-encoded frame's pixels (its output for those is discarded).
#ifndef TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H
#define TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H

#include <memory>

#include "buffered-write-sequencer.h"
#include "terminal-canvas.h"
#include "timg_hip.h"
#include "unicode-block-canvas.h"

namespace timg {

class HipUnicodeBlockCanvas final : public TerminalCanvas {
public:
    HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                          bool use_upper_half_block, bool use_256_color);
    ~HipUnicodeBlockCanvas() override;

    int cell_height_for_pixels(int pixels) const final { return (pixels - 1) / 2; }
    void Send(int x, int dy, const Framebuffer &framebuffer, SeqType seq_type,
              Duration end_of_frame) override;

private:
    const bool use_quarter_blocks_;
    const bool use_upper_half_block_;
    const bool use_256_color_;
    timg_hip_ctx *const ctx_;
    int last_framebuffer_height_ = 0;
    int last_x_indent_           = 0;
    // difference frames: the reference implementation on a private sequencer
    // whose output is forwarded (or dropped when it only serves to sync state)
    struct DiffPath;
    std::unique_ptr<DiffPath> diff_;
};

}  // namespace timg
#endif
