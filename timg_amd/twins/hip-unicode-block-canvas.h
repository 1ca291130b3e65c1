// timg_amd/twins/hip-unicode-block-canvas.h -- GPU twin of
// timg::UnicodeBlockCanvas (src/unicode-block-canvas.h:31-80): same
// constructor, same TerminalCanvas interface, byte-identical output --
// including the frame-difference encoding of animations, whose backing store
// (the previous frame) lives on the device inside a timg_hip_block_canvas.
#ifndef TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H
#define TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H

#include "buffered-write-sequencer.h"
#include "terminal-canvas.h"
#include "timg_hip.h"

namespace timg {

class HipUnicodeBlockCanvas final : public TerminalCanvas {
public:
    // Terminates the process with a message when no HIP device is usable:
    // construct a timg::UnicodeBlockCanvas instead if HipTwinsEnabled() /
    // SharedHipContext() say so (src/timg.cc:319-345).
    HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                          bool use_upper_half_block, bool use_256_color);
    ~HipUnicodeBlockCanvas() override;

    int cell_height_for_pixels(int pixels) const final { return (pixels - 1) / 2; }
    void Send(int x, int dy, const Framebuffer &framebuffer, SeqType seq_type,
              Duration end_of_frame) override;

private:
    timg_hip_ctx *const ctx_;
    timg_hip_block_canvas *canvas_ = nullptr;
};

}  // namespace timg
#endif
