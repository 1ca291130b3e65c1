// timg_amd/twins/hip-unicode-block-canvas.h -- GPU twin of
// timg::UnicodeBlockCanvas (src/unicode-block-canvas.h:31-80): same
// constructor, same TerminalCanvas interface, byte-identical output --
// including the frame-difference encoding of animations, whose backing store
// (the previous frame) lives on the device inside a timg_hip_block_canvas.
#ifndef TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H
#define TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H

#include <cstdint>
#include <memory>
#include <vector>

#include "buffered-write-sequencer.h"
#include "cpu-sibling.h"
#include "held-rows.h"
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

    // Grid awareness.  MultiColumnRenderer (src/renderer.cc:81-189)
    // issues one Send per image, each at its column's x.  With columns > 1 the Sends of a grid
    // row are encoded by ONE device call (timg_hip_block_encode_grid; held-rows.h): every Send
    // queues a future at once, so the terminal stream is byte for byte that of separate Sends
    // -- whatever else reaches the sequencer between them (CursorOn after every image) -- and
    // the futures of a row are fulfilled together.  A Send at the position of the previous one
    // (an animation inside a cell, which needs the frame difference) is encoded on its own.
    // 0 / 1: every Send is encoded at once (default).
    void SetGridColumns(int columns);
    // Encodes what is still held (nothing has to call this: an idle row encodes itself).
    void Flush();

private:
    void SendNow(HeldFrame &p, const uint8_t *pixels, bool on_device, int width, int height, SeqType seq_type,
                 Duration end_of_frame);
    void EncodeBatch(HeldBatch &batch);
    // the device failed: the frame's bytes from the reference's own UnicodeBlockCanvas (cpu-sibling.h); returns their
    // length, written behind p's prefix (the buffer grows if it has to)
    size_t EncodeOnCpu(HeldFrame &p, const uint8_t *pixels, bool on_device, int width, int height, const char *what);

    timg_hip_ctx *const ctx_;
    const int flags_;
    timg_hip_block_canvas *canvas_ = nullptr;
    int hold_limit_   = 1;
    bool have_last_x_ = false;
    int last_x_       = 0;  // x of the previous Send
    std::unique_ptr<CpuSibling> cpu_;
    // The frame the DEVICE canvas saw last (host-resident frames only), so that a CPU sibling that takes over in the
    // middle of an animation can be shown it first and goes on emitting frame DIFFERENCES where the reference would
    // (src/unicode-block-canvas.cc:343-346) -- until round 6 the frame after the switch was a full frame: valid, other bytes.
    void RememberFrame(int x, const uint8_t *pixels, bool on_device, int width, int height);
    std::vector<uint8_t> prev_pixels_;
    int prev_x_ = 0, prev_w_ = 0, prev_h_ = 0;
    bool prev_valid_ = false, sibling_primed_ = false;
    std::unique_ptr<HeldRows> rows_;  // (last member: its thread uses the ones above)
};

}  // namespace timg
#endif
