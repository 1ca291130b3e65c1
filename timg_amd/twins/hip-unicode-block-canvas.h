// timg_amd/twins/hip-unicode-block-canvas.h -- GPU twin of
// timg::UnicodeBlockCanvas (src/unicode-block-canvas.h:31-80): same
// constructor, same TerminalCanvas interface, byte-identical output --
// including the frame-difference encoding of animations, whose backing store
// (the previous frame) lives on the device inside a timg_hip_block_canvas.
#ifndef TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H
#define TIMG_AMD_TWINS_HIP_UNICODE_BLOCK_CANVAS_H

#include <cstdint>
#include <vector>

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

    // Grid awareness.  MultiColumnRenderer (src/renderer.cc:81-189)
    // issues one Send per image, each at its column's x.  With columns > 1 the canvas
    // holds such Sends back -- cursor prefix consumed, frame copied, on the calling
    // thread as always -- until a grid row is complete, encodes the row with ONE device
    // call (timg_hip_block_encode_grid) and hands the buffers to the sequencer in Send
    // order: the bytes per image are those of separate Sends.  A Send at the position of
    // the previous one (an animation inside a cell, which needs the frame difference) or of
    // another size ends the row early.  0 / 1: every Send is encoded at once (default).
    void SetGridColumns(int columns);
    // Encodes and hands over what is held back (also done by the destructor).
    void Flush();

private:
    struct Pending {
        char *buffer;   // new char[]: cursor prefix in front, room for the frame behind it
        size_t prefix, cap;
        int x, dy;
        SeqType seq_type;
        Duration end_of_frame;
    };
    void SendNow(Pending p, const uint8_t *pixels, int width, int height);

    timg_hip_ctx *const ctx_;
    const int flags_;
    timg_hip_block_canvas *canvas_ = nullptr;
    int grid_columns_ = 0;
    bool have_last_x_ = false;
    int last_x_       = 0;  // x of the previous Send
    std::vector<Pending> queue_;
    std::vector<uint8_t> queued_pixels_;  // the queue's frames, back to back
    int queued_w_ = 0, queued_h_ = 0;
};

}  // namespace timg
#endif
