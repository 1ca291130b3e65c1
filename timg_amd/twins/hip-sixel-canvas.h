// timg_amd/twins/hip-sixel-canvas.h -- GPU twin of timg::SixelCanvas
// (src/sixel-canvas.h:29-46): same constructor arguments, same TerminalCanvas
// interface.  The two libsixel calls of SixelCanvas::Send
// (src/sixel-canvas.cc:137-145) become one timg_hip_sixel_encode(), run on the
// encoder pool and handed to the sequencer as a future exactly like the
// reference does (src/sixel-canvas.cc:128-154).
#ifndef TIMG_AMD_TWINS_HIP_SIXEL_CANVAS_H
#define TIMG_AMD_TWINS_HIP_SIXEL_CANVAS_H

#include <cstdint>
#include <memory>
#include <vector>

#include "buffered-write-sequencer.h"
#include "display-options.h"
#include "term-query.h"
#include "terminal-canvas.h"
#include "thread-pool.h"
#include "timg_hip.h"

namespace timg {

class HipSixelCanvas final : public TerminalCanvas {
public:
    HipSixelCanvas(BufferedWriteSequencer *ws, ThreadPool *thread_pool,
                   const SixelOptions &sixel_options, const DisplayOptions &display_opts);

    ~HipSixelCanvas() override;

    int cell_height_for_pixels(int pixels) const final;
    void Send(int x, int dy, const Framebuffer &framebuffer, SeqType sequence_type,
              Duration end_of_frame) override;

    // Grid awareness.  The device encodes a BATCH of frames for little more
    // than one (64 frames of 800x450: 1.7 ms; one: 1.8 ms), but MultiColumnRenderer
    // (src/renderer.cc:81-189) issues one Send per image.  With columns > 1 the canvas holds
    // the still images of a grid row back -- cursor prefix consumed, frame copied, on the
    // calling thread as always -- and encodes the row with ONE timg_hip_sixel_encode call on
    // the encoder pool; the sequencer receives one future per Send, in Send order, and the
    // bytes per image are those of separate Sends.  Animation frames, a Send at the position
    // of the previous one or of another size are never held and end the row early.
    // 0 / 1: every Send is encoded on its own (default).
    void SetGridColumns(int columns);
    void Flush();

private:
    struct Pending {
        char *buffer, *offset;  // new char[]: cursor prefix in front, the frame goes to offset
        size_t cap;
        SeqType seq_type;
        Duration end_of_frame;
    };
    const DisplayOptions &options_;
    const bool full_cell_jump_;
    const bool broken_cursor_;
    ThreadPool *const executor_;
    timg_hip_ctx *const ctx_;
    int grid_columns_ = 0;
    bool have_last_x_ = false;
    int last_x_       = 0;
    std::vector<Pending> queue_;
    std::shared_ptr<std::vector<uint8_t>> queued_pixels_;  // the queue's frames, back to back
    int queued_w_ = 0, queued_h_ = 0;
    timg_hip_blend queued_pad_;
};

}  // namespace timg
#endif
