// timg_amd/twins/hip-sixel-canvas.h -- GPU twin of timg::SixelCanvas
// (src/sixel-canvas.h:29-46): same constructor arguments, same TerminalCanvas
// interface.  The two libsixel calls of SixelCanvas::Send
// (src/sixel-canvas.cc:137-145) become one timg_hip_sixel_encode(), run on the
// encoder pool and handed to the sequencer as a future exactly like the
// reference does (src/sixel-canvas.cc:128-154).
#ifndef TIMG_AMD_TWINS_HIP_SIXEL_CANVAS_H
#define TIMG_AMD_TWINS_HIP_SIXEL_CANVAS_H

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

    int cell_height_for_pixels(int pixels) const final;
    void Send(int x, int dy, const Framebuffer &framebuffer, SeqType sequence_type,
              Duration end_of_frame) override;

private:
    const DisplayOptions &options_;
    const bool full_cell_jump_;
    const bool broken_cursor_;
    ThreadPool *const executor_;
    timg_hip_ctx *const ctx_;
};

}  // namespace timg
#endif
