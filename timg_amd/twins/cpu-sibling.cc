#include "cpu-sibling.h"

#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "framebuffer.h"

namespace timg {

CpuSibling::CpuSibling(std::function<TerminalCanvas *(BufferedWriteSequencer *, ThreadPool *)> make) : make_(std::move(make)) {}

CpuSibling::~CpuSibling() {
    canvas_.reset();  // (before its sequencer and pool)
    pool_.reset();
    sequencer_.reset();
    if (fd_ >= 0) close(fd_);
}

std::string CpuSibling::Encode(int x, const uint8_t *pixels, int width, int height, int dy) {
    std::lock_guard<std::mutex> l(mu_);
    if (!canvas_) {
        fd_ = memfd_create("timg-cpu-sibling", 0);
        if (fd_ < 0) {
            perror("timg: memfd_create");
            abort();
        }
        // (no frame skipping, no pacing: whatever is sent is written at once)
        sequencer_.reset(new BufferedWriteSequencer(fd_, false, 4, true, interrupt_));
        pool_.reset(new ThreadPool(1));
        canvas_.reset(make_(sequencer_.get(), pool_.get()));
    }
    Framebuffer fb(width, height);
    memcpy((void *)fb.begin(), pixels, (size_t)width * height * 4);
    // The reference allocates one scratch row behind the image and leaves it uninitialised (src/framebuffer.cc:57-62);
    // AppendDoubleRow<2> reads its first pixel for odd widths (src/unicode-block-canvas.cc:242-243).  The device path
    // defines that pixel as transparent black: the sibling continues a stream the device began, so it does too.
    memset((void *)fb.end(), 0, (size_t)width * 4);
    canvas_->Send(x, dy, fb, SeqType::FrameImmediate, Duration());
    sequencer_->Flush();
    const off_t n = lseek(fd_, 0, SEEK_END);
    std::string bytes((size_t)(n > 0 ? n : 0), '\0');
    if (n > 0 && pread(fd_, &bytes[0], (size_t)n, 0) != n) {
        perror("timg: reading the CPU canvas' bytes");
        abort();
    }
    if (ftruncate(fd_, 0) != 0 || lseek(fd_, 0, SEEK_SET) != 0) {
        perror("timg: rewinding the CPU canvas' buffer");
        abort();
    }
    return bytes;
}

}  // namespace timg
