#include "hip-unicode-block-canvas.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <csignal>
#include <cstdlib>
#include <cstring>

#include "hip-context.h"

namespace timg {

// The reference canvas writing into a memfd, so its bytes can be re-queued on
// the real sequencer (keeping exactly one WriteBuffer per Send).
struct HipUnicodeBlockCanvas::DiffPath {
    int fd;
    volatile sig_atomic_t interrupt = 0;
    BufferedWriteSequencer seq;
    UnicodeBlockCanvas canvas;
    off_t consumed = 0;
    DiffPath(bool q, bool u, bool c)
        : fd(memfd_create("timg_hip_diff", 0)), seq(fd, false, 2, true, interrupt),
          canvas(&seq, q, u, c) {}
    ~DiffPath() { close(fd); }
    OutBuffer Take() {  // bytes produced since the last call
        seq.Flush();
        const off_t end = lseek(fd, 0, SEEK_END);
        const size_t n  = (size_t)(end - consumed);
        OutBuffer out(new char[n ? n : 1], n);
        if (n && pread(fd, out.data, n, consumed) != (ssize_t)n) out.size = 0;
        consumed = end;
        return out;
    }
};

HipUnicodeBlockCanvas::HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                                             bool use_upper_half_block, bool use_256_color)
    : TerminalCanvas(ws), use_quarter_blocks_(use_quarter),
      use_upper_half_block_(use_upper_half_block), use_256_color_(use_256_color),
      ctx_(SharedHipContext()),
      diff_(new DiffPath(use_quarter, use_upper_half_block, use_256_color)) {}

HipUnicodeBlockCanvas::~HipUnicodeBlockCanvas() {}

void HipUnicodeBlockCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                 Duration end_of_frame) {
    const int width = fb.width(), height = fb.height();
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    const int x_cells = use_quarter_blocks_ ? x / 2 : x;
    // src/unicode-block-canvas.cc:344-346
    const bool emit_difference = (x_cells == last_x_indent_) && (last_framebuffer_height_ > 0) &&
                                 abs(dy) == last_framebuffer_height_;
    last_framebuffer_height_ = height;
    last_x_indent_           = x_cells;

    // The wrapped canvas sees every frame so that its backing store is current.
    diff_->canvas.Send(x, dy, fb, SeqType::FrameImmediate, {});
    OutBuffer reference_bytes = diff_->Take();

    if (!emit_difference && ctx_ && width > 0 && height > 0) {
        const size_t cap = timg_hip_block_max_bytes(width, height) + 64;
        OutBuffer out(new char[cap], 0);
        char *pos = AppendPrefixToBuffer(out.data);
        const int flags = (use_quarter_blocks_ ? TIMG_HIP_BLOCK_QUARTER : 0) |
                          (use_upper_half_block_ ? TIMG_HIP_BLOCK_UPPER : 0) |
                          (use_256_color_ ? TIMG_HIP_BLOCK_COLOR256 : 0);
        size_t len = 0;
        const size_t room = cap - (size_t)(pos - out.data);
        if (timg_hip_block_encode(ctx_, (const uint8_t *)fb.begin(), width, height, 0, 0, 0, 1, flags,
                                  x, pos, room, 0, &len, nullptr) == TIMG_HIP_OK) {
            out.size = (size_t)(pos - out.data) + len;
            write_sequencer_->WriteBuffer(std::move(out), seq_type, end_of_frame);
            return;
        }
        // device trouble: the reference bytes below are complete but lack the
        // prefix that was just consumed -- put it in front
        const size_t pre = (size_t)(pos - out.data);
        OutBuffer joined(new char[pre + reference_bytes.size + 1], pre + reference_bytes.size);
        memcpy(joined.data, out.data, pre);
        // the wrapped canvas emitted its own cursor-up prefix for dy < 0
        memcpy(joined.data + pre, reference_bytes.data, reference_bytes.size);
        write_sequencer_->WriteBuffer(std::move(joined), seq_type, end_of_frame);
        return;
    }
    // difference frame (or no device): the reference's own bytes.  They start
    // with the wrapped canvas' cursor-up move when dy < 0, which this canvas has
    // queued as a prefix too; emit our prefix only if the reference produced
    // nothing of its own to keep a single cursor move.
    OutBuffer out(new char[reference_bytes.size + 256], 0);
    char scratch[128];
    char *end_prefix = AppendPrefixToBuffer(scratch);  // consume (and drop) our copy
    (void)end_prefix;
    memcpy(out.data, reference_bytes.data, reference_bytes.size);
    out.size = reference_bytes.size;
    write_sequencer_->WriteBuffer(std::move(out), seq_type, end_of_frame);
}

}  // namespace timg
