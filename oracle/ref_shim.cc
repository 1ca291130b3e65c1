// oracle/ref_shim.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI wrapper around the *unmodified* reference sources, compiled where
// they lie under /root/reference/src by oracle/Makefile into
// oracle/_ref/libtimg_ref.so.  Nothing from the reference is copied into this
// repository; this file only *calls* the reference's public C++ interfaces:
//
//   timg::ImageScaler::Create / Scale           (src/image-scaler.h:33-39,
//                                                src/image-scaler.cc:75-98 STB back-end)
//   timg::Framebuffer::AlphaComposeBackground   (src/framebuffer.h:103-106,
//                                                src/framebuffer.cc:108-150)
//   timg::UnicodeBlockCanvas::Send              (src/unicode-block-canvas.cc:323-403)
//   timg::TerminalCanvas prefix machinery       (src/terminal-canvas.cc:53-99)
//
// The escape bytes of a canvas are captured by pointing the reference's own
// BufferedWriteSequencer at a memfd and reading it back after the sequencer's
// destructor has flushed.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "buffered-write-sequencer.h"
#include "framebuffer.h"
#include "image-scaler.h"
#include "unicode-block-canvas.h"

using timg::Framebuffer;
using timg::rgba_t;

static rgba_t unpack(uint32_t c) {  // memory order r,g,b,a (little endian u32)
    rgba_t r;
    memcpy(&r, &c, 4);
    return r;
}

extern "C" {

// in_fmt: 0 = RGBA (ColorFmt::kRGBA), 1 = BGRA-in-memory (ColorFmt::kRGB32)
int ref_scale(const uint8_t *src, int sw, int sh, int in_fmt, uint8_t *dst,
              int dw, int dh) {
    Framebuffer in(sw, sh);
    memcpy((void *)in.begin(), src, (size_t)sw * sh * 4);
    Framebuffer out(dw, dh);
    auto scaler = timg::ImageScaler::Create(
        sw, sh,
        in_fmt == 0 ? timg::ImageScaler::ColorFmt::kRGBA
                    : timg::ImageScaler::ColorFmt::kRGB32,
        dw, dh);
    if (!scaler) return -1;
    scaler->Scale(in, &out);
    memcpy(dst, out.begin(), (size_t)dw * dh * 4);
    return 0;
}

// In-place on fb (w*h*4 bytes). has_getter==0 models "-b none" (null function).
// Returns the number of times the bg getter was invoked (laziness check).
int ref_alpha_compose(uint8_t *fb, int w, int h, int has_getter, uint32_t bg,
                      uint32_t pattern, int pw, int ph, int start_row) {
    Framebuffer f(w, h);
    memcpy((void *)f.begin(), fb, (size_t)w * h * 4);
    int calls = 0;
    Framebuffer::bgcolor_query getter;
    if (has_getter) {
        getter = [&calls, bg]() {
            ++calls;
            return unpack(bg);
        };
    }
    f.AlphaComposeBackground(getter, unpack(pattern), pw, ph, start_row);
    memcpy(fb, f.begin(), (size_t)w * h * 4);
    return calls;
}

// A persistent canvas so that multi-Send sequences (frame-diff mode) can be
// replayed.  Output of every Send accumulates in a memfd.
struct RefCanvas {
    int fd;
    volatile sig_atomic_t interrupt = 0;
    std::unique_ptr<timg::BufferedWriteSequencer> seq;
    std::unique_ptr<timg::UnicodeBlockCanvas> canvas;
};

void *ref_block_canvas_new(int quarter, int upper_block, int color256) {
    RefCanvas *c = new RefCanvas();
    c->fd        = memfd_create("timg_ref", 0);
    c->seq.reset(new timg::BufferedWriteSequencer(c->fd, false, 4, true,
                                                  c->interrupt));
    c->canvas.reset(new timg::UnicodeBlockCanvas(c->seq.get(), quarter != 0,
                                                 upper_block != 0,
                                                 color256 != 0));
    return c;
}

void ref_block_canvas_send(void *h, int x, int dy, const uint8_t *fb, int w,
                           int ht) {
    RefCanvas *c = (RefCanvas *)h;
    Framebuffer f(w, ht);
    memcpy((void *)f.begin(), fb, (size_t)w * ht * 4);
    // The reference allocates one scratch row behind the image
    // (src/framebuffer.cc:57-62) and AppendDoubleRow<2> reads one pixel into
    // it for odd widths (src/unicode-block-canvas.cc:242-243).  Its content is
    // indeterminate in the reference; pin it to transparent-black here.
    memset((void *)f.end(), 0, (size_t)w * 4);
    c->canvas->Send(x, dy, f, timg::SeqType::FrameImmediate, {});
}

// Flushes, copies everything written so far into out (up to cap); returns the
// total size written so far.
long ref_block_canvas_read(void *h, char *out, long cap) {
    RefCanvas *c = (RefCanvas *)h;
    c->seq->Flush();
    off_t size = lseek(c->fd, 0, SEEK_END);
    if (out && cap > 0) {
        long n = size < cap ? size : cap;
        if (pread(c->fd, out, n, 0) != n) return -1;
    }
    return (long)size;
}

void ref_block_canvas_free(void *h) {
    RefCanvas *c = (RefCanvas *)h;
    c->canvas.reset();
    c->seq.reset();
    close(c->fd);
    delete c;
}

// One-shot convenience: single Send(x, dy=0) on a fresh canvas.
long ref_block_encode(const uint8_t *fb, int w, int h, int quarter,
                      int upper_block, int color256, int x, char *out,
                      long cap) {
    void *c = ref_block_canvas_new(quarter, upper_block, color256);
    ref_block_canvas_send(c, x, 0, fb, w, h);
    long n = ref_block_canvas_read(c, out, cap);
    ref_block_canvas_free(c);
    return n;
}

int ref_cell_height_for_pixels(int quarter, int pixels) {
    volatile sig_atomic_t intr = 0;
    timg::BufferedWriteSequencer seq(-1, false, 1, true, intr);
    timg::UnicodeBlockCanvas c(&seq, quarter != 0, false, false);
    return c.cell_height_for_pixels(pixels);
}

uint8_t ref_as_256_term_color(uint32_t c) { return unpack(c).As256TermColor(); }

}  // extern "C"

#ifndef TIMG_REF_NO_PNG
// ---- graphics protocols: png::Encode (src/timg-png.cc + the libdeflate of this image),
// KittyGraphicsCanvas / ITerm2GraphicsCanvas (src/kitty-canvas.cc, src/iterm2-canvas.cc) ----
#include "display-options.h"
#include "iterm2-canvas.h"
#include "kitty-canvas.h"
#include "thread-pool.h"
#include "timg-png.h"

extern "C" {

long ref_png_encode(const uint8_t *fb, int w, int h, int level, int with_alpha, char *out, long cap) {
    Framebuffer f(w, h);
    memcpy((void *)f.begin(), fb, (size_t)w * h * 4);
    if ((size_t)cap < timg::png::UpperBound(w, h)) return -1;
    return (long)timg::png::Encode(f, level,
                                   with_alpha ? timg::png::ColorEncoding::kRGBA_32
                                              : timg::png::ColorEncoding::kRGB_24,
                                   out, (size_t)cap);
}

size_t ref_png_upper_bound(int w, int h) { return timg::png::UpperBound(w, h); }

// One Send(0, 0, fb) of the real canvas (kind 0: kitty, 1: iTerm2) through the real
// thread pool and write sequencer; returns everything that reached the terminal.
long ref_graphics_send(int kind, const uint8_t *fb, int w, int h, int level, int local_alpha, char *out,
                       long cap) {
    volatile sig_atomic_t interrupt = 0;
    const int fd = memfd_create("gfx", 0);
    if (fd < 0) return -1;
    {
        timg::ThreadPool pool(1);  // (outlives the sequencer: ~ThreadPool drops queued work)
        timg::BufferedWriteSequencer seq(fd, false, 4, true, interrupt);
        timg::DisplayOptions opts;
        opts.cell_x_px            = 9;
        opts.cell_y_px            = 18;
        opts.compress_pixel_level = level;
        opts.local_alpha_handling = local_alpha != 0;
        Framebuffer f(w, h);
        memcpy((void *)f.begin(), fb, (size_t)w * h * 4);
        std::unique_ptr<timg::TerminalCanvas> canvas;
        if (kind == 0)
            canvas.reset(new timg::KittyGraphicsCanvas(&seq, &pool, false, opts));
        else
            canvas.reset(new timg::ITerm2GraphicsCanvas(&seq, &pool, opts));
        canvas->Send(0, 0, f, timg::SeqType::FrameImmediate, {});
        canvas.reset();
    }
    const off_t n = lseek(fd, 0, SEEK_END);
    long got      = -1;
    if (n <= cap && pread(fd, out, (size_t)n, 0) == n) got = (long)n;
    close(fd);
    return got;
}

}  // extern "C"
#endif  // TIMG_REF_NO_PNG

#ifndef TIMG_REF_NO_SIXEL
// ---- the REAL timg::SixelCanvas (src/sixel-canvas.cc), compiled against oracle/stub/sixel.h:
// everything the class does itself -- pad rows, their background, cursor strings, prefix, the
// future on the encoder pool -- is the reference's; only the six libsixel calls land in this
// repository's restatement (parity of THOSE stays unpinned).
#include "display-options.h"
#include "sixel-canvas.h"
#include "sixel.h"
#include "term-query.h"
#include "thread-pool.h"

extern "C" {

// n_sends Sends of the same frame: the first at (x, dy = 0), the others at (x, dy = -h) -- the
// second form queues a cursor-up prefix (cell_height_for_pixels).  Returns everything that
// reached the terminal.
long ref_sixel_send(const uint8_t *fb, int w, int h, int x, int n_sends, int cell_x_px, int cell_y_px,
                    int has_getter, uint32_t bg, uint32_t pattern, int pattern_size, int broken_cursor,
                    int full_cell_jump, int lookup_mode, char *out, long cap) {
    volatile sig_atomic_t interrupt = 0;
    const int fd = memfd_create("sixel", 0);
    if (fd < 0) return -1;
    timg_stub_sixel_set_lookup_mode(lookup_mode);
    {
        timg::ThreadPool pool(2);  // (outlives the sequencer: ~ThreadPool drops queued work)
        timg::BufferedWriteSequencer seq(fd, false, 4, true, interrupt);
        timg::DisplayOptions opts;
        opts.cell_x_px        = cell_x_px;
        opts.cell_y_px        = cell_y_px;
        opts.pattern_size     = pattern_size;
        opts.bg_pattern_color = unpack(pattern);
        if (has_getter) opts.bgcolor_getter = [bg]() { return unpack(bg); };
        timg::SixelOptions so;
        so.known_broken_cursor_placement = broken_cursor != 0;
        so.full_cell_jump                = full_cell_jump != 0;
        Framebuffer f(w, h);
        memcpy((void *)f.begin(), fb, (size_t)w * h * 4);
        timg::SixelCanvas canvas(&seq, &pool, so, opts);
        for (int i = 0; i < n_sends; ++i) canvas.Send(x, i ? -h : 0, f, timg::SeqType::FrameImmediate, {});
        seq.Flush();
    }
    const off_t n = lseek(fd, 0, SEEK_END);
    long got      = -1;
    if (n <= cap && pread(fd, out, (size_t)n, 0) == n) got = (long)n;
    close(fd);
    return got;
}

int ref_sixel_cell_height(int pixels, int cell_y_px, int full_cell_jump) {
    volatile sig_atomic_t intr = 0;
    timg::BufferedWriteSequencer seq(-1, false, 1, true, intr);
    timg::DisplayOptions opts;
    opts.cell_x_px = 9;
    opts.cell_y_px = cell_y_px;
    timg::SixelOptions so;
    so.full_cell_jump = full_cell_jump != 0;
    timg::SixelCanvas c(&seq, nullptr, so, opts);
    return c.cell_height_for_pixels(pixels);
}

}  // extern "C"
#endif  // TIMG_REF_NO_SIXEL
