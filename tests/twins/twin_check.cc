// tests/twins/twin_check.cc -- TEST DRIVER (not part of the product).
//
// Links the reference's own classes (compiled from /root/reference where they
// lie) next to the GPU twins and drives both through the SAME calls the
// renderer makes (ImageScaler::Create/Scale, AlphaComposeBackground,
// TerminalCanvas::Send through a BufferedWriteSequencer), then compares bytes.
// Built by tests/twins/Makefile into build/twin_check (git-ignored, travels
// to the GPU box); tests/test_twins.py runs it under -m gpu.
#include <sys/mman.h>
#include <unistd.h>

#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "buffered-write-sequencer.h"
#include "display-options.h"
#include "framebuffer.h"
#include "hip-context.h"
#include "hip-gather-writer.h"
#include "hip-graphics-canvas.h"
#include "hip-image-scaler.h"
#include "hip-raw-rgba-source.h"
#include "hip-sixel-canvas.h"
#include "hip-unicode-block-canvas.h"
#include "host-frames-source.h"
#include "timg_oracle.h"
#include "image-scaler.h"
#include "image-source.h"
#include "iterm2-canvas.h"
#include "kitty-canvas.h"
#include "renderer.h"
#include "sixel-canvas.h"
#include "sixel.h"
#include "thread-pool.h"
#include "unicode-block-canvas.h"
#include "qoi.h"

using namespace timg;

static uint32_t rng_state = 12345;
static uint32_t Rand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

static void Fill(Framebuffer *fb, int mode) {
    for (int y = 0; y < fb->height(); ++y) {
        for (int x = 0; x < fb->width(); ++x) {
            rgba_t c;
            c.r = (uint8_t)(mode == 2 ? (x * 7 + y * 3) & 0xc0 : Rand());
            c.g = (uint8_t)(mode == 2 ? (x + y * 5) & 0xc0 : Rand());
            c.b = (uint8_t)(mode == 2 ? (x * 2 + y) & 0xc0 : Rand());
            c.a = mode == 1 ? (uint8_t)Rand() : 255;
            if (mode == 1 && (Rand() & 3) == 0) c.a = (Rand() & 1) ? 0 : 255;
            fb->SetPixel(x, y, c);
        }
    }
    // The reference allocates one scratch row behind the image and leaves it
    // uninitialised (src/framebuffer.cc:57-62); AppendDoubleRow<2> reads its first
    // pixel for odd widths (src/unicode-block-canvas.cc:242-243).  Pin it to
    // transparent black -- the value the device path defines for that pixel.
    memset((void *)fb->end(), 0, (size_t)fb->width() * 4);
}

static std::string Slurp(int fd) {
    const off_t n = lseek(fd, 0, SEEK_END);
    std::string s((size_t)n, '\0');
    if (n && pread(fd, &s[0], (size_t)n, 0) != n) s.clear();
    return s;
}

static int failures = 0;
// frames the modes below KNOW the device refuses (TIMG_HIP_ERR_UNSUPP): the only ones a CPU sibling may encode
static unsigned long g_expected_cpu_frames = 0;
#define CHECK(cond, ...)                   \
    do {                                   \
        if (!(cond)) {                     \
            ++failures;                    \
            fprintf(stderr, "FAIL: ");     \
            fprintf(stderr, __VA_ARGS__);  \
            fprintf(stderr, "\n");         \
        }                                  \
    } while (0)

static void CheckScaler() {
    const int geoms[][4] = {{640, 480, 67, 50}, {320, 200, 100, 56}, {1920, 1080, 400, 225},
                            {50, 40, 120, 90}, {64, 64, 64, 64}, {3840, 2160, 800, 450}};
    for (const auto &g : geoms) {
        for (int mode = 0; mode < 2; ++mode) {
            for (auto fmt : {ImageScaler::ColorFmt::kRGBA, ImageScaler::ColorFmt::kRGB32}) {
                Framebuffer in(g[0], g[1]);
                Fill(&in, mode);
                Framebuffer want(g[2], g[3]), got(g[2], g[3]);
                auto cpu = ImageScaler::Create(g[0], g[1], fmt, g[2], g[3]);
                auto gpu = HipImageScaler::Create(g[0], g[1], fmt, g[2], g[3]);
                CHECK(cpu && gpu, "scaler creation %dx%d", g[0], g[1]);
                if (!cpu || !gpu) continue;
                cpu->Scale(in, &want);
                gpu->Scale(in, &got);
                const size_t n = (size_t)g[2] * g[3] * 4;
                CHECK(memcmp(want.begin(), got.begin(), n) == 0, "Scale %dx%d -> %dx%d mode %d", g[0],
                      g[1], g[2], g[3], mode);
                // scale + compose in one call vs the reference's two calls
                rgba_t bg, pat;
                bg.r = 30; bg.g = 30; bg.b = 46; bg.a = 255;
                pat.r = 200; pat.g = 190; pat.b = 180; pat.a = 255;
                int calls_cpu = 0, calls_gpu = 0;
                want.AlphaComposeBackground([&]() { ++calls_cpu; return bg; }, pat, 9, 9);
                static_cast<HipImageScaler *>(gpu.get())->ScaleAndCompose(
                    in, &got, [&]() { ++calls_gpu; return bg; }, pat, 9, 9);
                CHECK(memcmp(want.begin(), got.begin(), n) == 0, "ScaleAndCompose %dx%d mode %d", g[0],
                      g[1], mode);
                CHECK(calls_cpu == calls_gpu, "bg getter laziness: %d vs %d", calls_cpu, calls_gpu);
            }
        }
    }
    printf("scaler twin: checked\n");
    fflush(stdout);
}

static void CheckBlockCanvas() {
    for (int flags = 0; flags < 8; ++flags) {
        const bool quarter = flags & 1, upper = flags & 2, c256 = flags & 4;
        volatile sig_atomic_t intr = 0;
        const int fd_ref = memfd_create("ref", 0), fd_hip = memfd_create("hip", 0);
        {
            BufferedWriteSequencer seq_ref(fd_ref, false, 4, true, intr);
            BufferedWriteSequencer seq_hip(fd_hip, false, 4, true, intr);
            UnicodeBlockCanvas ref(&seq_ref, quarter, upper, c256);
            HipUnicodeBlockCanvas hip(&seq_hip, quarter, upper, c256);
            auto both = [&](int x, int dy, const Framebuffer &fb) {
                ref.Send(x, dy, fb, SeqType::FrameImmediate, {});
                hip.Send(x, dy, fb, SeqType::FrameImmediate, {});
            };
            // a grid row: three images side by side (cursor moves queued like the renderer does)
            Framebuffer a(100, 56);
            for (int col = 0; col < 3; ++col) {
                Fill(&a, col);
                if (col > 0) {
                    ref.MoveCursorDY(-ref.cell_height_for_pixels(-56));
                    hip.MoveCursorDY(-hip.cell_height_for_pixels(-56));
                }
                both(col * 102, 0, a);
            }
            // an animation: same place, small changes, one unchanged frame
            Framebuffer anim(67, 51);
            Fill(&anim, 2);
            both(4, 0, anim);
            for (int f = 0; f < 6; ++f) {
                if (f != 3) {
                    rgba_t c;
                    c.r = (uint8_t)(40 * f); c.g = 200; c.b = 10; c.a = 255;
                    for (int x = 5 * f; x < 5 * f + 9; ++x) anim.SetPixel(x, (7 * f) % 51, c);
                }
                both(4, -51, anim);
            }
        }  // sequencers flush on destruction
        const std::string r = Slurp(fd_ref), h = Slurp(fd_hip);
        CHECK(r == h && !r.empty(), "block canvas flags %d: %zu vs %zu bytes", flags, r.size(), h.size());
        close(fd_ref);
        close(fd_hip);
    }
    printf("block canvas twin: checked\n");
    fflush(stdout);
}

// The reference's OWN grid renderer (src/renderer.cc:81-189) drives both canvases; the twin
// holds a grid row's Sends back and encodes them with one device call (SetGridColumns).
static void CheckGridRenderer() {
    for (int flags : {0, 1, 5}) {
        for (int with_title = 0; with_title < 2; ++with_title) {
            const bool quarter = flags & 1, upper = flags & 2, c256 = flags & 4;
            volatile sig_atomic_t intr = 0;
            const int fd_ref = memfd_create("ref", 0), fd_hip = memfd_create("hip", 0);
            {
                BufferedWriteSequencer seq_ref(fd_ref, false, 4, true, intr);
                BufferedWriteSequencer seq_hip(fd_hip, false, 4, true, intr);
                UnicodeBlockCanvas ref(&seq_ref, quarter, upper, c256);
                HipUnicodeBlockCanvas hip(&seq_hip, quarter, upper, c256);
                const int columns = 4;
                hip.SetGridColumns(columns);
                DisplayOptions opts;
                opts.cell_x_px  = quarter ? 2 : 1;
                opts.cell_y_px  = 2;
                opts.width      = 104;  // the column's width in pixels
                opts.height     = 60;
                opts.show_title = with_title != 0;
                // (the renderers go first: their destructors still move the cursor)
                auto r_ref = Renderer::Create(&ref, opts, columns, 3, Duration(), Duration());
                auto r_hip = Renderer::Create(&hip, opts, columns, 3, Duration(), Duration());
                for (int i = 0; i < 11; ++i) {
                    // most images fill their cell; one is smaller, one is an animation
                    const int w = i == 5 ? 61 : 100, h = i == 5 ? 39 : 56;
                    Framebuffer fb(w, h);
                    Fill(&fb, i % 3);
                    const std::string title = "image " + std::to_string(i);
                    auto cb_ref = r_ref->render_cb(title);
                    auto cb_hip = r_hip->render_cb(title);
                    cb_ref(0, 0, fb, SeqType::FrameImmediate, {});
                    cb_hip(0, 0, fb, SeqType::FrameImmediate, {});
                    if (i == 2 || i == 7) {
                        for (int f = 0; f < 3; ++f) {
                            rgba_t c;
                            c.r = (uint8_t)(60 * f); c.g = 20; c.b = 220; c.a = 255;
                            for (int x = 3 * f; x < 3 * f + 11; ++x) fb.SetPixel(x, 5 + 9 * f, c);
                            cb_ref(0, -h, fb, SeqType::AnimationFrame, {});
                            cb_hip(0, -h, fb, SeqType::AnimationFrame, {});
                        }
                    }
                }
            }
            const std::string r = Slurp(fd_ref), h = Slurp(fd_hip);
            CHECK(r == h && !r.empty(), "grid renderer flags %d title %d: %zu vs %zu bytes", flags, with_title,
                  r.size(), h.size());
            close(fd_ref);
            close(fd_hip);
        }
    }
    printf("grid renderer over the block canvas twin: checked\n");
    fflush(stdout);
}

// The REAL SixelCanvas (src/sixel-canvas.cc compiled against oracle/stub/sixel.h, whose libsixel
// calls land in the oracle's restatement with the lookup the device implements) beside the twin:
// everything the class does itself -- pad rows and their background, cursor strings, prefix,
// one future per Send -- is the reference's own code.
static void CheckSixelCanvas(const char *dump_path) {
    std::string hip_stream;
    // the device's lookup (a cell answers with the entry nearest to its centre) on both sides: the stub's restatement
    // in lookup mode 1.  (libsixel's own first-hit cache is a checker in libtimg_hip_debug.so, compared with the
    // restatement's mode 0 by tests/test_gpu_parity.py: the twin has one rule.)
    timg_stub_sixel_set_lookup_mode(1);
    for (int variant = 0; variant < 4; ++variant) {
        std::string streams[2];
        for (int twin = 0; twin < 2; ++twin) {
            rng_state = 99 + variant;
            volatile sig_atomic_t intr = 0;
            const int fd = memfd_create("six", 0);
            {
                // the pool must outlive the sequencer: ~ThreadPool drops queued work, and the
                // sequencer's final flush would wait for those futures forever (timg.cc keeps the
                // same order: the encoder pool is deleted after the sequencer has been flushed)
                ThreadPool pool(2);
                BufferedWriteSequencer seq(fd, false, 4, true, intr);
                DisplayOptions opts;
                opts.cell_x_px      = 9;
                opts.cell_y_px      = 18;
                opts.pattern_size   = 1 + variant % 2;
                opts.bg_pattern_color.r = 200; opts.bg_pattern_color.g = 190; opts.bg_pattern_color.b = 180;
                opts.bg_pattern_color.a = variant >= 2 ? 255 : 0;
                if (variant != 3)
                    opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
                SixelOptions so;
                so.known_broken_cursor_placement = variant == 1;
                so.full_cell_jump                = variant == 2;
                std::unique_ptr<TerminalCanvas> canvas;
                if (twin) canvas.reset(new HipSixelCanvas(&seq, &pool, so, opts));
                else canvas.reset(new SixelCanvas(&seq, &pool, so, opts));
                const int sizes[][2] = {{200, 113}, {64, 36}, {90, 50}, {33, 7}};  // 113, 50, 7: pad rows
                int i = 0;
                for (const auto &wh : sizes) {
                    Framebuffer fb(wh[0], wh[1]);
                    Fill(&fb, 2);
                    canvas->Send(18 * i, i ? -wh[1] : 0, fb, SeqType::FrameImmediate, {});
                    if (i == 1) {  // an animation at its place
                        canvas->Send(18 * i, -wh[1], fb, SeqType::StartOfAnimation, {});
                        rgba_t c; c.r = 250; c.g = 10; c.b = 20; c.a = 255;
                        for (int x = 3; x < 30; ++x) fb.SetPixel(x, 5, c);
                        canvas->Send(18 * i, -wh[1], fb, SeqType::AnimationFrame, {});
                    }
                    ++i;
                }
                canvas.reset();
            }
            streams[twin] = Slurp(fd);
            close(fd);
        }
        CHECK(streams[0] == streams[1] && streams[0].size() > 1000 && streams[0].find("\033P") != std::string::npos,
              "sixel canvas variant %d: %zu (reference class) vs %zu bytes", variant, streams[0].size(),
              streams[1].size());
        if (variant == 0) hip_stream = streams[1];
    }
    // A frame the device REFUSES (wider than timg_hip_sixel_encode takes: TIMG_HIP_ERR_UNSUPP; the reference sizes its
    // buffer for any width, src/sixel-canvas.cc:123): that frame alone is the CPU sibling's, the back-end stays selected,
    // the frames before and after it are the device's.
    {
        std::string streams[2];
        const HipTwinCounts before = HipTwinStats();
        for (int twin = 0; twin < 2; ++twin) {
            rng_state = 4242;
            volatile sig_atomic_t intr = 0;
            const int fd = memfd_create("sixwide", 0);
            DisplayOptions opts;  // (outlives the encoder pool, as in src/timg.cc: canvases keep a reference)
            opts.cell_x_px      = 9;
            opts.cell_y_px      = 18;
            opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
            {
                ThreadPool pool(2);
                BufferedWriteSequencer seq(fd, false, 4, true, intr);
                SixelOptions so;
                std::unique_ptr<TerminalCanvas> canvas;
                if (twin) canvas.reset(new HipSixelCanvas(&seq, &pool, so, opts));
                else canvas.reset(new SixelCanvas(&seq, &pool, so, opts));
                const int sizes[][2] = {{120, 30}, {4200, 8}, {96, 24}};
                for (const auto &wh : sizes) {
                    Framebuffer fb(wh[0], wh[1]);
                    Fill(&fb, 2);
                    canvas->Send(0, 0, fb, SeqType::FrameImmediate, {});
                }
                canvas.reset();
            }
            streams[twin] = Slurp(fd);
            close(fd);
        }
        const HipTwinCounts after = HipTwinStats();
        CHECK(streams[0] == streams[1] && streams[0].size() > 1000, "a refused (4200 px wide) sixel frame between two others: %zu vs %zu bytes",
              streams[0].size(), streams[1].size());
        CHECK(!HipDegraded(), "a refused frame switched the back-end off");
        CHECK(after.cpu[kHipTwinSixel] - before.cpu[kHipTwinSixel] == 1 && after.device[kHipTwinSixel] - before.device[kHipTwinSixel] == 2,
              "refused frame: %lu on the CPU (want 1), %lu on the device (want 2)", after.cpu[kHipTwinSixel] - before.cpu[kHipTwinSixel],
              after.device[kHipTwinSixel] - before.device[kHipTwinSixel]);
        g_expected_cpu_frames += 1;
        printf("sixel canvas twin: a frame the device refuses goes to the CPU sibling alone, the back-end stays on\n");
    }
    if (dump_path) {
        FILE *f = fopen(dump_path, "wb");
        if (f) {
            fwrite(hip_stream.data(), 1, hip_stream.size(), f);
            fclose(f);
        }
    }
    printf("sixel canvas twin: identical to the reference class (%zu bytes)\n", hip_stream.size());
    fflush(stdout);
}

// The grid as src/timg.cc drives it (PresentImages, :311-396): the reference's MultiColumnRenderer,
// CursorOff before and CursorOn after every image, a last row that stays incomplete,
// sequencer->Flush() while the canvas is still alive, then renderer, canvas and the encoder pool
// destroyed in that order.  The reference canvas encodes every Send on its own; the twin holds
// grid rows back (SetGridColumns) -- the two terminal streams must not differ in a byte.
// (until round 6 the "degrade" mode ran the grid without its animation: after a device failure the block twin sent FULL
// frames where the device would have sent differences.  The CPU sibling is now shown the frame the device saw last
// (HipUnicodeBlockCanvas::RememberFrame), so the animation stays in: differences included, not a byte differs.)
static bool g_grid_animation = true;
template <class MakeCanvas>
static std::string RunGridLikeTimg(int fd, size_t queue_len, int columns, bool sixel, MakeCanvas make) {
    rng_state = 777;
    volatile sig_atomic_t intr = 0;
    {
        BufferedWriteSequencer seq(fd, false, queue_len, true, intr);
        std::unique_ptr<ThreadPool> pool(new ThreadPool(queue_len + 1));  // (src/timg.cc:321-337)
        DisplayOptions opts;
        opts.cell_x_px      = sixel ? 9 : 1;
        opts.cell_y_px      = sixel ? 18 : 2;
        opts.width          = sixel ? 126 : 104;  // the column's width in pixels
        opts.height         = sixel ? 90 : 60;
        opts.show_title     = true;
        opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
        std::unique_ptr<TerminalCanvas> canvas(make(&seq, pool.get(), opts));
        {
            auto renderer = Renderer::Create(canvas.get(), opts, columns, 3, Duration(), Duration());
            const int n_images = 2 * columns + columns / 2 + 1;  // the last row stays incomplete
            for (int i = 0; i < n_images; ++i) {
                const bool small = i == columns + 1;
                const int w = sixel ? (small ? 90 : 120) : (small ? 61 : 100);
                const int h = sixel ? (small ? 50 : 85) : (small ? 39 : 56);
                Framebuffer fb(w, h);
                Fill(&fb, i % 3);
                canvas->CursorOff();  // before_image_show
                auto cb = renderer->render_cb("image " + std::to_string(i));
                const bool animation = g_grid_animation && i == 2;
                cb(0, 0, fb, animation ? SeqType::StartOfAnimation : SeqType::FrameImmediate, {});
                if (animation) {
                    for (int f = 0; f < 2; ++f) {
                        rgba_t c;
                        c.r = 250; c.g = (uint8_t)(90 * f); c.b = 20; c.a = 255;
                        for (int x = 10; x < 40; ++x) fb.SetPixel(x, 20 + 11 * f, c);
                        cb(0, -h, fb, SeqType::AnimationFrame, {});
                    }
                }
                canvas->CursorOn();  // after_image_show
            }
            seq.Flush();  // src/timg.cc:392 -- the canvas is still alive
        }                 // renderer
        canvas.reset();   // canvas
        pool.reset();     // compression_pool: its destructor drops work that is still queued
    }                     // (the sequencer belongs to main(): it outlives all of them)
    return Slurp(fd);
}

static void CheckGridLikeTimg() {
    timg_stub_sixel_set_lookup_mode(1);
    for (int kind = 0; kind < 2; ++kind) {      // 0: block canvases, 1: sixel canvases
        // 4: src/timg.cc:972; 9: a whole row of four; 129: what a whole 8x8 grid would need -- with rows of four (the
        // queue is never the bound) and with rows of twenty (longer than the twins' batch cap: a row is cut into
        // batches, HipSixelCanvas BatchCap)
        const struct { size_t queue_len; int columns; } kCases[] = {{4, 4}, {9, 4}, {2, 4}, {129, 4}, {129, 20}};
        for (const auto &kc : kCases) {
            const size_t queue_len = kc.queue_len;
            const int columns      = kc.columns;
            std::string streams[2];
            for (int twin = 0; twin < 2; ++twin) {
                const int fd = memfd_create("grid", 0);
                streams[twin] = RunGridLikeTimg(
                    fd, queue_len, columns, kind == 1,
                    [&](BufferedWriteSequencer *seq, ThreadPool *pool, const DisplayOptions &opts) -> TerminalCanvas * {
                        static SixelOptions so;
                        // (as the patched PresentImages does, integration/timg-hip.patch: no usable device -- or a back-end
                        // that has been switched off after a failure -- means the reference's class)
                        const bool use_twin = twin && SharedHipContext() != nullptr;
                        if (kind == 0 && !use_twin) return new UnicodeBlockCanvas(seq, true, false, false);
                        if (kind == 1 && !use_twin) return new SixelCanvas(seq, pool, so, opts);
                        if (kind == 0) {
                            auto *c = new HipUnicodeBlockCanvas(seq, true, false, false);
                            c->SetGridColumns(columns);
                            return c;
                        }
                        auto *c = new HipSixelCanvas(seq, pool, so, opts);
                        c->SetGridColumns(columns);
                        return c;
                    });
                close(fd);
            }
            CHECK(streams[0] == streams[1] && streams[0].size() > 10000,
                  "grid like timg.cc, %s canvases, queue %zu, %d columns: %zu (reference) vs %zu bytes", kind ? "sixel" : "block",
                  queue_len, columns, streams[0].size(), streams[1].size());
            // the cursor is switched on again at the very end
            CHECK(streams[1].size() > 6 && streams[1].rfind("\033[?25h") != std::string::npos &&
                      streams[1].rfind("\033[?25h") > streams[1].rfind("\033[?25l"),
                  "cursor left off");
        }
    }
    printf("grid as src/timg.cc drives it (cursor writes between Sends, partial last row, Flush before the canvas "
           "goes): twins identical to the reference canvases\n");
    fflush(stdout);
}

// kitty / iTerm2 at --compress=0: the reference canvases (real png::Encode + libdeflate) beside
// the twins, through the same thread pool / sequencer machinery.  kitty image ids come from
// time() in both (src/kitty-canvas.cc:47-52): they are blanked out before comparing.
static std::string BlankKittyIds(std::string s) {
    size_t at = 0;
    while ((at = s.find("a=T,i=", at)) != std::string::npos) {
        at += 6;
        while (at < s.size() && s[at] >= '0' && s[at] <= '9') s[at++] = '#';
    }
    return s;
}

static void CheckGraphicsCanvases() {
    for (int kind = 0; kind < 2; ++kind) {          // 0 kitty, 1 iTerm2
        for (int local_alpha = 0; local_alpha < 2; ++local_alpha) {
            std::string streams[2];
            for (int twin = 0; twin < 2; ++twin) {
                rng_state = 4242;
                volatile sig_atomic_t intr = 0;
                const int fd = memfd_create("gfx", 0);
                {
                    ThreadPool pool(2);  // (outlives the sequencer, see above)
                    BufferedWriteSequencer seq(fd, false, 4, true, intr);
                    DisplayOptions opts;
                    opts.cell_x_px            = 9;
                    opts.cell_y_px            = 18;
                    opts.compress_pixel_level = 0;
                    opts.local_alpha_handling = local_alpha != 0;
                    std::unique_ptr<TerminalCanvas> canvas;
                    if (kind == 0 && twin) canvas.reset(new HipKittyGraphicsCanvas(&seq, &pool, false, opts));
                    if (kind == 0 && !twin) canvas.reset(new KittyGraphicsCanvas(&seq, &pool, false, opts));
                    if (kind == 1 && twin) canvas.reset(new HipITerm2GraphicsCanvas(&seq, &pool, opts));
                    if (kind == 1 && !twin) canvas.reset(new ITerm2GraphicsCanvas(&seq, &pool, opts));
                    const int sizes[][2] = {{67, 50}, {200, 113}, {400, 300}, {5, 3}};
                    int col = 0;
                    for (const auto &wh : sizes) {
                        Framebuffer fb(wh[0], wh[1]);
                        Fill(&fb, col % 2);
                        canvas->Send(18 * col, col ? -wh[1] : 0, fb, SeqType::FrameImmediate, {});
                        ++col;
                    }
                    // an animation: start + two frames (kitty alternates two ids)
                    Framebuffer anim(90, 40);
                    Fill(&anim, 1);
                    canvas->Send(0, 0, anim, SeqType::StartOfAnimation, {});
                    for (int f = 0; f < 2; ++f) {
                        rgba_t c;
                        c.r = 200; c.g = (uint8_t)(100 * f); c.b = 0; c.a = 255;
                        for (int x = 0; x < 30; ++x) anim.SetPixel(x, 3 + f, c);
                        canvas->Send(0, -40, anim, SeqType::AnimationFrame, {});
                    }
                    canvas.reset();
                }
                streams[twin] = Slurp(fd);
                close(fd);
            }
            const std::string r = BlankKittyIds(streams[0]), h = BlankKittyIds(streams[1]);
            CHECK(r == h && r.size() > 1000, "%s canvas, local alpha %d: %zu vs %zu bytes", kind ? "iTerm2" : "kitty",
                  local_alpha, r.size(), h.size());
        }
    }
    printf("kitty / iTerm2 canvas twins at --compress=0: checked\n");
    fflush(stdout);
}

// The device-resident ImageSource against the reference's own loader: the same
// pixels as a .qoi file through ImageSource::Create -> QOIImageSource (memcpy, STB scaler,
// AlphaComposeBackground on the host; src/qoi-image-source.cc:42-77) and as a .rgba file through
// HipRawRGBASource (upload, scale + compose on the device, framebuffer handed over in device
// memory), both through the reference's Renderer into a canvas.
static void CheckImageSource() {
    timg_stub_sixel_set_lookup_mode(1);
    char dir_template[] = "/tmp/twin_src_XXXXXX";
    const char *dir = mkdtemp(dir_template);
    CHECK(dir != nullptr, "mkdtemp");
    if (!dir) return;
    int n = 0;
    for (int mode = 0; mode < 3; ++mode) {
        const int sw = mode == 2 ? 333 : 640, sh = mode == 2 ? 250 : 480;
        Framebuffer src(sw, sh);
        rng_state = 31 + mode;
        Fill(&src, mode);
        const std::string qoi_name = std::string(dir) + "/f" + std::to_string(mode) + ".qoi";
        const std::string raw_name = std::string(dir) + "/f" + std::to_string(mode) + ".rgba";
        qoi_desc desc;
        desc.width = sw; desc.height = sh; desc.channels = 4; desc.colorspace = QOI_SRGB;
        CHECK(qoi_write(qoi_name.c_str(), src.begin(), &desc) > 0, "qoi_write");
        {
            FILE *f = fopen(raw_name.c_str(), "wb");
            const uint32_t dims[2] = {(uint32_t)sw, (uint32_t)sh};
            fwrite("TIMGRGBA", 1, 8, f);
            fwrite(dims, 4, 2, f);
            fwrite(src.begin(), 4, (size_t)sw * sh, f);
            fclose(f);
        }
        for (int canvas_kind = 0; canvas_kind < 2; ++canvas_kind) {  // 0: quarter blocks, 1: sixel
            // 0: reference loader + reference canvas; 1: device source + Hip canvas (pixels stay on the
            // device); 2: device source + reference canvas (pixels copied back for it)
            std::string streams[3];
            for (int path = 0; path < 3; ++path) {
                volatile sig_atomic_t intr = 0;
                const int fd = memfd_create("src", 0);
                {
                    BufferedWriteSequencer seq(fd, false, 4, true, intr);
                    ThreadPool pool(2);
                    DisplayOptions opts;
                    opts.cell_x_px        = canvas_kind ? 9 : 2;
                    opts.cell_y_px        = canvas_kind ? 18 : 2;
                    opts.width            = canvas_kind ? 200 : 160;
                    opts.height           = canvas_kind ? 126 : 90;
                    opts.width_stretch    = canvas_kind ? 1.0f : 2.0f;
                    opts.pattern_size     = 2;
                    opts.bg_pattern_color.r = 200; opts.bg_pattern_color.g = 190; opts.bg_pattern_color.b = 180;
                    opts.bg_pattern_color.a = 255;
                    opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
                    static SixelOptions so;
                    std::unique_ptr<TerminalCanvas> canvas;
                    const bool hip_canvas = path == 1;
                    if (canvas_kind == 0 && hip_canvas) canvas.reset(new HipUnicodeBlockCanvas(&seq, true, false, false));
                    if (canvas_kind == 0 && !hip_canvas) canvas.reset(new UnicodeBlockCanvas(&seq, true, false, false));
                    if (canvas_kind == 1 && hip_canvas) canvas.reset(new HipSixelCanvas(&seq, &pool, so, opts));
                    if (canvas_kind == 1 && !hip_canvas) canvas.reset(new SixelCanvas(&seq, &pool, so, opts));
                    {
                        auto renderer = Renderer::Create(canvas.get(), opts, 1, 1, Duration(), Duration());
                        std::string error;
                        std::unique_ptr<ImageSource> source(
                            path == 0 ? ImageSource::Create(qoi_name, opts, 0, 1, true, false, &error)
                                      : HipRawRGBASource::TryCreate(raw_name, opts, 0, 1));
                        CHECK(source != nullptr, "image source %d for %s: %s", path, raw_name.c_str(), error.c_str());
                        if (source) source->SendFrames(Duration(), 1, intr, renderer->render_cb(""));
                        seq.Flush();
                    }
                    canvas.reset();
                }
                streams[path] = Slurp(fd);
                close(fd);
            }
            CHECK(streams[0] == streams[1] && streams[0].size() > 1000,
                  "device-resident source + Hip canvas, mode %d canvas %d: %zu (reference) vs %zu bytes", mode,
                  canvas_kind, streams[0].size(), streams[1].size());
            CHECK(streams[0] == streams[2], "device-resident source + reference canvas, mode %d canvas %d: %zu vs %zu bytes",
                  mode, canvas_kind, streams[0].size(), streams[2].size());
            ++n;
        }
        unlink(qoi_name.c_str());
        unlink(raw_name.c_str());
    }
    {   // the measurement plan's frames by name: generated on the device, never on the host
        volatile sig_atomic_t intr = 0;
        const int fd = memfd_create("src", 0);
        {
            BufferedWriteSequencer seq(fd, false, 4, true, intr);
            DisplayOptions opts;
            opts.cell_x_px = 2; opts.cell_y_px = 2; opts.width = 160; opts.height = 90; opts.width_stretch = 2.0f;
            HipUnicodeBlockCanvas canvas(&seq, true, false, false);
            auto renderer = Renderer::Create(&canvas, opts, 1, 1, Duration(), Duration());
            std::unique_ptr<ImageSource> source(HipRawRGBASource::TryCreate("synth:photo:1920x1080:5:2", opts, 0, 1));
            CHECK(source != nullptr, "synth: source");
            if (source) source->SendFrames(Duration(), 1, intr, renderer->render_cb(""));
            CHECK(HipRawRGBASource::TryCreate("/no/such/file.png", opts, 0, 1) == nullptr, "foreign names are refused");
            seq.Flush();
        }
        const std::string s = Slurp(fd);
        CHECK(s.size() > 1000, "synth: source wrote %zu bytes", s.size());
        close(fd);
    }
    rmdir(dir);
    printf("device-resident image source: %d pipelines identical to QOIImageSource + reference scaler + reference canvas\n", n);
    fflush(stdout);
}


// One pipeline: source -> reference Renderer -> canvas -> sequencer on a memfd; returns the terminal stream.
// canvas_kind 0: quarter blocks, 1: sixel.  hip_canvas: the twin or the reference class.
template <class MakeSource>
static std::string RunSourcePipeline(int canvas_kind, bool hip_canvas, DisplayOptions opts, int loops, MakeSource make) {
    volatile sig_atomic_t intr = 0;
    const int fd = memfd_create("src", 0);
    {
        BufferedWriteSequencer seq(fd, false, 4, true, intr);
        ThreadPool pool(2);
        opts.cell_x_px        = canvas_kind ? 9 : 2;
        opts.cell_y_px        = canvas_kind ? 18 : 2;
        opts.width            = canvas_kind ? 200 : 160;
        opts.height           = canvas_kind ? 126 : 90;
        opts.width_stretch    = canvas_kind ? 1.0f : 2.0f;
        opts.pattern_size     = 2;
        opts.bg_pattern_color.r = 200; opts.bg_pattern_color.g = 190; opts.bg_pattern_color.b = 180;
        opts.bg_pattern_color.a = 255;
        opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
        static SixelOptions so;
        std::unique_ptr<TerminalCanvas> canvas;
        if (canvas_kind == 0 && hip_canvas) canvas.reset(new HipUnicodeBlockCanvas(&seq, true, false, false));
        if (canvas_kind == 0 && !hip_canvas) canvas.reset(new UnicodeBlockCanvas(&seq, true, false, false));
        if (canvas_kind == 1 && hip_canvas) canvas.reset(new HipSixelCanvas(&seq, &pool, so, opts));
        if (canvas_kind == 1 && !hip_canvas) canvas.reset(new SixelCanvas(&seq, &pool, so, opts));
        {
            auto renderer = Renderer::Create(canvas.get(), opts, 1, 1, Duration(), Duration());
            std::unique_ptr<ImageSource> source(make(opts));
            CHECK(source != nullptr, "pipeline source");
            if (source) source->SendFrames(Duration::InfiniteFuture(), loops, intr, renderer->render_cb(""));
            seq.Flush();
        }
        canvas.reset();
    }
    std::string s = Slurp(fd);
    close(fd);
    return s;
}

// A multi-frame stream (BASELINE config 4's shape: frames of one source, StartOfAnimation / AnimationFrame,
// dy = -height) through HipRawRGBASource -- frame_offset, frame_count and loops honoured as the reference's
// loaders do -- against the same frames through the reference's scaler + compose + canvases.
static void CheckAnimationSource() {
    timg_stub_sixel_set_lookup_mode(1);
    char dir_template[] = "/tmp/twin_anim_XXXXXX";
    const char *dir = mkdtemp(dir_template);
    CHECK(dir != nullptr, "mkdtemp");
    if (!dir) return;
    const int sw = 320, sh = 200, nf = 5;
    std::vector<uint8_t> frames((size_t)nf * sw * sh * 4);
    for (int f = 0; f < nf; ++f) {
        Framebuffer fb(sw, sh);
        rng_state = 900 + f;
        Fill(&fb, f == 3 ? 1 : 2);  // (one frame with alpha: the compose runs for the stream)
        if (f > 0 && f != 3) {      // consecutive frames share most pixels: the block canvas' frame-diff mode has work
            memcpy((void *)fb.begin(), frames.data(), (size_t)sw * sh * 4);
            rgba_t c; c.r = 250; c.g = (uint8_t)(60 * f); c.b = 20; c.a = 255;
            for (int y = 20 * f; y < 20 * f + 15; ++y)
                for (int x = 30; x < 200; ++x) fb.SetPixel(x, y, c);
        }
        memcpy(frames.data() + (size_t)f * sw * sh * 4, fb.begin(), (size_t)sw * sh * 4);
    }
    const std::string raw_name = std::string(dir) + "/anim.rgba";
    {
        FILE *f = fopen(raw_name.c_str(), "wb");
        const uint32_t dims[2] = {(uint32_t)sw, (uint32_t)sh};
        fwrite("TIMGRGBA", 1, 8, f);
        fwrite(dims, 4, 2, f);
        fwrite(frames.data(), 1, frames.size(), f);
        fclose(f);
    }
    int n = 0;
    const int cases[][3] = {{0, -1, 1}, {1, 3, 2}, {4, 5, 1}, {0, 1, 3}};  // frame_offset, frame_count, loops
    for (const auto &c : cases) {
        for (int canvas_kind = 0; canvas_kind < 2; ++canvas_kind) {
            DisplayOptions opts;
            const std::string want = RunSourcePipeline(canvas_kind, false, opts, c[2], [&](const DisplayOptions &o) -> ImageSource * {
                auto *s = new HostFramesSource("anim", frames.data(), nf, sw, sh);
                if (!s->LoadAndScale(o, c[0], c[1])) { delete s; return nullptr; }
                return s;
            });
            for (int hip_canvas = 1; hip_canvas >= 0; --hip_canvas) {
                const std::string got = RunSourcePipeline(canvas_kind, hip_canvas != 0, opts, c[2], [&](const DisplayOptions &o) {
                    return HipRawRGBASource::TryCreate(raw_name, o, c[0], c[1]);
                });
                CHECK(got == want && want.size() > 1000,
                      "animation offset %d count %d loops %d canvas %d (hip canvas %d): %zu (reference) vs %zu bytes", c[0],
                      c[1], c[2], canvas_kind, hip_canvas, want.size(), got.size());
                ++n;
            }
        }
    }
    unlink(raw_name.c_str());
    rmdir(dir);
    // the generator's streams: frame f of "synth:...:<first>:<count>" is frame <first> + f of the hash
    {
        DisplayOptions opts;
        const std::string a = RunSourcePipeline(0, true, opts, 1, [&](const DisplayOptions &o) {
            return HipRawRGBASource::TryCreate("synth:photo:640x360:9:2:3", o, 1, 1);
        });
        const std::string b = RunSourcePipeline(0, true, opts, 1, [&](const DisplayOptions &o) {
            return HipRawRGBASource::TryCreate("synth:photo:640x360:9:3", o, 0, -1);
        });
        CHECK(a == b && a.size() > 1000, "synth stream frame 1 of (first 2, count 3) == single frame 3: %zu vs %zu", a.size(), b.size());
    }
    printf("multi-frame device-resident source: %d streams identical to the reference's scaler + canvases\n", n);
    fflush(stdout);
}

// --crop-border / --auto-crop wired into the device-resident source (src/graphics-magick-source.cc:231-241:
// crop, then trim, before scaling; still images only) against the same window cut on the host (the oracle's
// bounding box -- parity unpinned against GraphicsMagick) through the reference's scaler.
static void CheckAutoCropSource() {
    timg_stub_sixel_set_lookup_mode(1);
    char dir_template[] = "/tmp/twin_crop_XXXXXX";
    const char *dir = mkdtemp(dir_template);
    CHECK(dir != nullptr, "mkdtemp");
    if (!dir) return;
    const int sw = 400, sh = 300;
    Framebuffer src(sw, sh);
    rng_state = 4242;
    Fill(&src, 0);
    rgba_t border; border.r = 12; border.g = 200; border.b = 90; border.a = 255;
    for (int y = 0; y < sh; ++y)
        for (int x = 0; x < sw; ++x)
            if (x < 31 || x >= sw - 22 || y < 17 || y >= sh - 40) src.SetPixel(x, y, border);
    const std::string raw_name = std::string(dir) + "/crop.rgba";
    {
        FILE *f = fopen(raw_name.c_str(), "wb");
        const uint32_t dims[2] = {(uint32_t)sw, (uint32_t)sh};
        fwrite("TIMGRGBA", 1, 8, f);
        fwrite(dims, 4, 2, f);
        fwrite(src.begin(), 4, (size_t)sw * sh, f);
        fclose(f);
    }
    int n = 0;
    const int cases[][2] = {{1, 0}, {1, 5}, {0, 9}, {1, 40}};  // auto_crop, crop_border
    for (const auto &c : cases) {
        int box[4] = {0, 0, sw, sh};
        if (c[0]) {
            oracle_autocrop_bbox((const uint8_t *)src.begin(), sw, sh, sw * 4, c[1], box);
        } else {
            box[0] = box[1] = c[1];
            box[2] = sw - 2 * c[1];
            box[3] = sh - 2 * c[1];
        }
        if (c[0] && c[1] == 0) CHECK(box[0] == 31 && box[1] == 17 && box[2] == sw - 53 && box[3] == sh - 57, "bbox %d %d %d %d", box[0], box[1], box[2], box[3]);
        for (int canvas_kind = 0; canvas_kind < 2; ++canvas_kind) {
            DisplayOptions opts;
            opts.auto_crop   = c[0] != 0;
            opts.crop_border = c[1];
            const std::string want = RunSourcePipeline(canvas_kind, false, opts, 1, [&](const DisplayOptions &o) -> ImageSource * {
                auto *s = new HostFramesSource("crop", (const uint8_t *)src.begin(), 1, sw, sh);
                s->SetCrop(box[0], box[1], box[2], box[3]);
                if (!s->LoadAndScale(o, 0, 1)) { delete s; return nullptr; }
                return s;
            });
            const std::string got = RunSourcePipeline(canvas_kind, true, opts, 1, [&](const DisplayOptions &o) {
                return HipRawRGBASource::TryCreate(raw_name, o, 0, 1);
            });
            CHECK(got == want && want.size() > 1000, "auto_crop %d crop_border %d canvas %d: %zu (host window) vs %zu bytes", c[0], c[1],
                  canvas_kind, want.size(), got.size());
            ++n;
        }
    }
    unlink(raw_name.c_str());
    rmdir(dir);
    printf("crop-border / auto-crop in the device-resident source: %d pipelines identical to the window cut on the host\n", n);
    fflush(stdout);
}

// TIMG_HIP_FILTER=bilinear (or a twin build with WITH_TIMG_SWS_RESIZE): HipImageScaler::Create asks the device for the
// triangle filter -- what a stock timg build (libswscale SWS_BILINEAR, src/image-scaler.cc:45-72) scales with.  libswscale
// is not in the tree (parity unpinned): the checker is the oracle's triangle resampler, itself within 0.5 LSB of an
// independent float64 one (tests/test_triangle_reference.py).  Run as its own process: the choice is made once.
static void CheckBilinearScaler() {
    const int geoms[][4] = {{640, 480, 67, 50}, {320, 240, 500, 300}, {1920, 1080, 400, 225}};
    for (const auto &g : geoms) {
        Framebuffer in(g[0], g[1]);
        Fill(&in, 1);
        Framebuffer got(g[2], g[3]);
        std::vector<uint8_t> want((size_t)g[2] * g[3] * 4);
        auto gpu = HipImageScaler::Create(g[0], g[1], ImageScaler::ColorFmt::kRGBA, g[2], g[3]);
        CHECK(gpu != nullptr, "bilinear scaler creation");
        if (!gpu) continue;
        gpu->Scale(in, &got);
        CHECK(oracle_scale((const uint8_t *)in.begin(), g[0], g[1], 0, want.data(), g[2], g[3], 2) == 0, "oracle_scale");
        CHECK(memcmp(want.data(), got.begin(), want.size()) == 0, "bilinear twin %dx%d -> %dx%d", g[0], g[1], g[2], g[3]);
        // ... and it is NOT the stb filter (the switch did something)
        std::vector<uint8_t> stb((size_t)g[2] * g[3] * 4);
        oracle_scale((const uint8_t *)in.begin(), g[0], g[1], 0, stb.data(), g[2], g[3], 0);
        CHECK(memcmp(stb.data(), got.begin(), stb.size()) != 0, "bilinear twin equals the stb filter");
    }
    printf("bilinear scaler twin (TIMG_HIP_FILTER=bilinear): checked against the triangle resampler\n");
    fflush(stdout);
}

// The multi-GPU exchange step through its C-ABI (include/timg_hip_comm.h) and its C++ caller: the
// frames a rank encoded, gathered over RCCL and handed to the reference's sequencer in frame order.
// One GPU is all this box has: world = 1 (the gather to oneself runs the same calls; the frame-order
// arithmetic for several ranks is checked on the CPU in tests/test_abi.py).
static void CheckGatherWriter() {
    timg_hip_ctx *ctx = SharedHipContext();
    uint8_t id[TIMG_HIP_COMM_ID_BYTES];
    timg_hip_comm *comm = nullptr;
    CHECK(timg_hip_comm_unique_id(id) == 0, "unique id: %s", timg_hip_comm_last_error(nullptr));
    CHECK(timg_hip_comm_create(0, 1, 0, id, &comm) == 0, "comm create: %s", timg_hip_comm_last_error(nullptr));
    if (!comm) return;
    const int n = 7, w = 100, h = 56;
    std::vector<uint8_t> frames((size_t)n * w * h * 4);
    rng_state = 4711;
    for (int i = 0; i < n; ++i) {
        Framebuffer fb(w, h);
        Fill(&fb, i % 3);
        memcpy(&frames[(size_t)i * w * h * 4], fb.begin(), (size_t)w * h * 4);
    }
    // encode on the device, leave the bytes there, pack them back to back
    const size_t slot = timg_hip_block_max_bytes(w, h);
    uint8_t *dev_out = nullptr, *packed = nullptr;
    CHECK(timg_hip_malloc(ctx, slot * n, (void **)&dev_out) == TIMG_HIP_OK &&
              timg_hip_malloc(ctx, slot * n, (void **)&packed) == TIMG_HIP_OK, "device buffers");
    std::vector<size_t> lens(n);
    std::vector<int> xs(n, 0);
    CHECK(timg_hip_block_encode_grid(ctx, frames.data(), w, h, 0, 0, 0, n, TIMG_HIP_BLOCK_QUARTER, xs.data(),
                                     (char *)dev_out, slot, 1, lens.data(), nullptr) == TIMG_HIP_OK, "encode: %s",
          timg_hip_last_error(ctx));
    std::vector<uint64_t> lens64(n);
    size_t at = 0;
    for (int i = 0; i < n; ++i) {
        timg_hip_memcpy_d2d(ctx, packed + at, dev_out + (size_t)i * slot, lens[i], nullptr);
        lens64[i] = lens[i];
        at += lens[i];
    }
    timg_hip_sync(ctx, nullptr);
    // what must arrive: the same frames encoded to host memory, one after the other
    std::vector<char> host(slot * n);
    std::vector<size_t> hl(n);
    timg_hip_block_encode_grid(ctx, frames.data(), w, h, 0, 0, 0, n, TIMG_HIP_BLOCK_QUARTER, xs.data(), host.data(), slot, 0,
                               hl.data(), nullptr);
    std::string want;
    for (int i = 0; i < n; ++i) want.append(host.data() + (size_t)i * slot, hl[i]);
    for (int round_robin = 0; round_robin < 2; ++round_robin) {
        volatile sig_atomic_t intr = 0;
        const int fd = memfd_create("gather", 0);
        {
            BufferedWriteSequencer seq(fd, false, 4, true, intr);
            HipGatherWriter writer(ctx, comm, 1, 0, &seq);
            CHECK(writer.GatherAndWrite(packed, lens64.data(), n, n, round_robin != 0), "GatherAndWrite");
            seq.Flush();
        }
        const std::string got = Slurp(fd);
        CHECK(got == want && !got.empty(), "gathered stream: %zu vs %zu bytes", got.size(), want.size());
        close(fd);
    }
    timg_hip_free(ctx, dev_out);
    timg_hip_free(ctx, packed);
    timg_hip_comm_destroy(comm);
    printf("RCCL gather to the root + ordered hand-over to the write sequencer (world 1): checked\n");
    fflush(stdout);
}

// The path real FILES take through a timg built with the twins: decoded frames in HOST memory, scaled by
// HipImageScaler where ImageScaler::Create stood (src/stb-image-source.cc:44-61, src/qoi-image-source.cc:42-77) -- on
// LOADER THREADS, each on its own context (LoaderHipContext), several at a time -- then the Hip canvas.  Against the
// reference's scaler + compose + canvases on the same frames: byte-identical streams.
static void CheckHostFramesPath() {
    timg_stub_sixel_set_lookup_mode(1);
    const int sw = 640, sh = 360, nsrc = 12;
    std::vector<uint8_t> frames((size_t)nsrc * sw * sh * 4);
    for (int i = 0; i < nsrc; ++i) {
        Framebuffer fb(sw, sh);
        rng_state = 500 + i;
        Fill(&fb, i % 3);  // opaque noise, random alpha, flat areas
        memcpy(frames.data() + (size_t)i * sw * sh * 4, fb.begin(), (size_t)sw * sh * 4);
    }
    int n = 0;
    for (int canvas_kind = 0; canvas_kind < 2; ++canvas_kind) {  // 0: quarter blocks, 1: sixel
        std::string streams[2];
        for (int hip = 0; hip < 2; ++hip) {
            volatile sig_atomic_t intr = 0;
            const int fd = memfd_create("host", 0);
            {
                DisplayOptions opts;
                opts.cell_x_px        = canvas_kind ? 9 : 2;
                opts.cell_y_px        = canvas_kind ? 18 : 2;
                opts.width            = canvas_kind ? 133 : 120;
                opts.height           = canvas_kind ? 90 : 68;
                opts.width_stretch    = canvas_kind ? 1.0f : 2.0f;
                opts.pattern_size     = 2;
                opts.bg_pattern_color.r = 200; opts.bg_pattern_color.g = 190; opts.bg_pattern_color.b = 180;
                opts.bg_pattern_color.a = 255;
                opts.bgcolor_getter = []() { rgba_t c; c.r = 30; c.g = 30; c.b = 46; c.a = 255; return c; };
                ThreadPool loaders(6);  // (more loaders than loader contexts are not needed to see them overlap)
                std::vector<std::future<ImageSource *>> loaded;
                for (int i = 0; i < nsrc; ++i) {
                    const std::function<ImageSource *()> f = [&, i]() -> ImageSource * {
                        auto *s = new HostFramesSource("host", frames.data() + (size_t)i * sw * sh * 4, 1, sw, sh, hip != 0);
                        if (!s->LoadAndScale(opts, 0, -1)) {
                            delete s;
                            return nullptr;
                        }
                        return s;
                    };
                    loaded.push_back(loaders.ExecAsync(f));
                }
                BufferedWriteSequencer seq(fd, false, 4, true, intr);
                ThreadPool pool(5);
                static SixelOptions so;
                std::unique_ptr<TerminalCanvas> canvas;
                // (a back-end switched off after a failure -- the "degrade" mode -- means the reference's class, as in the
                // patched PresentImages)
                const bool hip_canvas = hip && SharedHipContext() != nullptr;
                if (canvas_kind == 0 && hip_canvas) canvas.reset(new HipUnicodeBlockCanvas(&seq, true, false, false));
                if (canvas_kind == 0 && !hip_canvas) canvas.reset(new UnicodeBlockCanvas(&seq, true, false, false));
                if (canvas_kind == 1 && hip_canvas) canvas.reset(new HipSixelCanvas(&seq, &pool, so, opts));
                if (canvas_kind == 1 && !hip_canvas) canvas.reset(new SixelCanvas(&seq, &pool, so, opts));
                {
                    auto renderer = Renderer::Create(canvas.get(), opts, 4, 3, Duration(), Duration());
                    for (auto &fut : loaded) {
                        std::unique_ptr<ImageSource> source(fut.get());
                        CHECK(source != nullptr, "host-frames source (hip %d)", hip);
                        if (!source) continue;
                        canvas->CursorOff();
                        source->SendFrames(Duration::InfiniteFuture(), 1, intr, renderer->render_cb(""));
                        canvas->CursorOn();
                    }
                    seq.Flush();
                }
                canvas.reset();
            }
            streams[hip] = Slurp(fd);
            close(fd);
        }
        CHECK(streams[0] == streams[1] && streams[0].size() > 1000, "host frames through the twins, canvas %d: %zu (reference) vs %zu bytes",
              canvas_kind, streams[0].size(), streams[1].size());
        ++n;
    }
    printf("host frames -> HipImageScaler on loader threads -> Hip canvas: %d grids identical to the reference classes\n", n);
    fflush(stdout);
}

// What the twins cache on the device is bounded and can be given back (hip-context.h): idle scalers over ALL
// geometries (a slide show of differently sized images), pool blocks; after HipPoolTrim a scaler still scales.
static void CheckPools() {
    timg_hip_ctx *ctx = SharedHipContext();
    std::vector<timg_hip_scaler *> held;
    for (int i = 0; i < 40; ++i) {  // 40 geometries, each used once
        timg_hip_scaler *s = HipScalerAcquire(ctx, 64 + i, 48 + i, TIMG_HIP_FMT_RGBA, 20 + i % 7, 15 + i % 5, HipScalerFilter());
        CHECK(s != nullptr, "scaler %d", i);
        HipScalerRelease(s);
    }
    CHECK(HipIdleScalers() <= 24 && HipIdleScalers() >= 1, "idle scalers after 40 geometries: %zu", HipIdleScalers());
    // the most recently released geometry is still cached, the first one is not
    const size_t before = HipIdleScalers();
    timg_hip_scaler *recent = HipScalerAcquire(ctx, 64 + 39, 48 + 39, TIMG_HIP_FMT_RGBA, 20 + 39 % 7, 15 + 39 % 5, HipScalerFilter());
    CHECK(HipIdleScalers() == before - 1, "the last geometry came from the cache: %zu -> %zu", before, HipIdleScalers());
    HipScalerRelease(recent);
    void *blocks[4];
    for (int i = 0; i < 4; ++i) blocks[i] = HipPoolMalloc(ctx, 1000 + 4 * i);
    for (int i = 0; i < 4; ++i) HipPoolFree(ctx, blocks[i]);
    const size_t dropped = HipPoolTrim(ctx);
    CHECK(dropped >= 4 + 1 && HipIdleScalers() == 0, "trim dropped %zu objects, %zu scalers left", dropped, HipIdleScalers());
    CHECK(HipPoolTrim(ctx) == 0, "a second trim finds nothing");
    Framebuffer in(103, 87), out_ref(24, 19), out_hip(24, 19);
    Fill(&in, 0);
    std::unique_ptr<ImageScaler> ref = ImageScaler::Create(103, 87, ImageScaler::ColorFmt::kRGB32, 24, 19);
    std::unique_ptr<ImageScaler> hip = HipImageScaler::Create(103, 87, ImageScaler::ColorFmt::kRGB32, 24, 19);
    CHECK(ref && hip, "scalers after the trim");
    if (ref && hip) {
        ref->Scale(in, &out_ref);
        hip->Scale(in, &out_hip);
        CHECK(memcmp(out_ref.begin(), out_hip.begin(), 24 * 19 * 4) == 0, "a scaler created after the trim scales");
    }
    printf("scaler / block pools: bounded over 40 geometries, trimmed, still scaling\n");
    fflush(stdout);
}

// An animation at one place on ONE block canvas, four Sends = four device calls: with TIMG_HIP_FAIL_CALL=2 (or 3) the
// switch to the CPU sibling happens between two frames of it -- the frame after the switch must still be the frame
// DIFFERENCE the reference sends (src/unicode-block-canvas.cc:343-346), not a full frame.  First stage of `degrade`.
static void CheckBlockAnimationAcrossTheSwitch() {
    std::string streams[2];
    size_t first_frame[2] = {0, 0};
    for (int twin = 0; twin < 2; ++twin) {
        rng_state = 31337;
        volatile sig_atomic_t intr = 0;
        const int fd = memfd_create("anim", 0);
        {
            BufferedWriteSequencer seq(fd, false, 4, true, intr);
            std::unique_ptr<TerminalCanvas> canvas;
            if (twin && SharedHipContext()) canvas.reset(new HipUnicodeBlockCanvas(&seq, true, false, false));
            else canvas.reset(new UnicodeBlockCanvas(&seq, true, false, false));
            Framebuffer fb(100, 56);
            Fill(&fb, 1);
            canvas->Send(4, 0, fb, SeqType::StartOfAnimation, {});
            seq.Flush();
            first_frame[twin] = (size_t)lseek(fd, 0, SEEK_END);
            for (int f = 0; f < 3; ++f) {
                rgba_t c;
                c.r = 250; c.g = (uint8_t)(80 * f); c.b = 20; c.a = 255;
                for (int x = 10; x < 30; ++x) fb.SetPixel(x, 8 + 9 * f, c);
                canvas->Send(4, -56, fb, SeqType::AnimationFrame, {});
            }
            canvas.reset();
        }
        streams[twin] = Slurp(fd);
        close(fd);
    }
    CHECK(streams[0] == streams[1], "animation across the switch: %zu (reference) vs %zu bytes", streams[0].size(), streams[1].size());
    // (the three later frames together are far smaller than the first: they ARE differences)
    CHECK(streams[1].size() - first_frame[1] < first_frame[1] / 2, "the frames after the first are not differences: %zu + %zu bytes",
          first_frame[1], streams[1].size() - first_frame[1]);
    printf("block animation across the switch: %zu bytes, frames 2-4 are differences (%zu bytes)\n", streams[1].size(),
           streams[1].size() - first_frame[1]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    // twin_check [all|scaler|block|grid|sixel|timggrid|graphics|source|animation|autocrop|gather|hostpath|pools|bilinear|degrade] [sixel-dump-path]
    const std::string what = argc > 1 ? argv[1] : "all";
    if (!SharedHipContext()) {
        fprintf(stderr, "twin_check: no usable HIP device (%s)\n", timg_hip_last_error(nullptr));
        return 2;
    }
    if (what == "all" || what == "scaler") CheckScaler();
    if (what == "all" || what == "block") CheckBlockCanvas();
    if (what == "all" || what == "grid") CheckGridRenderer();
    if (what == "all" || what == "sixel") CheckSixelCanvas(argc > 2 ? argv[2] : nullptr);
    if (what == "all" || what == "sixelgrid" || what == "timggrid") CheckGridLikeTimg();
    if (what == "all" || what == "graphics") CheckGraphicsCanvases();
    if (what == "bilinear") {  // (own process: TIMG_HIP_FILTER=bilinear must be set before the first scaler is created)
        CheckBilinearScaler();
        if (failures) return 1;
        return 0;
    }
    if (what == "all" || what == "source") CheckImageSource();
    if (what == "all" || what == "source" || what == "animation") CheckAnimationSource();
    if (what == "all" || what == "source" || what == "autocrop") CheckAutoCropSource();
    if (what == "all" || what == "gather") CheckGatherWriter();
    if (what == "all" || what == "hostpath") CheckHostFramesPath();
    if (what == "degrade") {
        // run with TIMG_HIP_FAIL_CALL=k: the k-th device call of the process fails; the twins say so once on stderr and go
        // on with the reference's classes (cpu-sibling.h, HipImageScaler::ScaleOnCpu) -- same terminal streams
        CheckBlockAnimationAcrossTheSwitch();
        CheckGridLikeTimg();
        CheckHostFramesPath();
        printf("degrade: device failure injected (TIMG_HIP_FAIL_CALL=%s), degraded=%d: streams identical to the reference classes\n",
               getenv("TIMG_HIP_FAIL_CALL") ? getenv("TIMG_HIP_FAIL_CALL") : "-", (int)HipDegraded());
        fflush(stdout);
    }
    if (what == "all" || what == "pools") CheckPools();
    // What ran where.  Every mode but `degrade` compares DEVICE output with the reference's classes: a twin that quietly
    // went on with its CPU sibling (hip-context.h: HipDegrade) would compare the reference with itself -- that is a
    // failure here, whatever the bytes say.  (A frame the device REFUSED -- TIMG_HIP_ERR_UNSUPP -- is counted on the CPU
    // side without the switch; the modes that send such frames say how many they expect: g_expected_cpu_frames.)
    const HipTwinCounts counts = HipTwinStats();
    unsigned long on_device = 0, on_cpu = 0;
    for (int k = 0; k < kHipTwinKinds; ++k) {
        on_device += counts.device[k];
        on_cpu += counts.cpu[k];
    }
    printf("twin_check: frames on the device: scaler %lu block %lu sixel %lu graphics %lu; on the CPU: %lu; degraded %d\n",
           counts.device[0], counts.device[1], counts.device[2], counts.device[3], on_cpu, (int)HipDegraded());
    if (what != "degrade") {
        CHECK(!HipDegraded(), "the HIP back-end was switched off during the run: the comparisons above were reference against reference");
        CHECK(on_cpu == g_expected_cpu_frames, "frames encoded by the CPU sibling although no device call failed");
        if (what != "pools" && what != "gather") CHECK(on_device > 0, "no frame was produced on the device");
    }
    if (failures) {
        fprintf(stderr, "twin_check: %d failure(s)\n", failures);
        return 1;
    }
    printf("twin_check: all twins match the reference classes\n");
    fflush(stdout);
    return 0;
}
