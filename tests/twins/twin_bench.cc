// tests/twins/twin_bench.cc -- TEST / MEASUREMENT DRIVER (not part of the product): the drop-in path, timed
// the way timg itself runs it, for the GPU twins and for the reference's own CPU classes.
//
// What is reproduced from src/timg.cc: a loader pool creates the image sources in parallel
// (src/timg.cc:948-968: ImageSource::Create on ThreadPool threads -- decode is replaced by frames that
// already exist: on the device for the twins, in host memory for the reference classes), the main thread
// presents them in order (PresentImages, src/timg.cc:311-396: canvas, Renderer::Create with the grid,
// CursorOff / SendFrames / CursorOn per source, sequencer->Flush()), the bytes go through the reference's
// BufferedWriteSequencer (queue length 4 = src/timg.cc:972, 2 * columns + 1: what lets the twins hold a
// whole grid row, held-rows.h) into /dev/null.
//
//   GPU path: HipRawRGBASource ("synth:..." frames generated in device memory) -> Renderer ->
//             HipSixelCanvas / HipUnicodeBlockCanvas
//   CPU path: HostFramesSource (tests/twins/host-frames-source.h: the reference's ImageScaler +
//             AlphaComposeBackground on host frames) -> Renderer -> SixelCanvas (the reference's class over
//             the oracle's libsixel restatement -- libsixel is not in the tree, parity unpinned) /
//             UnicodeBlockCanvas
//
// Configurations (BASELINE.json): c2 one 4K frame -> 800x450 sixel; c3 64 4K frames -> 200x56 quarter blocks,
// grid 8x8; c4 an N-frame 4K stream -> 800x450 sixel; metric 64 4K frames -> 800x450 sixel, grid 8x8.
// Prints one JSON object per (configuration, path, queue length).  The timed region starts when the first
// source is submitted to the loader pool and ends after Flush() with the canvas destroyed; the host frames
// of the CPU path exist before it starts (a real run would have decoded them from files).
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <string>
#include <thread>
#include <vector>

#include "buffered-write-sequencer.h"
#include "display-options.h"
#include "framebuffer.h"
#include "hip-context.h"
#include "hip-raw-rgba-source.h"
#include "hip-sixel-canvas.h"
#include "hip-unicode-block-canvas.h"
#include "host-frames-source.h"
#include "image-source.h"
#include "renderer.h"
#include "sixel-canvas.h"
#include "sixel.h"
#include "thread-pool.h"
#include "unicode-block-canvas.h"

using namespace timg;

struct Config {
    const char *name;
    int sources;        // image sources (one per "file")
    int frames_each;    // frames per source (> 1: a stream)
    int in_w, in_h;
    bool sixel;
    int cols, rows;     // grid
    int cell_w, cell_h; // DisplayOptions::width / height per grid cell
    float width_stretch;
    int cell_x_px, cell_y_px;
};

static const Config kConfigs[] = {
    {"c2", 1, 1, 3840, 2160, true, 1, 1, 800, 450, 1.0f, 9, 18},
    {"c3", 64, 1, 3840, 2160, false, 8, 8, 200, 112, 2.0f, 2, 2},
    {"c4", 1, 600, 3840, 2160, true, 1, 1, 800, 450, 1.0f, 9, 18},
    {"metric", 64, 1, 3840, 2160, true, 8, 8, 800, 450, 1.0f, 9, 18},
};

struct RunResult {
    double seconds = 0;
    size_t bytes   = 0;
    int frames     = 0;
};

static double Now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One run like src/timg.cc.  make_source(i, opts): called on loader-pool threads.
static RunResult RunLikeTimg(const Config &c, bool gpu, size_t queue_len, int loader_threads,
                             const std::function<ImageSource *(int, const DisplayOptions &)> &make_source) {
    volatile sig_atomic_t intr = 0;
    const int fd = open("/dev/null", O_WRONLY);
    RunResult res;
    DisplayOptions opts;
    opts.width          = c.cell_w;
    opts.height         = c.cell_h;
    opts.cell_x_px      = c.cell_x_px;
    opts.cell_y_px      = c.cell_y_px;
    opts.width_stretch  = c.width_stretch;
    opts.bgcolor_getter = []() { rgba_t bg; bg.r = 0x1e; bg.g = 0x1e; bg.b = 0x2e; bg.a = 255; return bg; };
    // TWIN_BENCH_TIMELINE in the environment: when (ms after the start) every source was loaded and presented
    static const bool timeline = getenv("TWIN_BENCH_TIMELINE") != nullptr;
    std::vector<double> t_loaded(c.sources, 0.0), t_got(c.sources, 0.0), t_sent(c.sources, 0.0);
    const double t0 = Now();
    {
        ThreadPool loaders(loader_threads);
        std::vector<std::future<ImageSource *>> loaded;
        for (int i = 0; i < c.sources; ++i) {
            const std::function<ImageSource *()> f = [i, &opts, &make_source, &t_loaded, t0]() {
                ImageSource *s = make_source(i, opts);
                t_loaded[i]    = Now() - t0;
                return s;
            };
            loaded.push_back(loaders.ExecAsync(f));
        }
        BufferedWriteSequencer seq(fd, false, queue_len, true, intr);
        {
            std::unique_ptr<ThreadPool> compression_pool;
            std::unique_ptr<TerminalCanvas> canvas;
            static SixelOptions so;
            // images per device call: a grid row -- or, when the queue is long enough for it, several rows (the twins
            // bound it by what the queue holds: HeldRows::HoldLimit)
            const int hold = queue_len >= (size_t)(2 * c.cols * c.rows + 1) ? c.cols * c.rows : c.cols;
            if (c.sixel) {
                compression_pool.reset(new ThreadPool(seq.max_queue_len() + 1));  // src/timg.cc:332-337
                if (gpu) {
                    auto *cv = new HipSixelCanvas(&seq, compression_pool.get(), so, opts);
                    cv->SetGridColumns(hold);
                    if (c.frames_each > 1) cv->SetStreamHold((int)seq.max_queue_len());
                    canvas.reset(cv);
                } else {
                    canvas.reset(new SixelCanvas(&seq, compression_pool.get(), so, opts));
                }
            } else if (gpu) {
                auto *cv = new HipUnicodeBlockCanvas(&seq, true, false, false);
                cv->SetGridColumns(hold);
                canvas.reset(cv);
            } else {
                canvas.reset(new UnicodeBlockCanvas(&seq, true, false, false));
            }
            {
                auto renderer = Renderer::Create(canvas.get(), opts, c.cols, c.rows, Duration(), Duration());
                int idx = -1;
                for (auto &fut : loaded) {
                    std::unique_ptr<ImageSource> source(fut.get());
                    t_got[++idx] = Now() - t0;
                    if (!source) {
                        fprintf(stderr, "twin_bench: a source could not be created\n");
                        exit(1);
                    }
                    canvas->CursorOff();
                    source->SendFrames(Duration::InfiniteFuture(), 1, intr, renderer->render_cb(""));
                    canvas->CursorOn();
                    t_sent[idx] = Now() - t0;
                    ++res.frames;
                }
                const double t_flush = Now() - t0;
                seq.Flush();
                if (timeline) {
                    fprintf(stderr, "timeline %s (ms): loaded", c.name);
                    for (int i = 0; i < c.sources; ++i) fprintf(stderr, " %.2f", t_loaded[i] * 1e3);
                    fprintf(stderr, "\ntimeline %s (ms): presented", c.name);
                    for (int i = 0; i < c.sources; ++i) fprintf(stderr, " %.2f", t_sent[i] * 1e3);
                    fprintf(stderr, "\ntimeline %s (ms): flush from %.2f to %.2f\n", c.name, t_flush * 1e3, (Now() - t0) * 1e3);
                }
            }
            canvas.reset();
            compression_pool.reset();
        }
        res.bytes = (size_t)seq.bytes_total();
    }
    res.seconds = Now() - t0;
    close(fd);
    return res;
}

int main(int argc, char **argv) {
    std::string which = "c2,c3,c4,metric", paths = "gpu,host,cpu";
    int frames_override = 0, cpu_frames_cap = 128, repeat = 3, loader_threads = 0;
    std::vector<size_t> queues;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--config") which = next();
        else if (a == "--paths") paths = next();
        else if (a == "--frames") frames_override = atoi(next());
        else if (a == "--cpu-frames") cpu_frames_cap = atoi(next());
        else if (a == "--repeat") repeat = atoi(next());
        else if (a == "--loader-threads") loader_threads = atoi(next());
        else if (a == "--queue") queues.push_back((size_t)atoi(next()));
        else {
            fprintf(stderr, "usage: twin_bench [--config c2,c3,c4,metric] [--paths gpu,cpu] [--frames N] [--cpu-frames N] "
                            "[--repeat R] [--loader-threads T] [--queue Q]...\n");
            return 2;
        }
    }
    timg_hip_ctx *ctx = SharedHipContext();
    if (!ctx) {
        fprintf(stderr, "twin_bench: no usable HIP device (%s)\n", timg_hip_last_error(nullptr));
        return 2;
    }
    timg_stub_sixel_set_lookup_mode(0);  // the CPU side runs libsixel's own rule (first hit), as the reference does
    const int cores = (int)std::thread::hardware_concurrency();
    // src/timg.cc:153-154: 3/4 of the hardware threads load and scale
    const int ref_loaders = loader_threads > 0 ? loader_threads : std::max(1, cores * 3 / 4);
    for (const Config &base : kConfigs) {
        if ((',' + which + ',').find(std::string(",") + base.name + ",") == std::string::npos) continue;
        Config c = base;
        if (frames_override > 0) (c.frames_each > 1 ? c.frames_each : c.sources) = frames_override;
        if (c.frames_each == 1 && c.sources < c.cols * c.rows) c.rows = std::max(1, (c.sources + c.cols - 1) / c.cols);
        std::vector<size_t> qs = queues;
        if (qs.empty()) {
            qs.push_back(4);  // src/timg.cc:972
            if (c.cols > 1) qs.push_back((size_t)(2 * c.cols + 1));             // a grid row per device call
            if (c.cols > 1) qs.push_back((size_t)(4 * c.cols + 1));             // two rows: what integration/timg-hip.patch sets
            if (c.cols > 1) qs.push_back((size_t)(2 * c.cols * c.rows + 1));    // the whole grid
            if (c.frames_each > 1) qs.push_back(64);  // a stream: frames held per device call = the queue
        }
        const size_t frame_bytes = (size_t)c.in_w * c.in_h * 4;
        for (const char *path : {"gpu", "host", "cpu"}) {
            if ((',' + paths + ',').find(std::string(",") + path + ",") == std::string::npos) continue;
            // "host": what real FILES take through a timg built with the twins -- the decoded frame is in host memory,
            // HipImageScaler uploads, scales and composes it, the Hip canvas encodes the (host) result
            const bool host = !strcmp(path, "host");
            const bool gpu  = !strcmp(path, "gpu") || host;
            Config rc = c;
            std::vector<uint8_t> host_frames;
            if (!gpu || host) {  // a bounded sample of the same workload; the frames exist before the clock starts
                const int total = std::min(c.sources * c.frames_each, host ? std::max(cpu_frames_cap, 64) : cpu_frames_cap);
                if (c.frames_each > 1) rc.frames_each = total;
                else rc.sources = total;
                if (rc.frames_each == 1) rc.rows = std::max(1, (rc.sources + rc.cols - 1) / rc.cols);
                host_frames.resize(frame_bytes * total);
                if (timg_hip_synth_frames(ctx, TIMG_HIP_SYNTH_PHOTO, c.in_w, c.in_h, 0, 0, total, host_frames.data(), 0, 0,
                                          nullptr) != TIMG_HIP_OK) {
                    fprintf(stderr, "twin_bench: synth frames: %s\n", timg_hip_last_error(ctx));
                    return 1;
                }
            }
            // the frames as a decoder leaves them: Framebuffers in host memory, before the clock starts (both host-memory
            // paths alike)
            std::vector<std::unique_ptr<Framebuffer>> decoded_own;
            std::vector<Framebuffer *> decoded;
            for (size_t f = 0; f * frame_bytes < host_frames.size(); ++f) {
                decoded_own.emplace_back(new Framebuffer(c.in_w, c.in_h));
                memcpy((void *)decoded_own.back()->begin(), host_frames.data() + f * frame_bytes, frame_bytes);
                decoded.push_back(decoded_own.back().get());
            }
            const int threads = (gpu && !host) ? std::min(ref_loaders, std::max(1, std::min(rc.sources, 8)))
                                               : std::min(host ? std::min(ref_loaders, 16) : ref_loaders, std::max(1, rc.sources));
            auto make = [&](int i, const DisplayOptions &opts) -> ImageSource * {
                if (gpu && !host) {
                    char name[96];
                    if (rc.frames_each > 1) snprintf(name, sizeof(name), "synth:photo:%dx%d:0:0:%d", rc.in_w, rc.in_h, rc.frames_each);
                    else snprintf(name, sizeof(name), "synth:photo:%dx%d:0:%d", rc.in_w, rc.in_h, i);
                    return HipRawRGBASource::TryCreate(name, opts, 0, -1);
                }
                auto *s = new HostFramesSource("host", host_frames.data() + (rc.frames_each > 1 ? 0 : frame_bytes * i),
                                               rc.frames_each, rc.in_w, rc.in_h, host);
                std::vector<Framebuffer *> *mine = new std::vector<Framebuffer *>(  // (leaked: a handful of pointers per source)
                    decoded.begin() + (rc.frames_each > 1 ? 0 : i), decoded.begin() + (rc.frames_each > 1 ? rc.frames_each : i + 1));
                s->UseDecodedFrames(mine);
                if (!s->LoadAndScale(opts, 0, -1)) {
                    delete s;
                    return nullptr;
                }
                return s;
            };
            for (size_t q : qs) {
                RunResult best;
                for (int r = 0; r < repeat + 1; ++r) {  // (first run: warm-up -- code objects, scratch, clocks)
                    const RunResult rr = RunLikeTimg(rc, gpu, q, threads, make);
                    if (r == 0) continue;
                    if (best.seconds == 0 || rr.seconds < best.seconds) best = rr;
                }
                const int frames  = rc.sources * rc.frames_each;
                const double mpx  = (double)frames * c.in_w * c.in_h / 1e6 / best.seconds;
                printf("{\"config\": \"%s\", \"path\": \"%s\", \"frames\": %d, \"in\": \"%dx%d\", \"canvas\": \"%s\", \"grid\": \"%dx%d\", "
                       "\"queue_len\": %zu, \"loader_threads\": %d, \"host_cores\": %d, \"seconds\": %.5f, \"ms_per_frame\": %.4f, "
                       "\"mpx_per_s\": %.1f, \"bytes_written\": %zu, \"classes\": \"%s\"}\n",
                       c.name, path, frames, c.in_w, c.in_h, c.sixel ? "sixel" : "quarter", rc.cols, rc.rows, q, threads, cores,
                       best.seconds, best.seconds * 1e3 / frames, mpx, best.bytes,
                       host ? "host frames -> HipImageScaler::ScaleAndCompose (upload, scale, compose, download) + reference Renderer "
                              "+ Hip canvas + reference BufferedWriteSequencer"
                       : gpu ? "HipRawRGBASource + reference Renderer + Hip canvas + reference BufferedWriteSequencer"
                           : (c.sixel ? "reference ImageScaler/Framebuffer/Renderer/SixelCanvas (libsixel = oracle restatement, "
                                        "first-hit rule)/BufferedWriteSequencer"
                                      : "reference ImageScaler/Framebuffer/Renderer/UnicodeBlockCanvas/BufferedWriteSequencer"));
                fflush(stdout);
            }
        }
    }
    return 0;
}
