#include "hip-sixel-canvas.h"

#include <cstdlib>

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "hip-context.h"
#include "hip-device-frames.h"
#ifdef WITH_TIMG_SIXEL
#include "sixel-canvas.h"
#endif

namespace timg {

static inline int round_to_sixel(int pixels) {  // src/sixel-canvas.cc:91-94
    pixels += 5;
    return pixels - pixels % 6;
}

HipSixelCanvas::HipSixelCanvas(BufferedWriteSequencer *ws, ThreadPool *thread_pool,
                               const SixelOptions &sixel_options,
                               const DisplayOptions &display_opts)
    : TerminalCanvas(ws), options_(display_opts), full_cell_jump_(sixel_options.full_cell_jump),
      broken_cursor_(sixel_options.known_broken_cursor_placement),
      sixel_options_(sixel_options),
      executor_(thread_pool),
      ctx_(SharedHipContext()) {
    if (!ctx_) HipFatal(ctx_, "HipSixelCanvas");
    if (HipTwinTrace()) fprintf(stderr, "HipSixelCanvas: created\n");
#ifdef WITH_TIMG_SIXEL
    // (captures what the reference class would be given -- the caller's DisplayOptions by reference, as SixelCanvas
    // itself keeps them -- and nothing of this object)
    const SixelOptions so       = sixel_options;
    const DisplayOptions *dopts = &display_opts;
    cpu_.reset(new CpuSibling([so, dopts](BufferedWriteSequencer *seq, ThreadPool *pool) -> TerminalCanvas * {
        return new SixelCanvas(seq, pool, so, *dopts);  // (constructed on first use only)
    }));
#endif
    DeviceFrameConsumerCreated();
}

size_t HipSixelCanvas::EncodeOnCpu(const std::shared_ptr<CpuSibling> &cpu_, int rc, timg_hip_ctx *ctx, char *&buffer, size_t prefix, size_t &cap, const uint8_t *pixels,
                                   bool on_device, int w, int h, const char *what) {
    if (!cpu_) HipFatal(ctx, what);  // (a timg build without libsixel has no CPU sixel canvas to go on with)
    // (rc == TIMG_HIP_ERR_UNSUPP: the device refused THIS frame's geometry; it stays selected for the others)
    HipDegradeUnless(rc, ctx, what);
    HipCountFrames(kHipTwinSixel, false);
    std::vector<uint8_t> host;
    if (on_device) {
        host.resize((size_t)w * h * 4);
        if (timg_hip_memcpy_d2h(ctx, host.data(), pixels, host.size(), nullptr) != TIMG_HIP_OK) HipFatal(ctx, what);
        pixels = host.data();
    }
    // (x = 0, dy = 0: the cursor moves of this Send are in the twin's prefix already)
    const std::string bytes = cpu_->Encode(0, pixels, w, h);
    if (prefix + bytes.size() > cap) {
        char *bigger = new char[prefix + bytes.size()];
        memcpy(bigger, buffer, prefix);
        delete[] buffer;
        buffer = bigger;
        cap    = prefix + bytes.size();
    }
    memcpy(buffer + prefix, bytes.data(), bytes.size());
    return bytes.size();
}

int HipSixelCanvas::EncodeFlags() const { return broken_cursor_ ? TIMG_HIP_SIXEL_BROKEN_CURSOR : 0; }

int HipSixelCanvas::cell_height_for_pixels(int pixels) const {  // src/sixel-canvas.cc:157-172
    assert(pixels <= 0);
    pixels = -pixels;
    if (full_cell_jump_) return -((round_to_sixel(pixels) - 6) / options_.cell_y_px + 1);
    return -((round_to_sixel(pixels) + options_.cell_y_px - 1) / options_.cell_y_px);
}

// Room for what TerminalCanvas queues in front of a frame (cursor moves, clear screen, the
// --title line: at most a terminal line of UTF-8); its length cannot be asked for.
static constexpr size_t kPrefixBudget = 16 * 1024;

HipSixelCanvas::~HipSixelCanvas() {
    rows_.reset();  // (encodes what is held, joins)
    DeviceFrameConsumerDestroyed();
}

// Frames per device call, at most.  A held batch is ONE chain of kernels, one read-back and one hand-over of its
// buffers, all of it behind the batch's LAST Send: with a queue long enough for a whole 8x8 grid (129) the twins used
// to hold all 64 images back -- nothing was encoded while the sources were still being scaled, two of the three
// encode workers never ran, and the run took 2.5x as long as with rows of 8 (17.4 against 45.1 Gpx/s, BENCH_r04).
// Beyond a grid row's worth a batch buys nothing on the device either (the chain's length is the diffusion's
// W + 2(H - 1) dependent steps whatever the batch), so longer queues now mean MORE batches in flight, not longer ones.
// Measured (profiles/r5/twin_batch_cap.txt, metric configuration / the 600-frame stream, Gpx/s by queue length
// 4 / 17 / 33 / 64 / 129): cap 8: 22 / 47 / 64 / 48 / 46 and 38 / 86 / 108 / 103 / 97; cap 16: 20 / 51 / 59 / 33 / 34
// and 39 / 53 / 84 / 88 / 88; cap 64 (round 4): 17-23 at queue 129.
// (TIMG_HIP_TWIN_BATCH_CAP: tuning / the old behaviour for comparison.)
void HipSixelCanvas::SetGridColumns(int columns) {
    Flush();
    hold_limit_ = std::min(HeldRows::HoldLimit(columns, write_sequencer_->max_queue_len()), HeldRows::BatchCap());
    if (hold_limit_ > 1 && !rows_) rows_.reset(new HeldRows(ctx_, [this](HeldBatch &b, timg_hip_ctx *c) { EncodeBatch(b, c); }, kEncodeWorkers));
}

void HipSixelCanvas::SetStreamHold(int frames) {
    Flush();
    const int by_queue = (int)write_sequencer_->max_queue_len();  // the writer's future + a full queue behind it
    stream_hold_       = std::max(1, std::min(std::min(frames, by_queue), HeldRows::BatchCap()));
    if (stream_hold_ > 1 && !rows_) rows_.reset(new HeldRows(ctx_, [this](HeldBatch &b, timg_hip_ctx *c) { EncodeBatch(b, c); }, kEncodeWorkers));
}

void HipSixelCanvas::Flush() {
    if (rows_) rows_->Drain();
}

// A held-back row: one batched encode, every future of the row fulfilled (worker thread).
void HipSixelCanvas::EncodeBatch(HeldBatch &batch, timg_hip_ctx *ctx) {
    const size_t n    = batch.frames.size();
    const size_t slot = timg_hip_sixel_max_bytes(batch.w, batch.h) * 2;
    // The batch's staging buffer: one per worker thread, kept (a fresh 29 MB mapping per batch cost its page faults
    // every time); uninitialised on purpose -- only the bytes a frame produced are touched.
    thread_local std::unique_ptr<char[]> staging;
    thread_local size_t staging_cap = 0;
    if (staging_cap < slot * n) {
        staging.reset(new char[slot * n]);
        staging_cap = slot * n;
    }
    char *const bytes_ptr = staging.get();
    struct { char *p; char *get() const { return p; } } bytes{bytes_ptr};
    std::vector<size_t> lens(n);
    const int flags = EncodeFlags();
    const bool trace = HipTwinTrace();
    const auto t0 = std::chrono::steady_clock::now();
    // (HipCall: out of device memory -- the encoder's scratch grows with the batch -- is retried once after the twins'
    // caches have been given back)
    const int rc = HipDegraded() ? TIMG_HIP_ERR_DEVICE : HipCall(ctx, [&]() {
        return timg_hip_sixel_encode(ctx, batch.data(), batch.w, batch.h, 0, 0, batch.on_device, (int)n, flags, &batch.pad,
                                     bytes.get(), slot, 0, lens.data(), nullptr);
    });
    const bool on_device = rc == TIMG_HIP_OK;
    if (on_device) HipCountFrames(kHipTwinSixel, true, n);
    const auto t1 = std::chrono::steady_clock::now();
    const size_t frame_bytes = (size_t)batch.w * batch.h * 4;
    for (size_t i = 0; i < n; ++i) {
        HeldFrame &f = batch.frames[i];
        if (on_device) {
            // the buffer the sequencer gets (freed with delete[] by the writer thread): the prefix kept at Send time +
            // the frame, exactly
            char *final = new char[f.prefix + lens[i]];
            memcpy(final, f.buffer, f.prefix);
            memcpy(final + f.prefix, bytes.get() + i * slot, lens[i]);
            delete[] f.buffer;
            f.buffer = final;
            f.cap    = f.prefix + lens[i];
        } else {
            lens[i] = EncodeOnCpu(cpu_, rc, ctx, f.buffer, f.prefix, f.cap, batch.data() + i * frame_bytes, batch.on_device, batch.w,
                                  batch.h, "timg_hip_sixel_encode");
        }
        f.promise.set_value(OutBuffer(f.buffer, f.prefix + lens[i]));
    }
    if (trace)
        fprintf(stderr, "HipSixelCanvas: batch of %zu frames %dx%d: encode call %.3f ms, hand-over %.3f ms\n", n, batch.w, batch.h,
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
}

void HipSixelCanvas::Send(int x, int dy, const Framebuffer &fb_orig, SeqType seq_type,
                          Duration end_of_frame) {
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    MoveCursorDX(x / options_.cell_x_px);

    const int w = fb_orig.width(), h = fb_orig.height();
    const bool in_stream = seq_type == SeqType::StartOfAnimation || seq_type == SeqType::AnimationFrame;
    const bool may_hold  = in_stream ? stream_hold_ > 1
                                     : hold_limit_ > 1 && seq_type == SeqType::FrameImmediate && !(have_last_x_ && x == last_x_);
    const int limit      = in_stream ? stream_hold_ : hold_limit_;
    have_last_x_ = true;
    last_x_      = x;

    // Background for the pad rows (src/sixel-canvas.cc:115-118): the getter is
    // only consulted when there are pad rows, which are fully transparent.
    timg_hip_blend pad;
    memset(&pad, 0, sizeof(pad));
    if (round_to_sixel(h) != h && options_.bgcolor_getter) {
        const rgba_t bg = options_.bgcolor_getter();
        pad.enabled     = 1;
        memcpy(&pad.bg, &bg, 4);
        memcpy(&pad.pattern, &options_.bg_pattern_color, 4);
        pad.pattern_w = options_.pattern_size * options_.cell_x_px;
        pad.pattern_h = options_.pattern_size * options_.cell_y_px / 2;
        pad.start_row = h;
    }
    // a frame of a device-resident source is encoded where it is (hip-device-frames.h)
    const uint8_t *const device = DevicePixels(fb_orig);
    if (may_hold) {
        // A held frame's bytes arrive in the batch's staging buffer and are copied into the buffer the sequencer gets:
        // that buffer is allocated THEN, with the size the frame has (a few hundred KB).  Here only the prefix is kept.
        // (Round 4 allocated the worst case -- 16 KB + 2 x 1.8 MB -- per Send: with a long queue the main thread ran
        // far ahead of the writer and every one of those was a fresh mapping.)
        char prefix[kPrefixBudget];
        const size_t prefix_len = (size_t)(AppendPrefixToBuffer(prefix) - prefix);  // must happen on this thread
        HeldFrame f;
        f.buffer = new char[prefix_len ? prefix_len : 1];
        memcpy(f.buffer, prefix, prefix_len);
        f.prefix = prefix_len;
        f.cap    = prefix_len;
        f.x      = x;
        f.dy     = dy;
        write_sequencer_->WriteBuffer(rows_->Hold(w, h, device ? device : (const uint8_t *)fb_orig.begin(), device != nullptr,
                                                  &pad, std::move(f), limit),
                                      seq_type, end_of_frame);
        return;
    }
    if (rows_) rows_->Seal();  // (a row in progress ends here; its futures are already queued in front of this one)
    const size_t cap     = kPrefixBudget + timg_hip_sixel_max_bytes(w, h) * 2;
    char *const buffer   = new char[cap];
    char *const offset   = AppendPrefixToBuffer(buffer);  // must happen on this thread
    // The framebuffer is only valid during this call: copy before going async (device frames:
    // a device-to-device copy on the copy context's stream; the source waits for that stream before it frees a frame).
    const size_t frame_bytes = (size_t)w * h * 4;
    auto pixels              = std::make_shared<std::vector<uint8_t>>(device ? 0 : frame_bytes);
    uint8_t *device_copy     = nullptr;
    timg_hip_ctx *ctx        = ctx_;
    if (device) {
        device_copy = (uint8_t *)HipPoolMalloc(ctx, frame_bytes);
        // (on the copy context's stream, hip-context.h; the encode job below waits for it)
        if (!device_copy || timg_hip_memcpy_d2d(CopyHipContext(), device_copy, device, frame_bytes, nullptr) != TIMG_HIP_OK)
            HipFatal(ctx, "HipSixelCanvas::Send");
    } else {
        memcpy(pixels->data(), fb_orig.begin(), frame_bytes);
    }
    const int flags = EncodeFlags();
    const size_t prefix_len = (size_t)(offset - buffer);
    const std::shared_ptr<CpuSibling> cpu = cpu_;
    const std::function<OutBuffer()> encode_fun = [=]() {
        size_t len = 0;
        char *buf  = buffer;
        size_t room = cap;
        if (device_copy) HipFrameCopiesDone();
        const int rc = HipDegraded() ? TIMG_HIP_ERR_DEVICE : HipCall(ctx, [&]() {
            return timg_hip_sixel_encode(ctx, device_copy ? device_copy : pixels->data(), w, h, 0, 0, device_copy != nullptr,
                                         1, flags, &pad, buf + prefix_len, room - prefix_len, 0, &len, nullptr);
        });
        if (rc == TIMG_HIP_OK)
            HipCountFrames(kHipTwinSixel, true);
        else
            len = EncodeOnCpu(cpu, rc, ctx, buf, prefix_len, room, device_copy ? device_copy : pixels->data(), device_copy != nullptr, w, h,
                              "timg_hip_sixel_encode");
        if (device_copy) HipPoolFree(ctx, device_copy);
        return OutBuffer(buf, prefix_len + len);
    };
    write_sequencer_->WriteBuffer(executor_->ExecAsync(encode_fun), seq_type, end_of_frame);
}

}  // namespace timg
