#include "hip-unicode-block-canvas.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hip-context.h"
#include "unicode-block-canvas.h"
#include "hip-device-frames.h"

namespace timg {

// Room for what TerminalCanvas queues in front of a frame (cursor moves, clear screen, the
// --title line: at most a terminal line of UTF-8); its length cannot be asked for.
static constexpr size_t kPrefixBudget = 16 * 1024;

HipUnicodeBlockCanvas::HipUnicodeBlockCanvas(BufferedWriteSequencer *ws, bool use_quarter,
                                             bool use_upper_half_block, bool use_256_color)
    : TerminalCanvas(ws),
      ctx_(SharedHipContext()),
      flags_((use_quarter ? TIMG_HIP_BLOCK_QUARTER : 0) | (use_upper_half_block ? TIMG_HIP_BLOCK_UPPER : 0) |
             (use_256_color ? TIMG_HIP_BLOCK_COLOR256 : 0)) {
    if (!ctx_ || timg_hip_block_canvas_create(ctx_, flags_, &canvas_) != TIMG_HIP_OK)
        HipFatal(ctx_, "HipUnicodeBlockCanvas");
    if (HipTwinTrace()) fprintf(stderr, "HipUnicodeBlockCanvas: created (flags %d)\n", flags_);
    const bool quarter = use_quarter, upper = use_upper_half_block, c256 = use_256_color;
    cpu_.reset(new CpuSibling([quarter, upper, c256](BufferedWriteSequencer *seq, ThreadPool *) -> TerminalCanvas * {
        return new UnicodeBlockCanvas(seq, quarter, upper, c256);  // (constructed on first use only)
    }));
    DeviceFrameConsumerCreated();
}

void HipUnicodeBlockCanvas::RememberFrame(int x, const uint8_t *pixels, bool on_device, int width, int height) {
    prev_valid_ = !on_device;  // (a device-resident frame is not brought back for this: after a switch its successor is a full frame)
    if (on_device) return;
    prev_pixels_.assign(pixels, pixels + (size_t)width * height * 4);
    prev_x_ = x;
    prev_w_ = width;
    prev_h_ = height;
}

size_t HipUnicodeBlockCanvas::EncodeOnCpu(HeldFrame &p, const uint8_t *pixels, bool on_device, int width, int height,
                                          const char *what) {
    HipDegrade(ctx_, what);
    HipCountFrames(kHipTwinBlock, false);
    std::vector<uint8_t> host;
    if (on_device) {  // (a frame of a device-resident source: its pixels have to come back first -- if the device still answers)
        host.resize((size_t)width * height * 4);
        if (timg_hip_memcpy_d2h(ctx_, host.data(), pixels, host.size(), nullptr) != TIMG_HIP_OK) HipFatal(ctx_, what);
        pixels = host.data();
    }
    // The sibling starts where the device canvas stopped: it is shown the frame the device saw last (its bytes are
    // thrown away), so that THIS frame is a difference exactly when the reference would send one.
    if (!sibling_primed_) {
        sibling_primed_ = true;
        if (prev_valid_) (void)cpu_->Encode(prev_x_, prev_pixels_.data(), prev_w_, prev_h_, 0);
    }
    // (dy goes to the sibling -- the difference rule needs it -- which then writes the cursor-up jump this twin has
    // already queued in p's prefix: taken off again.  Nothing emitted stays nothing: src/unicode-block-canvas.cc:390-395.)
    std::string bytes = cpu_->Encode(p.x, pixels, width, height, p.dy);
    if (p.dy < 0 && !bytes.empty()) {
        char jump[32];
        const size_t n = (size_t)snprintf(jump, sizeof(jump), "\033[%dA", -cell_height_for_pixels(p.dy));
        if (cell_height_for_pixels(p.dy) != 0 && bytes.compare(0, n, jump) == 0) bytes.erase(0, n);
    }
    if (p.prefix + bytes.size() > p.cap) {
        char *bigger = new char[p.prefix + bytes.size()];
        memcpy(bigger, p.buffer, p.prefix);
        delete[] p.buffer;
        p.buffer = bigger;
        p.cap    = p.prefix + bytes.size();
    }
    memcpy(p.buffer + p.prefix, bytes.data(), bytes.size());
    return bytes.size();
}

HipUnicodeBlockCanvas::~HipUnicodeBlockCanvas() {
    rows_.reset();  // encodes what is held (before ~TerminalCanvas writes a left-over cursor prefix)
    DeviceFrameConsumerDestroyed();
    timg_hip_block_canvas_destroy(canvas_);
}

void HipUnicodeBlockCanvas::SetGridColumns(int columns) {
    Flush();
    // (capped like the sixel twin's batches: with a queue long enough for a whole 8x8 grid one 64-frame batch held every
    // Send back until the last source was scaled -- 31 Gpx/s at queue 129 against 40 at queue 17, profiles/r5/twin_bench.txt)
    hold_limit_ = std::min(HeldRows::HoldLimit(columns, write_sequencer_->max_queue_len()), HeldRows::BatchCap());
    if (hold_limit_ > 1 && !rows_) rows_.reset(new HeldRows(ctx_, [this](HeldBatch &b, timg_hip_ctx *) { EncodeBatch(b); }));  // (one worker: the canvas has state)
}

void HipUnicodeBlockCanvas::Flush() {
    if (rows_) rows_->Drain();
}

// One Send through the stateful device canvas (which decides about the frame difference the
// way Send does, src/unicode-block-canvas.cc:343-346) and on to the sequencer.
void HipUnicodeBlockCanvas::SendNow(HeldFrame &p, const uint8_t *pixels, bool on_device, int width, int height,
                                    SeqType seq_type, Duration end_of_frame) {
    size_t len = 0;
    if (HipDegraded() ||
        HipCall(ctx_, [&]() {
            return timg_hip_block_canvas_send(canvas_, p.x, p.dy, pixels, width, height, 0, on_device, p.buffer + p.prefix,
                                              p.cap - p.prefix, &len, nullptr);
        }) != TIMG_HIP_OK) {
        len = EncodeOnCpu(p, pixels, on_device, width, height, "timg_hip_block_canvas_send");
    } else {
        HipCountFrames(kHipTwinBlock, true);
        RememberFrame(p.x, pixels, on_device, width, height);
    }
    // nothing emitted: the reference keeps the buffer size zero, dropping the
    // cursor jump as well (:390-395)
    write_sequencer_->WriteBuffer(OutBuffer(p.buffer, len ? p.prefix + len : 0), seq_type, end_of_frame);
}

// A held-back row (worker thread; the caller's thread waits in Drain() before it touches the
// device canvas again).
void HipUnicodeBlockCanvas::EncodeBatch(HeldBatch &batch) {
    const size_t n = batch.frames.size(), frame_bytes = (size_t)batch.w * 4 * batch.h;
    // all but the last frame of the row: one launch.  They are full encodes by construction
    // (every one of them follows a Send at another x).
    if (n > 1) {
        const size_t slot = timg_hip_block_max_bytes(batch.w, batch.h);
        std::unique_ptr<char[]> bytes(new char[slot * (n - 1)]);  // (uninitialised on purpose)
        std::vector<size_t> lens(n - 1);
        std::vector<int> xs(n - 1);
        for (size_t i = 0; i + 1 < n; ++i) xs[i] = batch.frames[i].x;
        const bool on_device = !HipDegraded() &&
            HipCall(ctx_, [&]() {
                return timg_hip_block_encode_grid(ctx_, batch.data(), batch.w, batch.h, 0, 0, batch.on_device, (int)(n - 1),
                                                  flags_, xs.data(), bytes.get(), slot, 0, lens.data(), nullptr);
            }) == TIMG_HIP_OK;
        for (size_t i = 0; i + 1 < n; ++i) {
            HeldFrame &p = batch.frames[i];
            if (on_device) {
                memcpy(p.buffer + p.prefix, bytes.get() + i * slot, lens[i]);
                HipCountFrames(kHipTwinBlock, true);
            } else
                lens[i] = EncodeOnCpu(p, batch.data() + i * frame_bytes, batch.on_device, batch.w, batch.h, "timg_hip_block_encode_grid");
            p.promise.set_value(OutBuffer(p.buffer, lens[i] ? p.prefix + lens[i] : 0));
        }
    }
    // the last one through the stateful canvas, so that an animation that continues at its
    // place finds it as the previous frame; what the canvas remembers from before the row
    // is not the previous Send any more
    timg_hip_block_canvas_forget(canvas_);
    HeldFrame &p = batch.frames[n - 1];
    size_t len   = 0;
    if (HipDegraded() ||
        HipCall(ctx_, [&]() {
            return timg_hip_block_canvas_send(canvas_, p.x, p.dy, batch.data() + (n - 1) * frame_bytes, batch.w, batch.h, 0,
                                              batch.on_device, p.buffer + p.prefix, p.cap - p.prefix, &len, nullptr);
        }) != TIMG_HIP_OK)
        len = EncodeOnCpu(p, batch.data() + (n - 1) * frame_bytes, batch.on_device, batch.w, batch.h, "timg_hip_block_canvas_send");
    else {
        HipCountFrames(kHipTwinBlock, true);
        RememberFrame(p.x, batch.data() + (n - 1) * frame_bytes, batch.on_device, batch.w, batch.h);
    }
    p.promise.set_value(OutBuffer(p.buffer, len ? p.prefix + len : 0));
}

// src/unicode-block-canvas.cc:323-403.  What stays on the host is what has to:
// the cursor prefix (queued by the renderer on this thread) and the hand-over
// of exactly one buffer per Send to the write sequencer.
void HipUnicodeBlockCanvas::Send(int x, int dy, const Framebuffer &fb, SeqType seq_type,
                                 Duration end_of_frame) {
    const int width = fb.width(), height = fb.height();
    // a Send at the previous position may be a frame difference: it needs its predecessor encoded
    const bool may_hold = hold_limit_ > 1 && !(have_last_x_ && x == last_x_);
    have_last_x_ = true;
    last_x_      = x;

    // RequestBuffers (:405-424) leaves room for the prefix in front of the frame
    HeldFrame p;
    p.cap    = timg_hip_block_max_bytes(width, height) + kPrefixBudget;
    p.buffer = new char[p.cap];
    if (dy < 0) MoveCursorDY(cell_height_for_pixels(dy));
    p.prefix = (size_t)(AppendPrefixToBuffer(p.buffer) - p.buffer);
    p.x      = x;
    p.dy     = dy;
    // a frame of a device-resident source is encoded where it is (hip-device-frames.h)
    const uint8_t *const device = DevicePixels(fb);
    const uint8_t *const pixels = device ? device : (const uint8_t *)fb.begin();
    if (!may_hold) {
        if (rows_) rows_->Drain();  // (the device canvas has to have seen the row's last frame)
        SendNow(p, pixels, device != nullptr, width, height, seq_type, end_of_frame);
        return;
    }
    write_sequencer_->WriteBuffer(rows_->Hold(width, height, pixels, device != nullptr, nullptr, std::move(p), hold_limit_),
                                  seq_type, end_of_frame);
}

}  // namespace timg
