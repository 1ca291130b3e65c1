#include "held-rows.h"

#include "hip-context.h"

#include <algorithm>
#include <cstdlib>
#include <utility>

namespace timg {

constexpr std::chrono::milliseconds HeldRows::kIdle;

HeldRows::HeldRows(timg_hip_ctx *ctx, std::function<void(HeldBatch &, timg_hip_ctx *)> encode, int workers)
    : ctx_(ctx), encode_(std::move(encode)) {
    workers_.emplace_back(&HeldRows::Work, this, ctx_);
    for (int k = 1; k < workers; ++k) {
        timg_hip_ctx *c = ExtraHipContext(k);  // process-wide: its scratch and code objects stay warm between canvases
        if (!c) break;                         // (fewer workers: still correct)
        workers_.emplace_back(&HeldRows::Work, this, c);
    }
}

void HeldRows::SealLocked() {
    sealed_.push_back(std::move(open_));
    open_      = HeldBatch();
    have_open_ = false;
}

HeldRows::~HeldRows() {
    {
        std::lock_guard<std::mutex> l(mu_);
        if (have_open_) {
            sealed_.push_back(std::move(open_));
            open_      = HeldBatch();
            have_open_ = false;
        }
        exiting_ = true;
    }
    wake_.notify_all();
    for (auto &w : workers_) w.join();  // (the workers empty sealed_ before they leave)
}

int HeldRows::BatchCap() {
    static const int cap = []() {
        const char *e = getenv("TIMG_HIP_TWIN_BATCH_CAP");
        const int v   = e ? atoi(e) : 0;
        return v > 0 ? v : 8;
    }();
    return cap;
}

int HeldRows::HoldLimit(int grid_columns, size_t sequencer_queue_len) {
    // the writer holds the first future; behind it the queue must take the other futures of the
    // batch and one control write (CursorOn) per image
    const int by_queue = (int)((sequencer_queue_len + 1) / 2);
    return std::max(1, std::min(grid_columns, by_queue));
}

std::future<OutBuffer> HeldRows::Hold(int w, int h, const uint8_t *pixels, bool on_device, const timg_hip_blend *pad,
                                      HeldFrame &&frame, int limit) {
    std::future<OutBuffer> result = frame.promise.get_future();
    const size_t frame_bytes      = (size_t)w * h * 4;
    {
        std::lock_guard<std::mutex> l(mu_);
        if (have_open_ && (open_.w != w || open_.h != h || open_.on_device != on_device)) SealLocked();
        if (!have_open_) {
            open_.w         = w;
            open_.h         = h;
            open_.on_device = on_device;
            have_open_      = true;
            if (on_device) {
                open_.dev_pixels = (uint8_t *)HipPoolMalloc(ctx_, frame_bytes * (size_t)limit);
                if (!open_.dev_pixels) abort();
            }
        }
        if (pad) open_.pad = *pad;
        if (on_device) {
            // (on the copy context's stream, hip-context.h: the worker waits for it before it encodes, the source before
            // it frees the frame)
            if (timg_hip_memcpy_d2d(CopyHipContext(), open_.dev_pixels + open_.frames.size() * frame_bytes, pixels, frame_bytes,
                                    nullptr) != TIMG_HIP_OK)
                abort();
        } else {
            open_.pixels.insert(open_.pixels.end(), pixels, pixels + frame_bytes);
        }
        open_.frames.push_back(std::move(frame));
        deadline_ = std::chrono::steady_clock::now() + kIdle;
        if ((int)open_.frames.size() >= limit) SealLocked();
    }
    wake_.notify_all();  // (the worker also has to learn the new deadline)
    return result;
}

void HeldRows::Seal() {
    {
        std::lock_guard<std::mutex> l(mu_);
        if (!have_open_) return;
        sealed_.push_back(std::move(open_));
        open_      = HeldBatch();
        have_open_ = false;
    }
    wake_.notify_all();
}

void HeldRows::Drain() {
    Seal();
    std::unique_lock<std::mutex> l(mu_);
    idle_.wait(l, [this]() { return sealed_.empty() && busy_ == 0; });
}

void HeldRows::Work(timg_hip_ctx *worker_ctx) {
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
        if (sealed_.empty()) {
            if (exiting_) return;
            if (have_open_) {
                if (wake_.wait_until(l, deadline_) == std::cv_status::timeout && have_open_ &&
                    std::chrono::steady_clock::now() >= deadline_) {
                    sealed_.push_back(std::move(open_));  // nothing arrived for kIdle
                    open_      = HeldBatch();
                    have_open_ = false;
                }
            } else {
                wake_.wait(l);
            }
            continue;
        }
        HeldBatch batch = std::move(sealed_.front());
        sealed_.pop_front();
        ++busy_;
        l.unlock();
        // device frames were gathered with copies on the COPY context's stream (Hold): they have to have landed
        // before this worker's stream reads them
        if (batch.on_device && timg_hip_sync(CopyHipContext(), nullptr) != TIMG_HIP_OK) abort();
        encode_(batch, worker_ctx);
        if (batch.dev_pixels) {
            // (the pool hands the block out again at once -- to a row whose copies run on another stream: this worker's
            // stream must be done with it)
            (void)timg_hip_sync(worker_ctx, nullptr);
            HipPoolFree(ctx_, batch.dev_pixels);
        }
        l.lock();
        --busy_;
        if (sealed_.empty() && busy_ == 0) idle_.notify_all();
    }
}

}  // namespace timg
