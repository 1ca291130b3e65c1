// timg_amd/csrc/context.h -- internal definitions behind the opaque handles of
// include/timg_hip.h.
#ifndef TIMG_AMD_CONTEXT_H
#define TIMG_AMD_CONTEXT_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/timg_hip.h"
#include "dev_alloc.h"
#include "device_plan.h"
#include "resample_plan.h"

// Growable device / pinned-host scratch area.
struct TimgBuffer {
    void *ptr     = nullptr;
    size_t bytes  = 0;
    bool pinned   = false;
    hipError_t Reserve(size_t want) {
        if (want <= bytes) return hipSuccess;
        if (ptr) {
            if (pinned)
                (void)hipHostFree(ptr);
            else
                (void)timg_amd::DevFree(ptr);
            ptr   = nullptr;
            bytes = 0;
        }
        // grow geometrically so repeated calls settle quickly (guard mode, dev_alloc.h: exactly `want`,
        // so that nothing hides in the slack)
        size_t cap = (!pinned && timg_amd::GuardMode()) ? want : want + want / 4 + 256;
        hipError_t e = pinned ? hipHostMalloc(&ptr, cap, hipHostMallocDefault)
                              : timg_amd::DevMalloc(&ptr, cap);
        if (e == hipSuccess) bytes = cap;
        return e;
    }
    void Release() {
        if (ptr) {
            if (pinned)
                (void)hipHostFree(ptr);
            else
                (void)timg_amd::DevFree(ptr);
        }
        ptr   = nullptr;
        bytes = 0;
    }
};

struct timg_hip_ctx {
    int device          = 0;
    hipStream_t stream  = nullptr;  // used when the caller passes stream == NULL
    std::mutex mu;                  // serialises scratch use per context
    std::string last_error;
    int cu_count = 0;

    // scratch: [0] staged source, [1] staged destination, [2] flags/lengths,
    // [3..] canvas intermediates
    TimgBuffer dev[8];
    TimgBuffer pin[4];

    // Side streams for kernels that are latency-bound on a handful of CUs (the serial
    // stages of the sixel canvas): a batch is cut into groups whose chains run
    // concurrently, forked from and joined to the caller's stream with events.
    static constexpr int kSideStreams = 4;
    hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr, join_event[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t hook_event[2 * kSideStreams] = {};  // (timing: around a piece's scale launches)
    hipError_t EnsureSideStreams() {
        if (fork_event) return hipSuccess;
        for (auto &st : side) {
            hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
        }
        for (auto &ev : join_event) {
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        for (auto &ev : hook_event) {
            hipError_t e = hipEventCreate(&ev);
            if (e != hipSuccess) return e;
        }
        return hipEventCreateWithFlags(&fork_event, hipEventDisableTiming);
    }

    // The sixel encoder's scratch (dev[5]) belongs to ONE call at a time.  A blocking call has left it when it returns;
    // an asynchronous one (timg_hip_sixel_encode_async) has not: `sixel_done` is recorded behind its last kernel on
    // `sixel_stream`.  The next sixel call on ANOTHER stream waits for it on the device (hipStreamWaitEvent); one that
    // must grow the scratch waits for it on the host before the old block is freed.
    hipEvent_t sixel_done   = nullptr;   // (unused since round 6: the event is the last job's, sixel_last_job->done)
    struct timg_hip_sixel_job *sixel_last_job = nullptr;  // the asynchronous call in flight, or the last one (cleared by its destroy)
    hipStream_t sixel_stream = nullptr;
    bool sixel_in_flight    = false;

    // streams handed out by timg_hip_stream_create (CU-masked / prioritised), ended with the context at the latest;
    // the event timg_hip_stream_wait_stream orders two of them with (recorded and waited for under `mu`)
    std::vector<hipStream_t> owned_streams;
    hipEvent_t order_event = nullptr;

    timg_hip_ctx() {
        for (auto &p : pin) p.pinned = true;
    }
    std::mutex err_mu;  // last_error is written from whichever thread fails (entry points that
                        // work on device memory only do not hold `mu`)
    int Fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        std::lock_guard<std::mutex> l(err_mu);
        last_error = buf;
        return code;
    }
    // A device call failed, possibly after copies and kernels on this context's scratch memory
    // were enqueued: nothing may still be in flight when the caller's lock on `mu` is released
    // and the next caller reuses (or re-allocates) that scratch.
    // (Out of device memory is its own code: the one failure a caller can do something about -- give back what it
    // caches and call again, timg_amd/twins/hip-context.h HipCall.)
    int FailHip(hipError_t e, const char *what) {
        (void)hipDeviceSynchronize();
        if (e == hipErrorOutOfMemory) (void)hipGetLastError();  // (not sticky: the next call starts clean)
        return Fail(e == hipErrorOutOfMemory ? TIMG_HIP_ERR_NOMEM : TIMG_HIP_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
    }
    hipStream_t Stream(void *s) { return s ? (hipStream_t)s : stream; }
};

// One call of timg_hip_sixel_encode_async in flight: where its frames' byte counts land (pinned, + the device's error
// word) and the event behind them.
struct timg_hip_sixel_job {
    timg_hip_ctx *ctx          = nullptr;
    hipEvent_t done            = nullptr;
    unsigned long long *len_h  = nullptr;  // max_frames + 1 words of pinned host memory
    int max_frames             = 0;
    int n                      = 0;        // frames of the call in flight
    size_t out_cap             = 0;
    bool pending               = false;
};

struct timg_hip_scaler {
    timg_hip_ctx *ctx = nullptr;
    timg_amd::ResamplePlan plan;
    timg_amd::DevPlan dev{};
    void *tables      = nullptr;  // one allocation holding every table
    int forced_kernel = 0;
    bool streaming_ok = false;
    // streaming-kernel schedule (see scale_stream.hip)
    void *stream_tables = nullptr;
    int stream_cfg[8]   = {0};
};

// Shared by the canvas entry points.
timg_amd::DevBlend MakeDevBlend(const timg_hip_blend *b);

// Encoded frames (device slots of out_cap bytes, out_len[i] bytes used) into the caller's HOST slots.  The caller's
// memory is pageable and usually fresh (twins: a new char[] per Send): copied to directly, the runtime stages and
// page-faults its way through it at ~1 GB/s (a 64-frame sixel batch: 24 ms for 22 MB).  So batches travel as ONE
// strided copy into the context's pinned buffer (the longest frame's length from every slot -- per-copy latency, not
// PCIe, is what many small copies pay) and are distributed from there with memcpy.  The caller holds ctx->mu.
// Synchronises `st`.
int CopyFramesToHost(timg_hip_ctx *ctx, char *out, size_t out_cap, const char *dout, const size_t *out_len,
                     int n_frames, hipStream_t st);

#define TIMG_HIP_TRY(ctx, expr)                                  \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return (ctx)->FailHip(_e, #expr); \
    } while (0)

#endif
