// timg_amd/twins/hip-context.h -- process-wide handle to libtimg_hip.so for the
// GPU-backed twins of timg's renderer classes.  One context per process (the
// reference is a single-process CLI); calls are serialised inside the library
// where they share scratch memory, so loader-pool threads may scale
// concurrently (src/timg.cc:948-968).
#ifndef TIMG_AMD_TWINS_HIP_CONTEXT_H
#define TIMG_AMD_TWINS_HIP_CONTEXT_H

#include "timg_hip.h"

namespace timg {

// Returns the shared context, creating it on first use; nullptr when no HIP
// device is usable -- callers then keep the CPU implementation, the same
// "factory returns null, next one is tried" convention the reference uses
// (src/image-source.cc:162-221, src/stb-image-source.cc:54-56).
timg_hip_ctx *SharedHipContext();
// Further process-wide contexts on the same device (k = 1, 2, ...: own stream, own scratch), created on first
// use and kept: encoders that run several batches at the same time (held-rows.h) put each on one of them.
// nullptr when it cannot be created (callers then make do with fewer).
timg_hip_ctx *ExtraHipContext(int k);

// The context a LOADER thread scales host frames on (HipImageScaler): one of a few contexts with their own streams and
// staging buffers, fixed per thread, so that the uploads of several loaders overlap instead of queueing on the shared
// context's lock.  Falls back to the shared context.
timg_hip_ctx *LoaderHipContext();

// The context whose stream carries the canvases' device-to-device frame copies (a held grid row gathers the frames of
// its Sends back to back, an asynchronous sixel Send keeps a copy: the Framebuffer is only valid during the call) and
// NOTHING else: such a copy is a microsecond of work, and whoever has to know that it has landed -- the worker that
// encodes the row, the source that frees the frame -- waits for a stream that never holds a 0.8 ms encode.  (Until
// round 6 the copies went on the shared context's stream, which orders them against everything else on that stream;
// with the sources loading on contexts of their own that order no longer covers a frame block the pool hands to the
// next loader.)  Falls back to the shared context.
timg_hip_ctx *CopyHipContext();
// Waits until every frame copy enqueued so far has landed (no-op when no copy was ever made).
void HipFrameCopiesDone();

// GPU selection: TIMG_HIP_DEVICE=<n> (default 0), TIMG_HIP=0 disables.
bool HipTwinsEnabled();
// TIMG_HIP_TWIN_TRACE in the environment: the twins say on stderr what they do (the context, every scaler and canvas
// they create, every held batch) -- how tests/test_timg_binary.py knows that a patched timg really ran on the device.
bool HipTwinTrace();

// Which resampling filter the twins ask the device for (timg_hip_scaler_create's `filter`).  A timg build
// scales with ONE back-end, chosen at build time (src/image-scaler.cc:24-35: libswscale's SWS_BILINEAR when
// WITH_TIMG_SWS_RESIZE, else stb_image_resize2); the twin follows the same macro so that a GPU run of a
// given timg build shows what its CPU run shows -- TIMG_HIP_FILTER_TRIANGLE for an swscale build,
// TIMG_HIP_FILTER_STB_DEFAULT otherwise.  TIMG_HIP_FILTER=stb|bilinear in the environment overrides it.
int HipScalerFilter();

// Device memory for frames, recycled: hipMalloc / hipFree cost far more than the kernels that use a frame
// (a 4K frame is scaled in ~20 us), and every image source and every held grid row needs a buffer.
// Blocks are handed out again by exact size; at most kPoolBytes (4 GiB) stay cached, the rest is really freed, and
// everything cached is given back when an allocation fails (HipPoolTrim).
// Thread-safe.  nullptr on failure.
void *HipPoolMalloc(timg_hip_ctx *ctx, size_t bytes);
void HipPoolFree(timg_hip_ctx *ctx, void *ptr);
// Everything the twins cache on the device and nobody is using -- idle blocks, idle scalers -- goes back to the
// driver; returns how many objects that were.  Called when an allocation fails, here and (HipCall) in the library.
size_t HipPoolTrim(timg_hip_ctx *ctx);

// A device call that fails for lack of device memory (TIMG_HIP_ERR_NOMEM: a scratch buffer of the library could not
// grow, a staging copy could not be allocated) is made ONCE more after HipPoolTrim: a transient shortage in the
// middle of a 600-frame stream must not end the process while gigabytes sit in the pool.  Any other failure, and a
// second one, is the caller's (HipFatal).
// (TIMG_HIP_FAIL_CALL=k in the environment: the k-th HipCall of the process fails with TIMG_HIP_ERR_DEVICE without reaching
// the library -- fault injection for the degrade paths, tests/test_twins.py)
bool HipFailInjected();
template <class Call>
int HipCall(timg_hip_ctx *ctx, Call &&call) {
    if (HipFailInjected()) return TIMG_HIP_ERR_DEVICE;
    int rc = call();
    if (rc == TIMG_HIP_ERR_NOMEM) {
        HipPoolTrim(ctx);
        rc = call();
    }
    return rc;
}

// Scalers, recycled by geometry: creating one builds the resampling plan on the host and uploads its tables
// (O(W + H), but hundreds of microseconds) -- a grid of equally sized images needs it once, not per image.
// A scaler is used by one caller at a time (its tile bookkeeping is per scaler): Acquire hands out an idle
// one or creates one, Release returns it.  Thread-safe.  nullptr on failure.
// At most kMaxIdleScalers (24) idle scalers stay cached over ALL geometries, the most recently released ones.
timg_hip_scaler *HipScalerAcquire(timg_hip_ctx *ctx, int in_w, int in_h, int in_fmt, int out_w, int out_h, int filter);
void HipScalerRelease(timg_hip_scaler *s);
size_t HipIdleScalers();  // (for the tests)

// What the twins really did, counted per process: frames produced by a device call and frames produced by the reference's
// own classes (cpu-sibling.h, HipImageScaler::ScaleOnCpu).  A parity test that compares a twin's stream with the
// reference's proves nothing when the twin quietly took the CPU path: tests/twins/twin_check.cc and
// tests/test_timg_binary.py read these -- with TIMG_HIP_TWIN_TRACE set the process prints them on stderr when it ends:
//   "timg_hip twins: frames on the device: scaler N block N sixel N graphics N; on the CPU: scaler N block N sixel N graphics N; degraded D"
enum HipTwinKind { kHipTwinScaler = 0, kHipTwinBlock = 1, kHipTwinSixel = 2, kHipTwinGraphics = 3, kHipTwinKinds = 4 };
struct HipTwinCounts {
    unsigned long device[kHipTwinKinds];
    unsigned long cpu[kHipTwinKinds];
};
void HipCountFrames(HipTwinKind kind, bool on_device, size_t frames = 1);
HipTwinCounts HipTwinStats();

// A device call failed after the GPU back-end had been selected and the twin has a CPU implementation to go on with
// (cpu-sibling.h; HipImageScaler: the reference's own scaler): says so ONCE on stderr and switches the back-end off for
// the rest of the process -- SharedHipContext() returns nullptr from now on, so every later factory call builds the
// reference's classes.  Objects that exist keep their context (they check HipDegraded() themselves).
void HipDegrade(timg_hip_ctx *ctx, const char *what);
// ... unless the call only REFUSED this frame (TIMG_HIP_ERR_UNSUPP: a geometry the kernels do not take): that frame
// alone goes to the reference's class, the device stays selected for every other one, nothing is printed (with
// TIMG_HIP_TWIN_TRACE: one line per refused frame).
void HipDegradeUnless(int rc, timg_hip_ctx *ctx, const char *what);
bool HipDegraded();

// A device call failed after the GPU back-end had been selected and there is nothing to fall back to: print
// timg_hip_last_error() and terminate.  The twins never substitute CPU results
// for a failed device call -- nothing in Scale()/Send() can fail in the
// reference either, so there is no error path to return through.
[[noreturn]] void HipFatal(timg_hip_ctx *ctx, const char *what);

}  // namespace timg
#endif
