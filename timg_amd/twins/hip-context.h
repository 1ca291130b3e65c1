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

// GPU selection: TIMG_HIP_DEVICE=<n> (default 0), TIMG_HIP=0 disables.
bool HipTwinsEnabled();

// A device call failed after the GPU back-end had been selected: print
// timg_hip_last_error() and terminate.  The twins never substitute CPU results
// for a failed device call -- nothing in Scale()/Send() can fail in the
// reference either, so there is no error path to return through.
[[noreturn]] void HipFatal(timg_hip_ctx *ctx, const char *what);

}  // namespace timg
#endif
