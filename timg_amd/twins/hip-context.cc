#include "hip-context.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace timg {

bool HipTwinsEnabled() {
    const char *v = getenv("TIMG_HIP");
    return !(v && v[0] == '0');
}

timg_hip_ctx *SharedHipContext() {
    static std::once_flag once;
    static timg_hip_ctx *ctx = nullptr;
    std::call_once(once, []() {
        if (!HipTwinsEnabled()) return;
        const char *d = getenv("TIMG_HIP_DEVICE");
        if (timg_hip_init(d ? atoi(d) : 0, &ctx) != TIMG_HIP_OK) ctx = nullptr;
    });
    return ctx;
}

void HipFatal(timg_hip_ctx *ctx, const char *what) {
    fprintf(stderr, "timg: HIP back-end failed in %s: %s\n", what, timg_hip_last_error(ctx));
    abort();
}

}  // namespace timg
