// timg_amd/csrc/sixel_canvas.hip -- placeholder until the sixel kernels land.
#include "context.h"

static size_t Round6(int h) { return (size_t)((h + 5) - (h + 5) % 6); }

extern "C" size_t timg_hip_sixel_max_bytes(int w, int h) {
    return 1024 + (size_t)w * Round6(h) * 5;  // src/sixel-canvas.cc:123
}

extern "C" int timg_hip_sixel_encode(timg_hip_ctx *ctx, const uint8_t *, int, int, int, size_t,
                                     int, int, int, const timg_hip_blend *, char *, size_t, int,
                                     size_t *, void *) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "sixel encode not built yet");
}
