// timg_amd/csrc/device_plan.h -- flat, device-resident view of a ResamplePlan
// plus the launch wrappers of the scale(+blend) kernels.
#ifndef TIMG_AMD_DEVICE_PLAN_H
#define TIMG_AMD_DEVICE_PLAN_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace timg_amd {

struct DevPlan {
    int in_w, in_h, out_w, out_h;
    int swap_rb;         // source is b,g,r,a
    int vertical_first;  // pass order
    int h_sequential;    // <=3 horizontal taps: one accumulation chain
    int h_width;         // floats per output column in h_coeff
    const int2 *h_taps;  // {first input column, tap count} per output column
    const float *h_coeff;
    const int2 *v_runs;  // {first index, count} per output row
    const int *v_rows;
    const float *v_coeff;
};

// Fused Framebuffer::AlphaComposeBackground parameters, pre-linearised on the
// host (LinearColor(rgba_t): c*c, src/framebuffer.h:142-143).
struct DevBlend {
    int enabled;     // blend at all (getter present and bg alpha != 0)
    int checker;     // alternate bg / pattern
    int pw, ph;      // checker cell size in pixels
    // ceil(2^32 / pw) (0 for pw == 1): x / pw == mulhi(x, pw_magic) for every x below 2^32 / pw -- one
    // instruction per pixel instead of the ~25 of a division by a run-time divisor
    unsigned pw_magic;
    int start_row;
    float bg[3];     // linear r,g,b of the background
    float pat[3];    // linear r,g,b of the pattern colour
};

#if defined(__HIPCC__)
// (x / pw + y / ph) & 1 of Framebuffer::AlphaComposeBackground's checkerboard (src/framebuffer.cc:135-149);
// x a pixel column (< 65 536), y uniform
__device__ __forceinline__ bool CheckerAlt(const DevBlend &blend, int x, int y) {
    unsigned qx;
    if ((unsigned)x < 65536u && (unsigned)blend.pw < 65536u)  // (x * pw < 2^32: the multiplication is exact)
        qx = blend.pw_magic ? __umulhi((unsigned)x, blend.pw_magic) : (unsigned)x;
    else
        qx = (unsigned)x / (unsigned)blend.pw;  // (frames wider than 65 536 pixels: the long way)
    return blend.checker && ((qx + (unsigned)(y / blend.ph)) & 1u);
}
#endif

struct FrameBatch {
    const uint8_t *src;
    size_t src_stride, src_frame_stride;
    uint8_t *dst;
    size_t dst_stride, dst_frame_stride;
    int n_frames;
    int *transparent_flags;  // per frame, may be null
};

// Always-applicable kernel: one thread per output pixel, taps gathered
// straight from global memory.
hipError_t LaunchScaleGeneric(const DevPlan &plan, const DevBlend &blend,
                              const FrameBatch &batch, hipStream_t stream);

// Plain copy (+ swizzle) for the 1:1 case, with the same fused blend.
hipError_t LaunchCopyBlend(const DevPlan &plan, const DevBlend &blend,
                           const FrameBatch &batch, hipStream_t stream);

// In-place alpha compose on already-scaled frames.
hipError_t LaunchAlphaCompose(uint8_t *fb, int w, int h, size_t stride,
                              size_t frame_stride, int n_frames,
                              const DevBlend &blend, int *transparent_flags,
                              hipStream_t stream);

}  // namespace timg_amd
#endif
