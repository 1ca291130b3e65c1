// timg_amd/csrc/pixel_math.h -- device-side per-pixel arithmetic shared by the
// scale, blend and canvas kernels.  Everything here must round exactly like
// the reference's CPU code, so this translation unit is built with
// -ffp-contract=off (no FMA contraction) and relies on hipcc's default
// correctly-rounded fp32 divide and sqrt.
#ifndef TIMG_AMD_PIXEL_MATH_H
#define TIMG_AMD_PIXEL_MATH_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace timg_amd {

// 2^-120: stb's "small float" (stb_image_resize2.h:1104)
#define TIMG_TINY_F32 0x1p-120f

// A decoded, alpha-weighted source pixel as stb keeps it internally:
// R G B A R*A G*A B*A, each channel u8/255 (stb_image_resize2.h:8300-8321,
// 4081-4175).
struct Px7 {
    float c[7];
};

__device__ __forceinline__ Px7 DecodePx(uint32_t px, int swap_rb) {
    const float k = 1.0f / 255.0f;
    float r       = (float)(px & 0xffu) * k;
    const float g = (float)((px >> 8) & 0xffu) * k;
    float b       = (float)((px >> 16) & 0xffu) * k;
    const float a = (float)(px >> 24) * k;
    if (swap_rb) {
        const float t = r;
        r             = b;
        b             = t;
    }
    Px7 o;
    o.c[0] = r;
    o.c[1] = g;
    o.c[2] = b;
    o.c[3] = a;
    o.c[4] = r * a;
    o.c[5] = g * a;
    o.c[6] = b * a;
    return o;
}

__device__ __forceinline__ uint32_t ToByte(float v) {
    // stb_image_resize2.h:8415-8432 / 1391-1403: v*255 + 0.5, clamp, truncate
    // (one v_med3_f32; the operands are never NaN: sums of finite products)
    const float f = __builtin_amdgcn_fmed3f(v * 255.0f + 0.5f, 0.0f, 255.0f);
    return (uint32_t)f;
}

// Undo the alpha weighting (stb_image_resize2.h:4247-4292) and quantise.
__device__ __forceinline__ uint32_t EncodePx(const Px7 &p) {
    const float alpha = p.c[3];
    float r, g, b;
    if (alpha < TIMG_TINY_F32) {
        r = p.c[0];
        g = p.c[1];
        b = p.c[2];
    } else {
        const float inv = 1.0f / alpha;
        r               = p.c[4] * inv;
        g               = p.c[5] * inv;
        b               = p.c[6] * inv;
    }
    return ToByte(r) | (ToByte(g) << 8) | (ToByte(b) << 16) | (ToByte(alpha) << 24);
}

// timg::LinearColor (src/framebuffer.h:138-174): c^2 as "linear", sqrt back.
__device__ __forceinline__ uint32_t GammaByte(float v) {
    const float s = sqrtf(v);
    return s > 255.0f ? 255u : (uint32_t)s;
}

// One channel of LinearColor(px).AlphaBlend(bg).repack(): x = c^2 a + bg^2 (255 - a) is an exact integer
// below 2^24 in fp32 (c, a bytes; bg^2 <= 65025), and the reference then computes
//     byte = trunc(min(255, sqrtf(x / 255.0f)))      -- correctly rounded divide and square root.
// Over ALL 16 581 376 possible x that byte equals floor(sqrt(x / 255)) in exact arithmetic, i.e. the largest
// k with 255 k^2 <= x (tests/test_blend_identity.py enumerates them: the rounding of the quotient and of the
// root never carries a value across an integer).  So no IEEE divide (10 instructions) and no correctly rounded
// sqrt (~20): an approximate root, truncated, is at most one off, and two comparisons against 255 k^2 and
// 255 (k + 1)^2 -- exact in fp32 -- settle it.  ~14 instructions per channel instead of ~45.
__device__ __forceinline__ uint32_t BlendChannelByte(float x) {
    float k = __builtin_truncf(__builtin_amdgcn_sqrtf(x * (1.0f / 255.0f)));
    k       = (k * k) * 255.0f > x ? k - 1.0f : k;                    // (products exact: k <= 256)
    const float k1 = k + 1.0f;
    k       = (k1 * k1) * 255.0f <= x ? k1 : k;
    return (uint32_t)k;                                               // (x <= 255 * 255^2: k <= 255)
}

// LinearColor(px).AlphaBlend(bg).repack() for a pixel whose alpha != 255
// (src/framebuffer.h:155-161, src/framebuffer.cc:126-131).
__device__ __forceinline__ uint32_t BlendOver(uint32_t px, const float bg[3]) {
    const uint32_t r8 = px & 0xffu, g8 = (px >> 8) & 0xffu, b8 = (px >> 16) & 0xffu;
    const float a  = (float)(px >> 24);
    const float na = 255.0f - a;
    const float r  = (float)(r8 * r8) * a + bg[0] * na;  // (exact integers: see BlendChannelByte)
    const float g  = (float)(g8 * g8) * a + bg[1] * na;
    const float b  = (float)(b8 * b8) * a + bg[2] * na;
    return BlendChannelByte(r) | (BlendChannelByte(g) << 8) | (BlendChannelByte(b) << 16) | 0xff000000u;
}

}  // namespace timg_amd
#endif
