// timg_amd/csrc/synth.hip -- the synthetic RGBA8 frames of the measurement plan (SURVEY.md §8d:
// S-noise, S-photo, S-alpha), generated where they are consumed: in device memory.
//
// Every byte is a pure function of (kind, seed, frame, x, y) built from a counter-based integer
// hash and integer arithmetic only, so the host twin (timg_amd/synth.py: hash_frame) produces the
// same bytes with numpy and the parity tests run on exactly the frames the benchmark times.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "context.h"

namespace timg_amd {
namespace {

__host__ __device__ inline uint32_t Mix(uint32_t x) {  // "lowbias32"
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ inline uint32_t FrameKey(uint32_t seed, uint32_t frame) {
    return Mix(seed * 0x9e3779b9u + frame * 0x85ebca6bu + 0x71170000u);
}

__host__ __device__ inline uint32_t PixelHash(uint32_t key, uint32_t x, uint32_t y, uint32_t k) {
    return Mix(Mix(key ^ (y * 0xc2b2ae35u)) + x * 0x27d4eb2fu + k * 0x165667b1u);
}

struct Wave {  // one triangle wave: phase = x * fx + y * fy + ph (mod 2^16)
    uint32_t fx, fy, ph;
};
struct SynthParams {
    int kind, w, h, border;
    uint32_t key;
    Wave wave[3][3];  // [channel][term]
    unsigned long long d2_max;  // alpha ramp: squared distance (in units of w*h/2) where alpha reaches 0
};

__host__ __device__ inline uint32_t Tri(uint32_t phase) {  // 0 .. 32767
    const uint32_t t = phase & 0xffffu;
    return t < 32768u ? t : 65535u - t;
}

__host__ __device__ inline uint32_t PhotoChannel(const SynthParams &p, int c, uint32_t x, uint32_t y, uint32_t hash) {
    uint32_t s = 0;
    for (int t = 0; t < 3; ++t) s += Tri(x * p.wave[c][t].fx + y * p.wave[c][t].fy + p.wave[c][t].ph);
    int v = (int)(s * 255u / 98301u);
    v += (int)((hash >> (8 * c)) & 15u) + (int)((hash >> (8 * c + 4)) & 15u) - 15;  // ~2.5 % noise
    return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

__host__ __device__ inline uint32_t SynthPixel(const SynthParams &p, uint32_t x, uint32_t y) {
    const uint32_t h0 = PixelHash(p.key, x, y, 0);
    if (p.kind == TIMG_HIP_SYNTH_NOISE) return h0;
    uint32_t px = PhotoChannel(p, 0, x, y, h0) | (PhotoChannel(p, 1, x, y, h0) << 8) |
                  (PhotoChannel(p, 2, x, y, h0) << 16);
    uint32_t a = 255;
    if (p.kind == TIMG_HIP_SYNTH_ALPHA) {
        const long long dx = 2 * (long long)x - p.w + 1, dy = 2 * (long long)y - p.h + 1;
        const unsigned long long d2 = (unsigned long long)(dx * dx) * p.h * p.h + (unsigned long long)(dy * dy) * p.w * p.w;
        a = d2 >= p.d2_max ? 0u : 255u - (uint32_t)(d2 * 255ull / p.d2_max);
        if ((int)x < p.border || (int)y < p.border || (int)x >= p.w - p.border || (int)y >= p.h - p.border) a = 0;
        const uint32_t h1 = PixelHash(p.key, x, y, 1);
        if (h1 % 10u == 0u) {
            const uint32_t special[4] = {0u, 0x5fu, 0x60u, 0xffu};
            a = special[(h1 >> 8) & 3u];
        }
    }
    return px | (a << 24);
}

SynthParams MakeParams(int kind, int w, int h, uint32_t seed, uint32_t frame) {
    SynthParams p;
    p.kind   = kind;
    p.w      = w;
    p.h      = h;
    p.key    = FrameKey(seed, frame);
    p.border = kind == TIMG_HIP_SYNTH_ALPHA ? (64 < (w < h ? w : h) / 8 ? 64 : ((w < h ? w : h) / 8 > 1 ? (w < h ? w : h) / 8 : 1)) : 0;
    for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 3; ++t) {
            const uint32_t a = Mix(p.key + 0x1000u + (uint32_t)(c * 3 + t) * 3u);
            const uint32_t b = Mix(p.key + 0x1001u + (uint32_t)(c * 3 + t) * 3u);
            const uint32_t d = Mix(p.key + 0x1002u + (uint32_t)(c * 3 + t) * 3u);
            // 0.5 .. 6 cycles over the frame, in 1/256 cycles
            p.wave[c][t].fx = (uint32_t)(((unsigned long long)(128u + a % 1409u) * 65536ull / 256ull) / (unsigned)(w > 0 ? w : 1));
            p.wave[c][t].fy = (uint32_t)(((unsigned long long)(128u + b % 1409u) * 65536ull / 256ull) / (unsigned)(h > 0 ? h : 1));
            p.wave[c][t].ph = d & 0xffffu;
        }
    // alpha reaches 0 at 1.2 x the half-diagonal-ish radius: d2 is in units of (w*h/2)^2
    p.d2_max = (unsigned long long)w * w * (unsigned long long)h * h * 36ull / 25ull;
    if (p.d2_max == 0) p.d2_max = 1;
    return p;
}

__global__ void __launch_bounds__(256) SynthKernel(SynthParams p, uint8_t *dst, size_t stride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= p.w) return;
    *reinterpret_cast<uint32_t *>(dst + (size_t)y * stride + (size_t)x * 4) = SynthPixel(p, (uint32_t)x, (uint32_t)y);
}

}  // namespace
}  // namespace timg_amd

using namespace timg_amd;

int timg_hip_synth_frames(timg_hip_ctx *ctx, int kind, int w, int h, uint32_t seed, int first_frame, int n_frames,
                          uint8_t *dst, size_t frame_stride, int dst_on_device, void *stream) {
    if (!ctx || !dst || w <= 0 || h <= 0 || n_frames <= 0 || kind < TIMG_HIP_SYNTH_NOISE || kind > TIMG_HIP_SYNTH_ALPHA)
        return TIMG_HIP_ERR_ARG;
    if ((long long)w * h > (1ll << 30)) return ctx->Fail(TIMG_HIP_ERR_ARG, "frame too large");
    if (frame_stride == 0) frame_stride = (size_t)w * h * 4;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
    uint8_t *d = dst;
    const size_t total = frame_stride * (size_t)(n_frames - 1) + (size_t)w * h * 4;
    if (!dst_on_device) {
        lock.lock();
        TIMG_HIP_TRY(ctx, ctx->dev[1].Reserve(total));
        d = (uint8_t *)ctx->dev[1].ptr;
    }
    for (int f = 0; f < n_frames; ++f) {
        const SynthParams p = MakeParams(kind, w, h, seed, (uint32_t)(first_frame + f));
        hipLaunchKernelGGL(SynthKernel, dim3((w + 255) / 256, h), dim3(256), 0, st, p, d + (size_t)f * frame_stride,
                           (size_t)w * 4);
    }
    TIMG_HIP_TRY(ctx, hipGetLastError());
    if (!dst_on_device) {
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(dst, d, total, hipMemcpyDeviceToHost, st));
        TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    return TIMG_HIP_OK;
}
