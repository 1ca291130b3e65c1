// timg_amd/csrc/scale_kernels.hip -- scale (+ fused alpha-compose) kernels for
// gfx950.  Device twin of timg::ImageScaler::Scale (src/image-scaler.cc:83-92)
// and Framebuffer::AlphaComposeBackground (src/framebuffer.cc:108-150).
//
// Arithmetic contract (what makes the bytes equal to the CPU reference):
//   * a source pixel becomes 7 floats (pixel_math.h DecodePx);
//   * the vertical pass is one multiply-add chain per channel in increasing
//     input-row order (stb_image_resize2.h:10036-10180 / 9864-10034);
//   * the horizontal pass keeps two chains -- taps at even / odd positions of
//     the column's window -- and adds them at the end, or one chain when the
//     window is <=3 wide (stb_image_resize2.h:5722-5793, 5621-5660);
//   * pass order per plan (vertical_first);
//   * no FMA contraction anywhere (-ffp-contract=off).
#include "device_plan.h"
#include "pixel_math.h"

namespace timg_amd {
namespace {

__device__ __forceinline__ uint32_t FinishPixel(const Px7 &acc, int x, int y,
                                                const DevBlend &blend,
                                                int *transparent_flag) {
    uint32_t out = EncodePx(acc);
    if ((out >> 24) != 0xffu && y >= blend.start_row) {
        if (transparent_flag) *transparent_flag = 1;  // benign race: all write 1
        if (blend.enabled) {
            const bool alt = CheckerAlt(blend, x, y);
            out            = BlendOver(out, alt ? blend.pat : blend.bg);
        }
    }
    return out;
}

// ---- generic: one thread per output pixel ---------------------------------
__global__ void __launch_bounds__(256)
ScaleGenericKernel(DevPlan plan, DevBlend blend, FrameBatch batch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int f = blockIdx.z;
    if (x >= plan.out_w || y >= plan.out_h) return;
    const uint8_t *src = batch.src + (size_t)f * batch.src_frame_stride;
    const int2 ht      = plan.h_taps[x];
    const int2 vr      = plan.v_runs[y];
    const float *hc    = plan.h_coeff + (size_t)x * plan.h_width;
    const int *vrow    = plan.v_rows + vr.x;
    const float *vc    = plan.v_coeff + vr.x;

    Px7 even, odd;
#pragma unroll
    for (int c = 0; c < 7; ++c) even.c[c] = odd.c[c] = 0.0f;

    if (plan.vertical_first) {
        // out = H( V(column) ): build each tapped column's vertical sum, then
        // feed it to the horizontal chains.
        for (int k = 0; k < ht.y; ++k) {
            const int xin = ht.x + k;
            Px7 col;
            for (int j = 0; j < vr.y; ++j) {
                const uint32_t px = *reinterpret_cast<const uint32_t *>(
                    src + (size_t)vrow[j] * batch.src_stride + (size_t)xin * 4);
                const Px7 d   = DecodePx(px, plan.swap_rb);
                const float w = vc[j];
                if (j == 0) {
#pragma unroll
                    for (int c = 0; c < 7; ++c) col.c[c] = d.c[c] * w;
                } else {
#pragma unroll
                    for (int c = 0; c < 7; ++c) col.c[c] = col.c[c] + d.c[c] * w;
                }
            }
            const float w = hc[k];
            const bool to_odd = !plan.h_sequential && (k & 1);
            if (k < 2 && (k == 0 || to_odd)) {
                Px7 &dst = to_odd ? odd : even;
#pragma unroll
                for (int c = 0; c < 7; ++c) dst.c[c] = col.c[c] * w;
            } else if (to_odd) {
#pragma unroll
                for (int c = 0; c < 7; ++c) odd.c[c] = odd.c[c] + col.c[c] * w;
            } else {
#pragma unroll
                for (int c = 0; c < 7; ++c) even.c[c] = even.c[c] + col.c[c] * w;
            }
        }
        if (!plan.h_sequential) {
#pragma unroll
            for (int c = 0; c < 7; ++c) even.c[c] = even.c[c] + odd.c[c];
        }
    } else {
        // out = V( H(row) )
        Px7 acc;
        for (int j = 0; j < vr.y; ++j) {
            const uint8_t *row = src + (size_t)vrow[j] * batch.src_stride;
            Px7 e, o;
#pragma unroll
            for (int c = 0; c < 7; ++c) e.c[c] = o.c[c] = 0.0f;
            for (int k = 0; k < ht.y; ++k) {
                const uint32_t px =
                    *reinterpret_cast<const uint32_t *>(row + (size_t)(ht.x + k) * 4);
                const Px7 d       = DecodePx(px, plan.swap_rb);
                const float w     = hc[k];
                const bool to_odd = !plan.h_sequential && (k & 1);
                if (k < 2 && (k == 0 || to_odd)) {
                    Px7 &dst = to_odd ? o : e;
#pragma unroll
                    for (int c = 0; c < 7; ++c) dst.c[c] = d.c[c] * w;
                } else if (to_odd) {
#pragma unroll
                    for (int c = 0; c < 7; ++c) o.c[c] = o.c[c] + d.c[c] * w;
                } else {
#pragma unroll
                    for (int c = 0; c < 7; ++c) e.c[c] = e.c[c] + d.c[c] * w;
                }
            }
            if (!plan.h_sequential) {
#pragma unroll
                for (int c = 0; c < 7; ++c) e.c[c] = e.c[c] + o.c[c];
            }
            const float w = vc[j];
            if (j == 0) {
#pragma unroll
                for (int c = 0; c < 7; ++c) acc.c[c] = e.c[c] * w;
            } else {
#pragma unroll
                for (int c = 0; c < 7; ++c) acc.c[c] = acc.c[c] + e.c[c] * w;
            }
        }
        even = acc;
    }
    int *flag = batch.transparent_flags ? batch.transparent_flags + f : nullptr;
    const uint32_t out = FinishPixel(even, x, y, blend, flag);
    *reinterpret_cast<uint32_t *>(batch.dst + (size_t)f * batch.dst_frame_stride +
                                  (size_t)y * batch.dst_stride + (size_t)x * 4) = out;
}

// ---- identity (1:1): copy + swizzle + blend ---------------------------------
__global__ void __launch_bounds__(256)
CopyBlendKernel(DevPlan plan, DevBlend blend, FrameBatch batch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int f = blockIdx.z;
    if (x >= plan.out_w || y >= plan.out_h) return;
    uint32_t px = *reinterpret_cast<const uint32_t *>(
        batch.src + (size_t)f * batch.src_frame_stride + (size_t)y * batch.src_stride +
        (size_t)x * 4);
    if (plan.swap_rb)
        px = (px & 0xff00ff00u) | ((px & 0xffu) << 16) | ((px >> 16) & 0xffu);
    if ((px >> 24) != 0xffu && y >= blend.start_row) {
        if (batch.transparent_flags) batch.transparent_flags[f] = 1;
        if (blend.enabled) {
            const bool alt = CheckerAlt(blend, x, y);
            px             = BlendOver(px, alt ? blend.pat : blend.bg);
        }
    }
    *reinterpret_cast<uint32_t *>(batch.dst + (size_t)f * batch.dst_frame_stride +
                                  (size_t)y * batch.dst_stride + (size_t)x * 4) = px;
}

// ---- standalone AlphaComposeBackground ---------------------------------------
__global__ void __launch_bounds__(256)
AlphaComposeKernel(uint8_t *fb, int w, int h, size_t stride, size_t frame_stride,
                   DevBlend blend, int *flags) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y + blend.start_row;
    const int f = blockIdx.z;
    if (x >= w || y >= h) return;
    uint32_t *p = reinterpret_cast<uint32_t *>(fb + (size_t)f * frame_stride +
                                               (size_t)y * stride + (size_t)x * 4);
    const uint32_t px = *p;
    if ((px >> 24) == 0xffu) return;
    if (flags) flags[f] = 1;
    if (!blend.enabled) return;
    const bool alt = CheckerAlt(blend, x, y);
    *p             = BlendOver(px, alt ? blend.pat : blend.bg);
}

}  // namespace

hipError_t LaunchScaleGeneric(const DevPlan &plan, const DevBlend &blend,
                              const FrameBatch &batch, hipStream_t stream) {
    const dim3 block(64, 4, 1);
    const dim3 grid((plan.out_w + 63) / 64, (plan.out_h + 3) / 4, batch.n_frames);
    hipLaunchKernelGGL(ScaleGenericKernel, grid, block, 0, stream, plan, blend, batch);
    return hipGetLastError();
}

hipError_t LaunchCopyBlend(const DevPlan &plan, const DevBlend &blend,
                           const FrameBatch &batch, hipStream_t stream) {
    const dim3 block(64, 4, 1);
    const dim3 grid((plan.out_w + 63) / 64, (plan.out_h + 3) / 4, batch.n_frames);
    hipLaunchKernelGGL(CopyBlendKernel, grid, block, 0, stream, plan, blend, batch);
    return hipGetLastError();
}

hipError_t LaunchAlphaCompose(uint8_t *fb, int w, int h, size_t stride,
                              size_t frame_stride, int n_frames,
                              const DevBlend &blend, int *transparent_flags,
                              hipStream_t stream) {
    const int rows = h - blend.start_row;
    if (rows <= 0 || w <= 0 || n_frames <= 0) return hipSuccess;
    const dim3 block(64, 4, 1);
    const dim3 grid((w + 63) / 64, (rows + 3) / 4, n_frames);
    hipLaunchKernelGGL(AlphaComposeKernel, grid, block, 0, stream, fb, w, h, stride,
                       frame_stride, blend, transparent_flags);
    return hipGetLastError();
}

}  // namespace timg_amd
