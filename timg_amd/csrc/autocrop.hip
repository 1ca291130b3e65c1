// timg_amd/csrc/autocrop.hip -- bounding-box reduction behind --auto-crop.
//
// The reference delegates this to GraphicsMagick (src/graphics-magick-source.cc:231-241:
// img.crop() by crop_border first, then img.trim(), both BEFORE the image is scaled), which
// is outside the reference tree ("parity unpinned").  This follows the
// published algorithm of trim() at fuzz 0 (GetImageBoundingBox): after removing crop_border
// pixels on each side, the left and top edges are measured against the top-left corner pixel,
// the right edge against the top-right corner, the bottom edge against the bottom-left one.
// One streaming read of the frame (16 B per lane); the crop itself costs nothing -- the
// scaler is created for the cropped size and reads the window through pointer + stride.
#include "context.h"

namespace timg_amd {
namespace {

__global__ void __launch_bounds__(256)
BBoxKernel(const uint8_t *src, int w, int h, size_t stride, size_t frame_stride, int border,
           int *boxes /* per frame: minx, miny, maxx, maxy */) {
    const int f          = blockIdx.z;
    const uint8_t *frame = src + (size_t)f * frame_stride;
    const int x0 = border, y0 = border, x1 = w - border, y1 = h - border;
    const uint32_t tl = *reinterpret_cast<const uint32_t *>(frame + (size_t)y0 * stride + (size_t)x0 * 4);
    const uint32_t tr = *reinterpret_cast<const uint32_t *>(frame + (size_t)y0 * stride + (size_t)(x1 - 1) * 4);
    const uint32_t bl = *reinterpret_cast<const uint32_t *>(frame + (size_t)(y1 - 1) * stride + (size_t)x0 * 4);
    int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -1, maxy = -1;
    // grid-stride over rows; a workgroup row-slice reads 16 B per lane
    for (int y = y0 + blockIdx.y; y < y1; y += gridDim.y) {
        const uint8_t *row = frame + (size_t)y * stride;
        for (int x = x0 + (blockIdx.x * blockDim.x + threadIdx.x) * 4; x < x1;
             x += gridDim.x * blockDim.x * 4) {
            uint32_t px[4];
            const int n = x1 - x < 4 ? x1 - x : 4;
            if (n == 4 && ((((uintptr_t)(row + (size_t)x * 4)) & 15) == 0)) {
                const uint4 v = *reinterpret_cast<const uint4 *>(row + (size_t)x * 4);
                px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w;
            } else {
                for (int i = 0; i < n; ++i)
                    px[i] = *reinterpret_cast<const uint32_t *>(row + (size_t)(x + i) * 4);
            }
            for (int i = 0; i < n; ++i) {
                if (px[i] != tl) {
                    minx = min(minx, x + i);
                    miny = min(miny, y);
                }
                if (px[i] != tr) maxx = max(maxx, x + i);
                if (px[i] != bl) maxy = max(maxy, y);
            }
        }
    }
    // wave reduction, then one atomic per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        minx = min(minx, __shfl_xor(minx, d));
        miny = min(miny, __shfl_xor(miny, d));
        maxx = max(maxx, __shfl_xor(maxx, d));
        maxy = max(maxy, __shfl_xor(maxy, d));
    }
    if ((threadIdx.x & 63) == 0) {
        if (minx != 0x7fffffff) atomicMin(&boxes[f * 4 + 0], minx);
        if (miny != 0x7fffffff) atomicMin(&boxes[f * 4 + 1], miny);
        if (maxx >= 0) atomicMax(&boxes[f * 4 + 2], maxx);
        if (maxy >= 0) atomicMax(&boxes[f * 4 + 3], maxy);
    }
}

__global__ void InitBoxes(int *boxes, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        boxes[i * 4 + 0] = 0x7fffffff;
        boxes[i * 4 + 1] = 0x7fffffff;
        boxes[i * 4 + 2] = -1;
        boxes[i * 4 + 3] = -1;
    }
}

}  // namespace
}  // namespace timg_amd

extern "C" int timg_hip_autocrop_bbox(timg_hip_ctx *ctx, const uint8_t *src, int w, int h,
                                      int stride, size_t frame_stride, int on_device,
                                      int n_frames, int crop_border, int *out_xywh,
                                      void *stream) {
    if (!ctx || !src || !out_xywh || w <= 0 || h <= 0 || n_frames <= 0 || crop_border < 0)
        return TIMG_HIP_ERR_ARG;
    if (stride == 0) stride = w * 4;
    if (stride < w * 4 || (stride & 3) || ((uintptr_t)src & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    if (frame_stride == 0) frame_stride = (size_t)stride * h;
    if (2 * crop_border >= w || 2 * crop_border >= h) {
        for (int i = 0; i < n_frames; ++i) {
            out_xywh[i * 4 + 0] = out_xywh[i * 4 + 1] = 0;
            out_xywh[i * 4 + 2] = out_xywh[i * 4 + 3] = 0;
        }
        return TIMG_HIP_OK;
    }
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::lock_guard<std::mutex> lock(ctx->mu);
    const size_t bytes = frame_stride * (size_t)(n_frames - 1) + (size_t)stride * h;
    const uint8_t *d   = src;
    if (!on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, src, bytes, hipMemcpyHostToDevice, st));
        d = (const uint8_t *)ctx->dev[0].ptr;
    }
    TIMG_HIP_TRY(ctx, ctx->dev[2].Reserve(sizeof(int) * 4 * n_frames));
    TIMG_HIP_TRY(ctx, ctx->pin[0].Reserve(sizeof(int) * 4 * n_frames));
    int *boxes = (int *)ctx->dev[2].ptr;
    hipLaunchKernelGGL(timg_amd::InitBoxes, dim3((n_frames + 63) / 64), dim3(64), 0, st, boxes,
                       n_frames);
    const int gx = (w / 4 + 255) / 256 > 0 ? (w / 4 + 255) / 256 : 1;
    const int gy = h < 512 ? h : 512;
    hipLaunchKernelGGL(timg_amd::BBoxKernel, dim3(gx, gy, n_frames), dim3(256), 0, st, d, w, h,
                       (size_t)stride, frame_stride, crop_border, boxes);
    TIMG_HIP_TRY(ctx, hipGetLastError());
    int *hb = (int *)ctx->pin[0].ptr;
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(hb, boxes, sizeof(int) * 4 * n_frames, hipMemcpyDeviceToHost, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < n_frames; ++i) {
        if (hb[i * 4 + 2] < hb[i * 4 + 0] || hb[i * 4 + 3] < hb[i * 4 + 1]) {  // nothing but border
            out_xywh[i * 4 + 0] = out_xywh[i * 4 + 1] = 0;
            out_xywh[i * 4 + 2] = out_xywh[i * 4 + 3] = 0;
        } else {
            out_xywh[i * 4 + 0] = hb[i * 4 + 0];
            out_xywh[i * 4 + 1] = hb[i * 4 + 1];
            out_xywh[i * 4 + 2] = hb[i * 4 + 2] - hb[i * 4 + 0] + 1;
            out_xywh[i * 4 + 3] = hb[i * 4 + 3] - hb[i * 4 + 1] + 1;
        }
    }
    return TIMG_HIP_OK;
}
