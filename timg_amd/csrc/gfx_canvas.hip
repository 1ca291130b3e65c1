// timg_amd/csrc/gfx_canvas.hip -- graphics-protocol canvases at --compress=0:
// png::Encode (src/timg-png.cc:91-153) over zlib stored blocks, base64 (src/timg-base64.h) and
// the framing of KittyGraphicsCanvas::Send (src/kitty-canvas.cc:167-214, no tmux) and
// ITerm2GraphicsCanvas::Send (src/iterm2-canvas.cc:52-71).
//
// With stored blocks every output byte has a position that is a pure function of its index
// (gfx_layout.h); what is not positional are the two checksums:
//   Adler-32  = plain sums  A' = sum d_j,  B' = sum (n - j) d_j  over the filtered bytes
//               (64-bit, per workgroup in LDS, one global atomic per workgroup), reduced mod 65521
//               once at the end;
//   CRC-32    = per-lane CRCs of 512-byte chunks, combined left to right in two levels with
//               crc(A || B) = x^(8|B|) crc(A) + crc(B) over GF(2) (the shift factors for the four
//               lengths that occur are computed on the host).
// HBM-bound byte work by nature (measured: profiles/r2/bench_modes.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "context.h"
#include "gfx_layout.h"

namespace timg_amd {
namespace {

struct GfxBatch {
    const uint8_t *fb;
    size_t stride, frame_stride;
    uint8_t *png;  // [frames] slots of png_stride bytes
    size_t png_stride;
    unsigned long long *sums;  // [frames][2]
    uint32_t *chunk_crc;       // [frames][n_chunks]
    uint32_t *segment_crc;     // [frames][n_segments]
    uint8_t *out;              // [frames] slots of out_cap bytes (framed kinds)
    size_t out_cap;
    const uint8_t *headers;    // [frames][kGfxHeaderCap]: header text, its length in the last byte
};

// pass 1: four pixels of a row per lane (PngBodyGroup, gfx_layout.h); the file's fixed head and the
// block headers on the side.  The Adler sums: per lane from byte sums and dot products, 64-bit wave
// reduction, one LDS atomic per wave, one pair of global atomics per workgroup.
// (First version: one BYTE per lane and two 64-bit LDS atomics per byte -- 3.1 ms per 64 frames of
// 800x450, 25x the scale kernel's time for a twentieth of its bytes.)
__device__ __forceinline__ unsigned long long WaveSumU64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256) PngBodyKernel(PngGeom g, GfxBatch b) {
    __shared__ unsigned long long s_a, s_b;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const int f      = blockIdx.y;
    uint8_t *png     = b.png + (size_t)f * b.png_stride;
    if (threadIdx.x == 0) s_a = s_b = 0;
    __syncthreads();
    uint32_t a = 0;
    unsigned long long bs = 0;
    if (j < PngBodyGroups(g)) {
        const uint32_t gpr = ((uint32_t)g.w + 3u) >> 2, y = j / gpr;
        PngBodyGroup(b.fb + (size_t)f * b.frame_stride, b.stride, g, y, j - y * gpr, png, &a, &bs);
    }
    if (j < kPngIdatData + 2) png[j] = g.head[j];
    if (j < g.n_blocks) {
        uint8_t hdr[5];
        PngBlockHeader(g, j, hdr);
        uint8_t *p = png + PngBlockHeaderOffset(j);
        for (int i = 0; i < 5; ++i) p[i] = hdr[i];
    }
    const unsigned long long wa = WaveSumU64(a), wb = WaveSumU64(bs);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_a, wa);
        atomicAdd(&s_b, wb);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&b.sums[2 * f], s_a);
        atomicAdd(&b.sums[2 * f + 1], s_b);
    }
}

// pass 2: the Adler-32 closes the zlib stream (one lane per frame)
__global__ void PngAdlerKernel(PngGeom g, GfxBatch b, int n_frames) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    PutBE32(b.png + (size_t)f * b.png_stride + PngAdlerOffset(g), AdlerFromSums(g, b.sums[2 * f], b.sums[2 * f + 1]));
}

// pass 3: CRC of every 512-byte chunk of "IDAT" + stream, a lane per chunk, a dword per step through four
// 256-entry tables in LDS (slicing by four; the tables are built by the workgroup itself: T0 bit by bit,
// Tk[i] = (Tk-1[i] >> 8) ^ T0[Tk-1[i] & 255]).  The checksummed region starts 37 bytes into the file:
// the dwords are unaligned loads.  Bit-by-bit over byte loads this pass took as long as the scale kernel
// takes for a fourth of the batch.
__global__ void __launch_bounds__(256) PngChunkCrcKernel(PngGeom g, GfxBatch b) {
    __shared__ uint32_t tab[4][256];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        tab[0][threadIdx.x] = c;
    }
    __syncthreads();
    {
        const uint32_t t0 = tab[0][threadIdx.x];
        const uint32_t t1 = (t0 >> 8) ^ tab[0][t0 & 255u];
        const uint32_t t2 = (t1 >> 8) ^ tab[0][t1 & 255u];
        const uint32_t t3 = (t2 >> 8) ^ tab[0][t2 & 255u];
        tab[1][threadIdx.x] = t1;
        tab[2][threadIdx.x] = t2;
        tab[3][threadIdx.x] = t3;
    }
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const int f      = blockIdx.y;
    if (c >= g.n_chunks) return;
    const uint32_t at = c * kCrcChunk, left = g.crc_len - at, n = left < kCrcChunk ? left : kCrcChunk;
    const uint8_t *p  = b.png + (size_t)f * b.png_stride + PngCrcRegion() + at;
    uint32_t crc      = 0xffffffffu;
    uint32_t i        = 0;
    for (; i + 4 <= n; i += 4) {
        crc ^= GfxLoadU32(p + i);
        crc = tab[3][crc & 255u] ^ tab[2][(crc >> 8) & 255u] ^ tab[1][(crc >> 16) & 255u] ^ tab[0][crc >> 24];
    }
    for (; i < n; ++i) crc = (crc >> 8) ^ tab[0][(crc ^ p[i]) & 255u];
    b.chunk_crc[(size_t)f * g.n_chunks + c] = ~crc;
}

// pass 4a: 64 chunk CRCs -> one segment CRC
__global__ void PngSegmentCrcKernel(PngGeom g, GfxBatch b) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const int f      = blockIdx.y;
    if (s >= g.n_segments) return;
    b.segment_crc[(size_t)f * g.n_segments + s] = PngSegmentCrc(b.chunk_crc + (size_t)f * g.n_chunks, g, s);
}

// pass 4b: segment CRCs -> IDAT's CRC; IEND behind it (one lane per frame)
__global__ void PngTailKernel(PngGeom g, GfxBatch b, int n_frames) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    PngTail(b.png + (size_t)f * b.png_stride, g, PngTotalCrc(b.segment_crc + (size_t)f * g.n_segments, g));
}

// pass 5: FOUR groups of three PNG bytes per lane -> sixteen base64 characters in their place
// (GfxFrameQuad, gfx_layout.h); header, kitty chunk separators and trailer on the side
__global__ void __launch_bounds__(256) GfxFrameKernel(PngGeom g, GfxBatch b, int kind) {
    const uint32_t quad = blockIdx.x * 256u + threadIdx.x;
    const int f         = blockIdx.y;
    const uint8_t *png  = b.png + (size_t)f * b.png_stride;
    uint8_t *out        = b.out + (size_t)f * b.out_cap;
    const uint8_t *head = b.headers + (size_t)f * kGfxHeaderCap;
    const GfxFraming fr = MakeFraming(kind, g, head[kGfxHeaderCap - 1]);
    GfxFrameQuad(png, g, fr, quad, out);
    if (quad < fr.header_len) out[quad] = head[quad];
    if (kind == kGfxKitty && quad >= 1 && quad < fr.n_kitty_chunks) {
        uint8_t sep[13];
        KittySeparator(fr, quad, sep);
        uint8_t *p = out + KittySeparatorOffset(fr, quad);
        for (int i = 0; i < 13; ++i) p[i] = sep[i];
    }
    if (quad == 0) GfxTrailer(fr, out);
}

}  // namespace
}  // namespace timg_amd

using namespace timg_amd;

extern "C" size_t timg_hip_gfx_max_bytes(int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    const PngGeom g = MakePngGeom(w, h, true);
    return kGfxHeaderCap + 4 * (size_t)((g.png_n + 2) / 3) + ((size_t)g.png_n / kKittyChunk + 1) * kKittySepLen + 8;
}

extern "C" size_t timg_hip_png_bytes(int w, int h, int flags) {
    if (w <= 0 || h <= 0) return 0;
    return MakePngGeom(w, h, !(flags & TIMG_HIP_GFX_RGB24)).png_n;
}

static int GfxEncode(int kind, timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride, size_t frame_stride,
                     int fb_on_device, int n_frames, int flags, const uint32_t *image_ids, char *out,
                     size_t out_cap, int out_on_device, size_t *out_len, void *stream) {
    if (!ctx || !fb || !out || !out_len || w <= 0 || h <= 0 || n_frames <= 0) return TIMG_HIP_ERR_ARG;
    if (kind == kGfxKitty && !image_ids) return TIMG_HIP_ERR_ARG;
    // 64 Mpixels: the Adler-32 B sum (sum over (n - j) * byte, at most 255 n^2 / 2 for n raw bytes)
    // stays below 2^64, so it can be reduced mod 65521 once at the end; 32-bit byte offsets hold too
    if ((unsigned long long)w * h > 64ull * 1024 * 1024)
        return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "frame of %d x %d pixels is too large", w, h);
    if (stride == 0) stride = w * 4;
    if (stride < w * 4) return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride");
    if (frame_stride == 0) frame_stride = (size_t)stride * h;
    const PngGeom g = MakePngGeom(w, h, !(flags & TIMG_HIP_GFX_RGB24));
    // per-frame header text (the image id is the caller's: src/kitty-canvas.cc:47-52 derives it from time())
    std::vector<uint8_t> headers((size_t)n_frames * kGfxHeaderCap, 0);
    size_t worst = g.png_n;
    for (int i = 0; i < n_frames; ++i) {
        uint32_t len = 0;
        if (kind != kGfxPng)
            len = FormatGfxHeader(kind, g, image_ids ? image_ids[i] : 0u, (char *)&headers[(size_t)i * kGfxHeaderCap]);
        if (len >= kGfxHeaderCap - 1) return ctx->Fail(TIMG_HIP_ERR_ARG, "header text too long");
        headers[(size_t)i * kGfxHeaderCap + kGfxHeaderCap - 1] = (uint8_t)len;
        out_len[i] = kind == kGfxPng ? g.png_n : MakeFraming(kind, g, len).total;
        if (out_len[i] > worst) worst = out_len[i];
    }
    if (worst > out_cap) return ctx->Fail(TIMG_HIP_ERR_SMALL, "frame needs %zu bytes, out_cap is %zu", worst, out_cap);

    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::lock_guard<std::mutex> lock(ctx->mu);
    const size_t nf       = (size_t)n_frames;
    const size_t fb_bytes = frame_stride * (nf - 1) + (size_t)stride * h;
    const uint8_t *dfb    = fb;
    if (!fb_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(fb_bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, fb, fb_bytes, hipMemcpyHostToDevice, st));
        dfb = (const uint8_t *)ctx->dev[0].ptr;
    }
    char *dout = out;
    if (!out_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[1].Reserve(out_cap * nf));
        dout = (char *)ctx->dev[1].ptr;
    }
    // scratch: PNG slots (framed kinds), sums, chunk and segment CRCs, header texts
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t png_slot = align(g.png_n);
    size_t off            = 0;
    auto carve            = [&](size_t bytes) {
        const size_t at = off;
        off             = align(off + bytes);
        return at;
    };
    const size_t o_png  = carve(kind == kGfxPng ? 0 : nf * png_slot);
    const size_t o_sums = carve(nf * 2 * sizeof(unsigned long long));
    const size_t o_ccrc = carve(nf * g.n_chunks * sizeof(uint32_t));
    const size_t o_scrc = carve(nf * g.n_segments * sizeof(uint32_t));
    const size_t o_head = carve(nf * kGfxHeaderCap);
    TIMG_HIP_TRY(ctx, ctx->dev[7].Reserve(off));
    char *base = (char *)ctx->dev[7].ptr;
    GfxBatch b;
    b.fb           = dfb;
    b.stride       = (size_t)stride;
    b.frame_stride = frame_stride;
    b.png          = kind == kGfxPng ? (uint8_t *)dout : (uint8_t *)(base + o_png);
    b.png_stride   = kind == kGfxPng ? out_cap : png_slot;
    b.sums         = (unsigned long long *)(base + o_sums);
    b.chunk_crc    = (uint32_t *)(base + o_ccrc);
    b.segment_crc  = (uint32_t *)(base + o_scrc);
    b.out          = (uint8_t *)dout;
    b.out_cap      = out_cap;
    b.headers      = (const uint8_t *)(base + o_head);
    TIMG_HIP_TRY(ctx, hipMemsetAsync(b.sums, 0, nf * 2 * sizeof(unsigned long long), st));
    if (kind != kGfxPng) {
        // (`headers` outlives the copy: the stream is synchronised before this function returns)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(base + o_head, headers.data(), headers.size(), hipMemcpyHostToDevice, st));
    }
    const unsigned frames = (unsigned)n_frames;
    // (lanes: a group of four pixels each, and at least one per byte of the fixed head / per block header)
    const uint32_t body_lanes = std::max(std::max(PngBodyGroups(g), kPngIdatData + 2), g.n_blocks);
    hipLaunchKernelGGL(PngBodyKernel, dim3((body_lanes + 255) / 256, frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngAdlerKernel, dim3((frames + 63) / 64), dim3(64), 0, st, g, b, n_frames);
    hipLaunchKernelGGL(PngChunkCrcKernel, dim3((g.n_chunks + 255) / 256, frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngSegmentCrcKernel, dim3((g.n_segments + 63) / 64, frames), dim3(64), 0, st, g, b);
    hipLaunchKernelGGL(PngTailKernel, dim3((frames + 63) / 64), dim3(64), 0, st, g, b, n_frames);
    if (kind != kGfxPng) {
        // (lanes: four base64 groups each, and at least one per header byte / per kitty separator)
        const uint32_t n_quads = ((g.png_n + 2) / 3 + 3) / 4;
        const uint32_t lanes   = std::max(std::max(n_quads, kGfxHeaderCap), (g.png_n + kKittyChunk - 1) / kKittyChunk);
        hipLaunchKernelGGL(GfxFrameKernel, dim3((lanes + 255) / 256, frames), dim3(256), 0, st, g, b, kind);
    }
    TIMG_HIP_TRY(ctx, hipGetLastError());
    if (!out_on_device) {
        for (int i = 0; i < n_frames; ++i)
            TIMG_HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)i * out_cap, dout + (size_t)i * out_cap, out_len[i],
                                             hipMemcpyDeviceToHost, st));
    }
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return TIMG_HIP_OK;
}

extern "C" int timg_hip_png_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                                   size_t frame_stride, int fb_on_device, int n_frames, int flags, char *out,
                                   size_t out_cap, int out_on_device, size_t *out_len, void *stream) {
    return GfxEncode(kGfxPng, ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, nullptr, out,
                     out_cap, out_on_device, out_len, stream);
}

extern "C" int timg_hip_kitty_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                                     size_t frame_stride, int fb_on_device, int n_frames, int flags,
                                     const uint32_t *image_ids, char *out, size_t out_cap, int out_on_device,
                                     size_t *out_len, void *stream) {
    return GfxEncode(kGfxKitty, ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, image_ids, out,
                     out_cap, out_on_device, out_len, stream);
}

extern "C" int timg_hip_iterm2_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                                      size_t frame_stride, int fb_on_device, int n_frames, int flags, char *out,
                                      size_t out_cap, int out_on_device, size_t *out_len, void *stream) {
    return GfxEncode(kGfxIterm2, ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, nullptr, out,
                     out_cap, out_on_device, out_len, stream);
}
