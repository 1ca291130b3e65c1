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
//   CRC-32    = per-lane CRCs of 512-byte chunks, combined as a binary tree with
//               crc(A || B) = x^(8|B|) crc(A) + crc(B) over GF(2) (the shift factors of every level, for
//               complete right children and for the one that holds the short last chunk, come from the host).
// HBM-bound byte work by nature (measured: profiles/r2/bench_modes.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "context.h"
#include "gfx_layout.h"
#include "wave_ops.h"

namespace timg_amd {
namespace {

struct GfxBatch {
    const uint8_t *fb;
    size_t stride, frame_stride;
    uint8_t *png;  // [frames] slots of png_stride bytes
    size_t png_stride;
    unsigned long long *sums;  // [frames][body workgroups][2]: every workgroup's share of the two Adler sums
    uint32_t sum_slots;        // body workgroups per frame
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
__global__ void __launch_bounds__(256) PngBodyKernel(PngGeom g, GfxBatch b) {
    __shared__ unsigned long long s_a, s_b;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const int f      = blockIdx.y;
    uint8_t *png     = b.png + (size_t)f * b.png_stride;
    if (threadIdx.x == 0) s_a = s_b = 0;
    __syncthreads();
    PngGroupSums r{0u, 0u, 0};
    const bool active = j < PngBodyGroups(g);
    if (active) {
        const uint32_t gpr = ((uint32_t)g.w + 3u) >> 2, y = j / gpr;
        r = PngBodyGroup(b.fb + (size_t)f * b.frame_stride, b.stride, g, y, j - y * gpr, png);
    }
    if (j < kPngIdatData + 2) png[j] = g.head[j];
    if (j < g.n_blocks) {
        uint8_t hdr[5];
        PngBlockHeader(g, j, hdr);
        uint8_t *p = png + PngBlockHeaderOffset(j);
        for (int i = 0; i < 5; ++i) p[i] = hdr[i];
    }
    // B' = sum (raw_n - j0) a - t over the groups.  With J = (first lane's j0) - 1 and j0 = J + delta
    // (1 <= delta <= 64 * 17 inside a wave: the groups are consecutive) a wave's share is
    // (raw_n - J) * sum a  -  sum (delta * a + t): two 32-bit wave sums (delta * a + t < 2^23) instead of
    // 64-bit ones.  (Active lanes are a prefix of the wave; an all-idle wave adds zero.)
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.j0) - 1u;
    const uint32_t wa   = WaveSum(r.a);
    const uint32_t wt   = WaveSum(active ? (r.j0 - base) * r.a + (uint32_t)r.t : 0u);
    if ((threadIdx.x & 63) == 0 && wa != 0) {
        atomicAdd(&s_a, (unsigned long long)wa);
        atomicAdd(&s_b, (unsigned long long)(g.raw_n - base) * wa - wt);
    }
    __syncthreads();
    // (a slot per workgroup, summed by the next pass: global atomics of 22 000 workgroups on the 128 words
    // of the batch's sums -- eight cache lines -- were what this pass waited for)
    if (threadIdx.x == 0) {
        unsigned long long *slot = b.sums + ((size_t)f * b.sum_slots + blockIdx.x) * 2;
        slot[0] = s_a;
        slot[1] = s_b;
    }
}

// pass 2: the workgroups' shares summed (a workgroup per frame); the Adler-32 closes the zlib stream
__device__ __forceinline__ unsigned long long WaveSumU64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__global__ void __launch_bounds__(256) PngAdlerKernel(PngGeom g, GfxBatch b) {
    __shared__ unsigned long long s_a, s_b;
    const int f = blockIdx.x;
    if (threadIdx.x == 0) s_a = s_b = 0;
    __syncthreads();
    const unsigned long long *slots = b.sums + (size_t)f * b.sum_slots * 2;
    unsigned long long a = 0, bs = 0;
    for (uint32_t i = threadIdx.x; i < b.sum_slots; i += 256u) {
        a += slots[2 * i];
        bs += slots[2 * i + 1];
    }
    a  = WaveSumU64(a);
    bs = WaveSumU64(bs);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_a, a);
        atomicAdd(&s_b, bs);
    }
    __syncthreads();
    if (threadIdx.x == 0) PutBE32(b.png + (size_t)f * b.png_stride + PngAdlerOffset(g), AdlerFromSums(g, s_a, s_b));
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
    // sixteen bytes per load, the next load in flight while these are folded in: the lanes of a wave read
    // 512 bytes apart, so every load touches 64 cache lines -- as dword loads (128 per chunk) the pass spent
    // its time on L2 round trips
    auto step4 = [&](uint32_t w) {
        crc ^= w;
        crc = tab[3][crc & 255u] ^ tab[2][(crc >> 8) & 255u] ^ tab[1][(crc >> 16) & 255u] ^ tab[0][crc >> 24];
    };
    if (n >= 16u) {
        uint32_t q[4], nx[4];
        __builtin_memcpy(q, p, 16);
        for (i = 16u; i + 16u <= n; i += 16u) {
            __builtin_memcpy(nx, p + i, 16);
            step4(q[0]), step4(q[1]), step4(q[2]), step4(q[3]);
            q[0] = nx[0], q[1] = nx[1], q[2] = nx[2], q[3] = nx[3];
        }
        step4(q[0]), step4(q[1]), step4(q[2]), step4(q[3]);
    }
    for (; i + 4 <= n; i += 4) step4(GfxLoadU32(p + i));
    for (; i < n; ++i) crc = (crc >> 8) ^ tab[0][(crc ^ p[i]) & 255u];
    b.chunk_crc[(size_t)f * g.n_chunks + c] = ~crc;
}

// pass 4a: a workgroup combines up to 2^10 chunk CRCs as a binary tree (levels 0..9 of the frame's tree,
// PngTreeParent in gfx_layout.h), a lane per parent.  (First version: a lane walked 64 chunks, then ONE lane
// per frame walked the segments -- 107 dependent GF(2) multiplications of 32 steps each, 240 us per batch.)
__global__ void __launch_bounds__(256) PngCrcTreeKernel(PngGeom g, GfxBatch b) {
    __shared__ uint32_t nodes[kCrcSegment];
    const int f         = blockIdx.y;
    const uint32_t seg  = blockIdx.x, tid = threadIdx.x;
    uint32_t base       = seg * kCrcSegment;
    uint32_t count      = g.n_chunks - base < kCrcSegment ? g.n_chunks - base : kCrcSegment;
    const uint32_t *src = b.chunk_crc + (size_t)f * g.n_chunks + base;
    for (uint32_t i = tid; i < count; i += 256u) nodes[i] = src[i];
    __syncthreads();
    for (uint32_t k = 0; k < kCrcSegmentLog; ++k) {
        const uint32_t pbase = base >> 1, parents = ((base + count + 1u) >> 1) - pbase;  // (<= 512: two per lane)
        uint32_t v[2] = {0u, 0u};
#pragma unroll
        for (uint32_t q = 0; q < 2u; ++q)
            if (tid + 256u * q < parents) v[q] = PngTreeParent(nodes, base, g, k, pbase + tid + 256u * q);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 2u; ++q)
            if (tid + 256u * q < parents) nodes[tid + 256u * q] = v[q];
        __syncthreads();
        base  = pbase;
        count = parents;
    }
    if (tid == 0) b.segment_crc[(size_t)f * g.n_segments + seg] = nodes[0];
}

// pass 4b: a workgroup per frame takes the tree from the segment nodes (level 10) to its root = IDAT's CRC;
// IEND behind it
constexpr uint32_t kMaxSegments = 1024;
__global__ void __launch_bounds__(256) PngTailKernel(PngGeom g, GfxBatch b) {
    __shared__ uint32_t nodes[kMaxSegments];
    const int f        = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    uint32_t count     = g.n_segments;
    for (uint32_t i = tid; i < count; i += 256u) nodes[i] = b.segment_crc[(size_t)f * g.n_segments + i];
    __syncthreads();
    for (uint32_t k = kCrcSegmentLog; count > 1u && k < kCrcLevels; ++k) {
        const uint32_t parents = (count + 1u) >> 1;  // (<= 512: two per lane)
        uint32_t v[2] = {0u, 0u};
#pragma unroll
        for (uint32_t q = 0; q < 2u; ++q)
            if (tid + 256u * q < parents) v[q] = PngTreeParent(nodes, 0u, g, k, tid + 256u * q);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 2u; ++q)
            if (tid + 256u * q < parents) nodes[tid + 256u * q] = v[q];
        __syncthreads();
        count = parents;
    }
    if (tid == 0) PngTail(b.png + (size_t)f * b.png_stride, g, nodes[0]);
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
    if (g.n_segments > kMaxSegments) return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "frame of %d x %d pixels is too large", w, h);
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
    // (lanes of the body pass: a group of four pixels each, and at least one per byte of the fixed head / per block header)
    const uint32_t body_lanes = std::max(std::max(PngBodyGroups(g), kPngIdatData + 2), g.n_blocks);
    const uint32_t body_wgs   = (body_lanes + 255) / 256;
    const size_t o_sums = carve(nf * body_wgs * 2 * sizeof(unsigned long long));
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
    b.sum_slots    = body_wgs;
    b.chunk_crc    = (uint32_t *)(base + o_ccrc);
    b.segment_crc  = (uint32_t *)(base + o_scrc);
    b.out          = (uint8_t *)dout;
    b.out_cap      = out_cap;
    b.headers      = (const uint8_t *)(base + o_head);
    if (kind != kGfxPng) {
        // (`headers` outlives the copy: the stream is synchronised before this function returns)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(base + o_head, headers.data(), headers.size(), hipMemcpyHostToDevice, st));
    }
    const unsigned frames = (unsigned)n_frames;
    hipLaunchKernelGGL(PngBodyKernel, dim3(body_wgs, frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngAdlerKernel, dim3(frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngChunkCrcKernel, dim3((g.n_chunks + 255) / 256, frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngCrcTreeKernel, dim3(g.n_segments, frames), dim3(256), 0, st, g, b);
    hipLaunchKernelGGL(PngTailKernel, dim3(frames), dim3(256), 0, st, g, b);
    if (kind != kGfxPng) {
        // (lanes: four base64 groups each, and at least one per header byte / per kitty separator)
        const uint32_t n_quads = ((g.png_n + 2) / 3 + 3) / 4;
        const uint32_t lanes   = std::max(std::max(n_quads, kGfxHeaderCap), (g.png_n + kKittyChunk - 1) / kKittyChunk);
        hipLaunchKernelGGL(GfxFrameKernel, dim3((lanes + 255) / 256, frames), dim3(256), 0, st, g, b, kind);
    }
    TIMG_HIP_TRY(ctx, hipGetLastError());
    if (!out_on_device) return CopyFramesToHost(ctx, out, out_cap, dout, out_len, n_frames, st);
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
