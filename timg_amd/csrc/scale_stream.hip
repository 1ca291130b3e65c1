// timg_amd/csrc/scale_stream.hip -- streaming scale(+blend) kernel for
// vertical-first shrink plans (what stb picks for 4K -> 800x450 and for
// 4K -> 200x56, see resample_plan.cc VerticalFirst).
//
// Data flow per workgroup = one (column strip, output-row band, frame):
//   * every lane owns kPix = 4 adjacent source columns (one 16-byte load per
//     source row, kPrefetch rows in flight) and walks the band's source rows
//     top to bottom ONCE;
//   * a source row feeds the <= kSlots output rows whose vertical filter
//     covers it; their running sums live in registers and are updated in
//     source-row order -- exactly stb's vertical chain;
//   * when an output row has seen its last source row, its column sums go to
//     one of two LDS staging rows (double buffered: one barrier per completed
//     row) and the strip's output pixels are produced by the horizontal
//     even/odd-chain gather, un-weighted, blended and stored as RGBA8.
// Which slot an output row uses, and the per-row weights, come from a small
// host-built schedule that arrives through the scalar cache, so the kernel has
// no data-dependent control flow beyond wave-uniform branches.
//
// HBM traffic: each source byte is read once per band that needs it (band
// height trades halo re-reads against grid size).  The kernel is VALU-bound,
// not HBM-bound: separately rounded multiplies and adds (no FMA: the
// reference's arithmetic is unfused) cost ~31 lane-ops per source pixel in the
// vertical chain alone -- see DESIGN.md "scale kernel roofline".
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "context.h"
#include "h2_strips.h"
#include "pixel_math.h"

namespace timg_amd {

namespace {

constexpr int kPix       = 4;    // source columns per lane (16-byte loads)
constexpr int kSlots     = 5;    // output rows in flight per lane
#ifndef TIMG_STREAM_THREADS
#define TIMG_STREAM_THREADS 256
#endif
constexpr int kThreads   = TIMG_STREAM_THREADS;
// (a workgroup of ONE wave needs no barriers: a wave's LDS operations execute in order)
__device__ __forceinline__ void BlockSync() {
    if (kThreads > 64) __syncthreads();
}
constexpr int kStripCols = kPix * kThreads;
// staging rows in LDS: two for the 3/4-channel sets (one barrier per completed row), one
// for the 7-channel set (a second barrier per completed row, but half the LDS: two
// workgroups per CU instead of one)
constexpr int kPrefetch  = 4;    // source rows in flight per lane
constexpr int kLdsCoeffFloats = 12 * 1024;  // horizontal weights kept in LDS up to this many

__device__ __forceinline__ uint32_t MinU32(uint32_t a, uint32_t b) { return a < b ? a : b; }

struct StripInfo {
    int ox0, ox1;  // output columns [ox0, ox1)
    int cx0;       // first source column held in LDS (multiple of kPix)
    int pad;
};

struct BandInfo {
    int oy0, oy1;  // output rows [oy0, oy1)
    int r0, r1;    // source rows [r0, r1]
    int sched;     // index of the band's first RowSched
    int pad[3];
};

// Everything one source row of a band does, 64 bytes so that it arrives with
// one scalar load: per slot the vertical weight and what happens to the slot
// (bit0 active, bit1 first contribution, bit2 last -> emit output row flags>>8),
// plus the vertical sum of an all-opaque alpha column for a completing row.
struct RowSched {
    float weight[kSlots];
    int flags[kSlots];
    float alpha_sum[kSlots];
    int any_last;  // some slot completes on this row
};
static_assert(sizeof(RowSched) == 64, "RowSched must stay one cache line");

struct StreamTables {
    const StripInfo *strips;
    const BandInfo *bands;
    const RowSched *sched;
    int n_strips, n_bands;
};

// The schedule tables are written by the host before the launch and never by a
// kernel: reading them through the constant address space lets the compiler use
// scalar loads (s_load_dwordx16 for a RowSched), which neither occupy the vector
// memory queue nor force the pixel prefetches to drain (vmcnt is in-order).
#define TIMG_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ T LoadConstant(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized tables only");
    const TIMG_CONST_AS uint32_t *w = (const TIMG_CONST_AS uint32_t *)(uintptr_t)p;
    uint32_t tmp[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) tmp[i] = w[i];
    T out;
    __builtin_memcpy(&out, tmp, sizeof(T));
    return out;
}

// Channel sets.  stb filters 7 floats per pixel (R G B A RA GA BA); which of
// them a tile really needs depends on its data:
//   kOpaque  every source alpha is 255: A == 1.0f exactly, so RA == R bit for
//            bit, the alpha column sum is the same for every column (host
//            table) and only three chains remain;
//   kPremult A RA GA BA: enough unless some filtered alpha is < 2^-120, in
//            which case stb outputs the straight-filtered RGB instead;
//   kFull    all seven.
// A tile is first tried with the cheapest set; a failed assumption re-runs it
// with the next one, so the result never depends on the shortcut.
enum Mode { kOpaque = 0, kPremult = 1, kFull = 2 };
template <int M> struct ModeTraits;
#ifndef TIMG_STAGE_ROWS
#define TIMG_STAGE_ROWS 2
#endif
#ifndef TIMG_OPAQUE_WAVES
#define TIMG_OPAQUE_WAVES 3
#endif
template <> struct ModeTraits<kOpaque>  { static constexpr int kCh = 3, kStride = 4, kStage = TIMG_STAGE_ROWS; };
template <> struct ModeTraits<kPremult> { static constexpr int kCh = 4, kStride = 4, kStage = TIMG_STAGE_ROWS; };
template <> struct ModeTraits<kFull>    { static constexpr int kCh = 7, kStride = 8, kStage = 1; };

template <int M>
__device__ __forceinline__ void DecodeMode(uint32_t px, float out[ModeTraits<M>::kCh]) {
    // (a b,g,r,a source is filtered as if it were r,g,b,a -- the three colour
    // channels go through identical arithmetic -- and swapped back on output)
    const float k = 1.0f / 255.0f;
    const float r = (float)(px & 0xffu) * k;
    const float g = (float)((px >> 8) & 0xffu) * k;
    const float b = (float)((px >> 16) & 0xffu) * k;
    if (M == kOpaque) {
        out[0] = r;
        out[1] = g;
        out[2] = b;
    } else {
        const float a = (float)(px >> 24) * k;
        if (M == kPremult) {
            out[0] = a;
            out[1] = r * a;
            out[2] = g * a;
            out[3] = b * a;
        } else {
            out[0] = r;
            out[1] = g;
            out[2] = b;
            out[3] = a;
            out[4] = r * a;
            out[5] = g * a;
            out[6] = b * a;
        }
    }
}

__device__ __forceinline__ uint32_t FinishStreamPixel(const Px7 &acc, int x, int y, int swap_rb,
                                                      const DevBlend &blend, int *flag) {
    uint32_t out = EncodePx(acc);
    if (swap_rb) out = (out & 0xff00ff00u) | ((out & 0xffu) << 16) | ((out >> 16) & 0xffu);
    if ((out >> 24) != 0xffu && y >= blend.start_row) {
        if (flag) *flag = 1;
        if (blend.enabled) {
            const bool alt = CheckerAlt(blend, x, y);
            const float bg[3] = {alt ? blend.pat[0] : blend.bg[0], alt ? blend.pat[1] : blend.bg[1],
                                 alt ? blend.pat[2] : blend.bg[2]};
            out = BlendOver(out, bg);
        }
    }
    return out;
}

struct TileCtx {
    const DevPlan *plan;
    const DevBlend *blend;
    const FrameBatch *batch;
    StripInfo si;
    BandInfo bi;
    const RowSched *sched;
    int f;
    float *stage;    // ModeTraits::kStage * kStripCols * kStride floats
    float *hcoef;    // h_width * hrow floats (k-major)
    int2 *htaps;     // hrow entries {first tap - cx0, tap count}
    int hrow;        // outputs per strip rounded up (row length of hcoef)
    int *fail;       // LDS flag
};

// Straight RGB of pixels with (nearly) no alpha only shows when they are not composed: a composed
// pixel whose filtered alpha is below 2^-120 has alpha byte 0 and becomes the background alone, so
// the premultiplied channel set may keep such tiles when every row of the band is composed.
__device__ __forceinline__ bool NeedStraight(const TileCtx &c) {
    return !(c.blend->enabled && c.blend->start_row <= c.bi.oy0);
}

// Horizontal pass of one completed output row y over the strip's outputs:
// stb's gather with an even and an odd accumulation chain (or one chain for
// <= 3 taps), stb_image_resize2.h:5801-6009.  The staged row holds kHc floats
// per source column: kOpaque R G B + the (column-independent) vertical alpha
// sum, kPremult A RA GA BA, kFull all seven.
template <int M>
__device__ __forceinline__ void HorizontalRow(const TileCtx &c, const float *stage_row, int y,
                                              int *flag, bool *ok) {
    constexpr int kStride = ModeTraits<M>::kStride;
    constexpr int kHc     = M == kFull ? 7 : 4;
    const DevPlan &plan     = *c.plan;
    const FrameBatch &batch = *c.batch;
    const int n_out         = c.si.ox1 - c.si.ox0;
    uint8_t *dst_row = batch.dst + (size_t)c.f * batch.dst_frame_stride + (size_t)y * batch.dst_stride;
    for (int o = threadIdx.x; o < n_out; o += kThreads) {
        const int ox      = c.si.ox0 + o;
        const int2 ht     = c.htaps[o];
        const float *base = stage_row + (size_t)ht.x * kStride;
        const float *hw   = c.hcoef + o;  // weights of this column, hrow floats apart
        float even[kHc], odd[kHc];
#pragma unroll
        for (int ch = 0; ch < kHc; ++ch) even[ch] = odd[ch] = 0.0f;
        auto tap = [&](int k, float v[kHc]) {
            const float4 t0 = *reinterpret_cast<const float4 *>(base + k * kStride);
            v[0] = t0.x;
            v[1] = t0.y;
            v[2] = t0.z;
            v[3] = t0.w;
            if (kHc > 4) {
                const float4 t1 = *reinterpret_cast<const float4 *>(base + k * kStride + 4);
                v[kHc > 4 ? 4 : 0] = t1.x;
                v[kHc > 5 ? 5 : 0] = t1.y;
                v[kHc > 6 ? 6 : 0] = t1.z;
            }
        };
        float v[kHc], v1[kHc];
        if (plan.h_sequential) {
            for (int k = 0; k < ht.y; ++k) {
                const float w = hw[k * c.hrow];
                tap(k, v);
#pragma unroll
                for (int ch = 0; ch < kHc; ++ch) even[ch] = even[ch] + v[ch] * w;  // 0 + x == x
            }
        } else {
            int k = 0;
#pragma unroll 2
            for (; k + 1 < ht.y; k += 2) {
                const float w0 = hw[k * c.hrow], w1 = hw[(k + 1) * c.hrow];
                tap(k, v);
                tap(k + 1, v1);
#pragma unroll
                for (int ch = 0; ch < kHc; ++ch) {
                    even[ch] = even[ch] + v[ch] * w0;
                    odd[ch]  = odd[ch] + v1[ch] * w1;
                }
            }
            if (k < ht.y) {
                const float w = hw[k * c.hrow];
                tap(k, v);
#pragma unroll
                for (int ch = 0; ch < kHc; ++ch) even[ch] = even[ch] + v[ch] * w;
            }
#pragma unroll
            for (int ch = 0; ch < kHc; ++ch) even[ch] = even[ch] + odd[ch];
        }
        Px7 px;
        if (M == kOpaque) {
            // with alpha == 1 the straight and the weighted sums coincide
            px.c[0] = even[0];
            px.c[1] = even[1];
            px.c[2] = even[2];
            px.c[3] = even[3];
            px.c[4] = even[0];
            px.c[5] = even[1];
            px.c[6] = even[2];
        } else if (M == kPremult) {
            px.c[0] = px.c[1] = px.c[2] = 0.0f;
            px.c[3] = even[0];
            px.c[4] = even[1];
            px.c[5] = even[2];
            px.c[6] = even[3];
            if (NeedStraight(c) && px.c[3] < TIMG_TINY_F32) *ok = false;  // needs the straight RGB sums
        } else {
#pragma unroll
            for (int ch = 0; ch < 7; ++ch) px.c[ch] = even[ch < kHc ? ch : 0];
        }
        const uint32_t out = FinishStreamPixel(px, ox, y, plan.swap_rb, *c.blend, flag);
        *reinterpret_cast<uint32_t *>(dst_row + (size_t)ox * 4) = out;
    }
}

// Runs one tile with channel set M.  Returns false (uniformly) when M's
// assumption did not hold; the caller then retries with the next set.
template <int M>
__device__ bool RunTile(const TileCtx &c) {
    constexpr int kCh = ModeTraits<M>::kCh, kStride = ModeTraits<M>::kStride;
    const DevPlan &plan     = *c.plan;
    const FrameBatch &batch = *c.batch;
    const int tid           = threadIdx.x;
    const int col0          = c.si.cx0 + tid * kPix;
    // A lane whose four columns straddle the end of the row (in_w not a multiple of 4) loads
    // the LAST four pixels of the row instead and shifts them into place when they are used;
    // lanes entirely past the end re-read those pixels too (never consumed: no tap reaches
    // past in_w).  So loads need no predicate (16-byte loads need only 4-byte alignment).
    const int over  = col0 + kPix - plan.in_w;
    const int shift = (col0 < plan.in_w && over > 0) ? over : 0;
    const uint32_t lane_off = (uint32_t)min(col0, plan.in_w - kPix) * 4u;
    const uint8_t *frame    = batch.src + (size_t)c.f * batch.src_frame_stride;
    int *flag               = batch.transparent_flags ? batch.transparent_flags + c.f : nullptr;
    const int r1            = c.bi.r1;                   // may lie past the image: see BuildVariant
    const int r_last        = min(r1, plan.in_h - 1);    // last row that exists

    auto load_row = [&](int r) -> uint4 {
        const uint8_t *row = frame + (size_t)min(r, r_last) * batch.src_stride;  // uniform
        return *reinterpret_cast<const uint4 *>(row + lane_off);
    };

    float acc[kSlots][kPix][kCh];
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int p = 0; p < kPix; ++p)
#pragma unroll
            for (int ch = 0; ch < kCh; ++ch) acc[s][p][ch] = 0.0f;

    bool ok = true;  // this lane has seen nothing that breaks M's assumption
    int ev  = 0;     // completed rows so far: picks the staging row

    // One source row: decode, feed the active slots, and -- when an output row
    // completes -- stage it and run the horizontal pass.  Returns false when the
    // tile has to be redone with a richer channel set (block-uniform).
    // The schedule entry of row r+1 is requested while row r is processed (the
    // table carries one blank entry past the last band): a scalar load issued and
    // consumed in the same step would expose the scalar-cache latency every row.
    RowSched rs_next = LoadConstant(c.sched);
    auto row_step = [&](const uint4 &q_in, int r) __attribute__((always_inline)) -> bool {
        uint4 q = q_in;
        if (shift) {  // (one lane of the last strip at most)
            if (shift == 1) q = make_uint4(q.y, q.z, q.w, q.w);
            else if (shift == 2) q = make_uint4(q.z, q.w, q.w, q.w);
            else q = make_uint4(q.w, q.w, q.w, q.w);
        }
        const RowSched rs = rs_next;
        asm volatile("" ::"s"(rs.flags[0]), "s"(rs.weight[0]));  // rs is complete here ...
        __builtin_amdgcn_sched_barrier(0);
        rs_next = LoadConstant(c.sched + (r + 1 - c.bi.r0));  // ... before the next request goes out
        if (M == kOpaque) ok = ok && ((q.x & q.y & q.z & q.w) >> 24) == 0xffu;
        // Fully transparent pixels announce filtered alphas of (or below) zero, which
        // need the straight RGB sums: give the tile to the full channel set right away
        // instead of discovering it output pixel by output pixel.
        if (M == kPremult && NeedStraight(c)) ok = ok && MinU32(MinU32(q.x, q.y), MinU32(q.z, q.w)) >= 0x01000000u;  // (alpha is the top byte: every alpha != 0)
        float d[kPix][kCh];
        DecodeMode<M>(q.x, d[0]);
        DecodeMode<M>(q.y, d[1]);
        DecodeMode<M>(q.z, d[2]);
        DecodeMode<M>(q.w, d[3]);

        // (the schedule never lets two output rows complete on the same source row)
        bool done = false;
        int done_y = 0;
        constexpr int kStage = ModeTraits<M>::kStage;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int fl = rs.flags[s];
            if (!(fl & 1)) continue;  // wave-uniform
            const float w = rs.weight[s];
            // (a slot is zero when its row starts: 0 + x == x, so the first
            // contribution needs no special case)
#pragma unroll
            for (int p = 0; p < kPix; ++p)
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch) acc[s][p][ch] = acc[s][p][ch] + d[p][ch] * w;
            if (fl & 4) {
                float *row = c.stage + (size_t)(ev & (kStage - 1)) * kStripCols * kStride;
#pragma unroll
                for (int p = 0; p < kPix; ++p) {
                    float *dst = row + (size_t)(tid * kPix + p) * kStride;
                    if (kStride == 4) {
                        float4 v;
                        v.x = acc[s][p][0];
                        v.y = acc[s][p][1];
                        v.z = acc[s][p][2];
                        v.w = kCh > 3 ? acc[s][p][kCh > 3 ? 3 : 0] : rs.alpha_sum[s];
                        *reinterpret_cast<float4 *>(dst) = v;
                    } else {
                        float4 v0, v1;
                        v0.x = acc[s][p][0];
                        v0.y = acc[s][p][1];
                        v0.z = acc[s][p][2];
                        v0.w = acc[s][p][kCh > 3 ? 3 : 0];
                        v1.x = acc[s][p][kCh > 4 ? 4 : 0];
                        v1.y = acc[s][p][kCh > 5 ? 5 : 0];
                        v1.z = acc[s][p][kCh > 6 ? 6 : 0];
                        v1.w = 0.0f;
                        *reinterpret_cast<float4 *>(dst)     = v0;
                        *reinterpret_cast<float4 *>(dst + 4) = v1;
                    }
                }
#pragma unroll
                for (int p = 0; p < kPix; ++p)
#pragma unroll
                    for (int ch = 0; ch < kCh; ++ch) acc[s][p][ch] = 0.0f;
                done   = true;
                done_y = fl >> 8;
            }
        }
        if (done) {  // wave- and block-uniform
            if (M != kFull && __any(!ok) && (tid & 63) == 0) *c.fail = 1;
#ifndef TIMG_ABL_NOBARRIER
            BlockSync();
#endif
            if (M != kFull && *c.fail) return false;
#ifndef TIMG_ABL_NOHORIZ
            HorizontalRow<M>(c, c.stage + (size_t)(ev & (kStage - 1)) * kStripCols * kStride, done_y, flag,
                             &ok);
#endif
            if (kStage == 1) BlockSync();  // the single staging row is free again
            ++ev;
        }
        return true;
    };

    // kPrefetch = 4 source rows in flight per lane; the ring is unrolled so that
    // "rotating" it is register naming, not moves (a move would have to wait for
    // its load and collapse the prefetch distance to one row)
    static_assert(kPrefetch == 4, "the register ring below is written out for 4 rows");
    uint4 q0 = load_row(c.bi.r0), q1 = load_row(c.bi.r0 + 1), q2 = load_row(c.bi.r0 + 2),
          q3 = load_row(c.bi.r0 + 3);
    for (int r = c.bi.r0; r <= r1; r += kPrefetch) {
        if (!row_step(q0, r)) return false;
        q0 = load_row(r + 4);
        if (r + 1 > r1) break;
        if (!row_step(q1, r + 1)) return false;
        q1 = load_row(r + 5);
        if (r + 2 > r1) break;
        if (!row_step(q2, r + 2)) return false;
        q2 = load_row(r + 6);
        if (r + 3 > r1) break;
        if (!row_step(q3, r + 3)) return false;
        q3 = load_row(r + 7);
    }
    if (M == kFull) return true;
    if (__any(!ok) && (tid & 63) == 0) *c.fail = 1;
    BlockSync();
    return *c.fail == 0;
}

// One kernel per channel set (each gets its own register budget).  They run
// back to back on the stream; tile_state[tile] says which tiles are still open (it holds the
// generation number of the scale call that completed the tile: no memset per call):
// 0 = not produced yet, 1 = done.  A kernel skips tiles that are done and marks
// the ones it completes.
template <int M>
__global__ void __launch_bounds__(kThreads, M == kFull ? 2 : (M == kOpaque ? TIMG_OPAQUE_WAVES : 3))
ScaleStreamKernel(DevPlan plan, StreamTables tab, DevBlend blend, FrameBatch batch,
                  int *tile_state, int gen, int hrow) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int fail;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tile_state[tile] == gen) return;  // done by an earlier kernel of this call (uniform: whole workgroup leaves)
    TileCtx c;
    c.plan       = &plan;
    c.blend      = &blend;
    c.batch      = &batch;
    c.si         = LoadConstant(tab.strips + blockIdx.x);
    c.bi         = LoadConstant(tab.bands + blockIdx.y);
    c.sched      = tab.sched + c.bi.sched;
    c.f          = blockIdx.z;
    c.stage      = lds;
    c.htaps      = reinterpret_cast<int2 *>(lds + ModeTraits<M>::kStage * kStripCols * ModeTraits<M>::kStride);
    c.hcoef      = reinterpret_cast<float *>(c.htaps + hrow);
    c.hrow       = hrow;
    c.fail       = &fail;
    for (int o = threadIdx.x; o < hrow; o += kThreads) {
        const int ox  = c.si.ox0 + o;
        const bool in = ox < c.si.ox1;
        int2 ht       = make_int2(0, 1);
        if (in) {
            ht = plan.h_taps[ox];
            ht.x -= c.si.cx0;
        }
        c.htaps[o]      = ht;
        const float *hc = plan.h_coeff + (size_t)ox * plan.h_width;
        for (int k = 0; k < plan.h_width; ++k) c.hcoef[k * hrow + o] = in ? hc[k] : 0.0f;
    }
    if (threadIdx.x == 0) fail = 0;
    BlockSync();
    if (RunTile<M>(c) && threadIdx.x == 0) tile_state[tile] = gen;
}


// ===================================================================================
// Vertical-first plans, the vertical products on the matrix cores (kOpaque / kPremult).
//
// The vertical chain of stb is  acc = acc + d * w  with the product and the sum rounded
// separately.  On the VALU that is a multiply and an add per (pixel, channel, output row);
// measured (profiles/r2): the kernel above is bound by the instructions it issues, not by
// HBM.  v_mfma_f32_4x4x1_16b_f32 with a zero accumulator delivers correctly rounded fp32
// PRODUCTS (D = A*B + 0; checked against v_mul_f32 on 10^6 operand pairs including
// denormals, scratch/ubench/mfma_probe.hip) as 4x4 outer products: with A = the weights of
// four output rows (lanes 0..3, broadcast to all blocks: cbsz 4) and B = a lane's own sample,
// every lane receives its sample times the four weights in four registers.  So the
// multiplications of four output rows leave the VALU (the matrix pipe runs beside it), the
// VALU keeps the additions -- as packed adds on (row 0,row 1) (row 2,row 3) accumulator
// pairs -- and the per-row "is this slot active" branches disappear: an idle slot has
// weight 0 and adds +0.  At most five output rows are live on a source row; the fifth
// ("overflow") row is carried on the VALU and moves into a matrix slot as soon as one
// completes (host-built schedule: RowCtl).
//
// Everything else -- strips, bands, staging rows, the horizontal even/odd gather, the
// channel-set fallback chain -- is as in the kernel above.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int kMSlots = 4;
#ifdef TIMG_M_TRACE
// (experiment build only) where a wave's time goes, s_memtime ticks summed over all waves:
// [0] waiting for its source row, [1] decode + vertical products and sums, [2] staging a completed row + barrier,
// [3] horizontal pass, [4] the barrier that frees the staging row, [5] waves, [6] prologue, [7] whole kernel
__device__ unsigned long long g_mtrace[8];
#define TIMG_M_MARK(ACC)                                        \
    {                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        ACC += now_ - t_last;                                   \
        t_last = now_;                                          \
    }
#else
#define TIMG_M_MARK(ACC)
#endif

struct RowW {
    float w[kMSlots];  // weights of the matrix slots on this source row (0: idle)
};
// flags: bit 0 overflow row active; bits 4..6: completing slot + 1 (1..4 matrix slot,
// 5 overflow row), 0 none; bit 7: the overflow row moves into the slot that completed
struct RowCtl {
    float ovf_w;
    int flags;
    int done_y;
    float alpha_sum;  // vertical sum of an all-opaque alpha column of the completing row
};
struct RowRec {  // 48 bytes through the scalar cache per source row, from a pointer that just advances
    RowW w;
    float keep[kMSlots];  // 0 on the source row that STARTS an output row in the slot, else 1 (see the sums)
    RowCtl ctl;
};
// kOpaque: what the horizontal pass would compute in its alpha channel.  Every staged column of an all-opaque
// tile carries the SAME alpha (the vertical weight sum of the row, a host constant), so the filtered alpha of
// output column x is a function of (x, that constant) alone -- the host evaluates stb's even/odd chain for it
// once per distinct constant and stores what the pixel needs of it: {1.0f / alpha, ToByte(alpha) << 24}.  The
// kernel's tap loop then carries three channels, and the un-weighting has no divide.  RowCtl::flags bits 8..23
// say which table row the completing output row uses.
struct AlphaCell {
    float inv;          // 1.0f / alpha  (IEEE divide, as EncodePx computes it)
    uint32_t byte_hi;   // ToByte(alpha) << 24
};
struct MTables {
    const RowRec *rec;  // indexed like StreamTables::sched (BandInfo::sched + row - r0)
    const AlphaCell *alpha_tab;  // [distinct vertical alpha sums][out_w]
};

// one correctly rounded fp32 multiplication / addition as ONE instruction (never contracted, never regrouped)
__device__ __forceinline__ float MulRn(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a float4 slot of LDS as ONE 16-byte read: the value passes through an (empty) asm as a whole register tuple, so
// that the compiler cannot split the read by its uses (a ds_read_b96, or two 8-byte halves, cost the LDS twice
// the cycles of a ds_read_b128)
__device__ __forceinline__ f4v LdsSlot(const void *p) {
    f4v v = *reinterpret_cast<const f4v *>(p);
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float AddRn(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Horizontal pass of a completed row, every lane its own output column(s): weights as
// float4 groups in LDS (taps beyond a column's count have weight 0 and read staged or
// zeroed data: acc + v * 0 leaves the sum as it is), tap k of the pair (k, k+1) feeds the
// even / odd chain -- stb_image_resize2.h:5801-6009 as in HorizontalRow above.
template <int M>
__device__ __forceinline__ void HorizontalRowM(const DevPlan &plan, const DevBlend &blend, const FrameBatch &batch,
                                               const StripInfo &si, int f, const float *stage_row, const float *hw,
                                               const int *hbase, int hrow, int hgroups, int y, int *flag, bool need_straight,
                                               bool *ok) {
    const int n_out  = si.ox1 - si.ox0;
    uint8_t *dst_row = batch.dst + (size_t)f * batch.dst_frame_stride + (size_t)y * batch.dst_stride;
    for (int o = threadIdx.x; o < n_out; o += kThreads) {
        const int ox          = si.ox0 + o;
        const char *base      = reinterpret_cast<const char *>(stage_row) + hbase[o];
        const float *wbase    = hw + (size_t)o * 4;
        float even[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (plan.h_sequential) {
            for (int g = 0; g < hgroups; ++g) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wbase + (size_t)g * hrow * 4);
                const float wk[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = *reinterpret_cast<const float4 *>(base + (g * 4 + j) * 16);
                    even[0] = even[0] + t.x * wk[j];
                    even[1] = even[1] + t.y * wk[j];
                    even[2] = even[2] + t.z * wk[j];
                    even[3] = even[3] + t.w * wk[j];
                }
            }
        } else {
            // Channel PAIRS (x, y) and (z, w) of a staged column as the operands of packed multiplies and adds -- said
            // as two-float vectors: left to pair the scalar form itself, the vectoriser paired channels of different
            // taps and paid ten register moves and four unpacked multiplies per four taps (28 instructions where 16 do;
            // round 5, read off the premultiplied kernel's assembly).  Same operations in the same order per channel.
            f2v e01 = {0.0f, 0.0f}, e23 = {0.0f, 0.0f}, o01 = {0.0f, 0.0f}, o23 = {0.0f, 0.0f};
            for (int g = 0; g < hgroups; ++g) {
                const f4v w4 = LdsSlot(wbase + (size_t)g * hrow * 4);
                const f4v t0 = LdsSlot(base + (g * 4 + 0) * 16);
                const f4v t1 = LdsSlot(base + (g * 4 + 1) * 16);
                const f4v t2 = LdsSlot(base + (g * 4 + 2) * 16);
                const f4v t3 = LdsSlot(base + (g * 4 + 3) * 16);
                e01 = e01 + f2v{t0.x, t0.y} * f2v{w4.x, w4.x};
                e23 = e23 + f2v{t0.z, t0.w} * f2v{w4.x, w4.x};
                o01 = o01 + f2v{t1.x, t1.y} * f2v{w4.y, w4.y};
                o23 = o23 + f2v{t1.z, t1.w} * f2v{w4.y, w4.y};
                e01 = e01 + f2v{t2.x, t2.y} * f2v{w4.z, w4.z};
                e23 = e23 + f2v{t2.z, t2.w} * f2v{w4.z, w4.z};
                o01 = o01 + f2v{t3.x, t3.y} * f2v{w4.w, w4.w};
                o23 = o23 + f2v{t3.z, t3.w} * f2v{w4.w, w4.w};
            }
            e01     = e01 + o01;
            e23     = e23 + o23;
            even[0] = e01.x;
            even[1] = e01.y;
            even[2] = e23.x;
            even[3] = e23.y;
        }
        Px7 px;
        if (M == kOpaque) {  // with alpha == 1 the straight and the weighted sums coincide
            px.c[0] = even[0];
            px.c[1] = even[1];
            px.c[2] = even[2];
            px.c[3] = even[3];
            px.c[4] = even[0];
            px.c[5] = even[1];
            px.c[6] = even[2];
        } else {
            px.c[0] = px.c[1] = px.c[2] = 0.0f;
            px.c[3] = even[0];
            px.c[4] = even[1];
            px.c[5] = even[2];
            px.c[6] = even[3];
            // a filtered alpha below 2^-120 makes stb output the straight-filtered RGB, which
            // this channel set does not carry -- unless the pixel is composed over the
            // background anyway: its alpha byte is 0 and the result is the background alone
            if (need_straight && px.c[3] < TIMG_TINY_F32) *ok = false;
        }
        const uint32_t out = FinishStreamPixel(px, ox, y, plan.swap_rb, blend, flag);
        *reinterpret_cast<uint32_t *>(dst_row + (size_t)ox * 4) = out;
    }
}

// The same pass for all-opaque tiles: three channels in the chains, alpha and its reciprocal from the host's
// table (AlphaCell).  R G B go through exactly the arithmetic of HorizontalRowM -- tap k of the pair (k, k + 1)
// feeds the even / odd chain, padded taps add v * 0 -- and are un-weighted by the multiplication EncodePx does.
__device__ __forceinline__ void HorizontalRowOpaque(const DevPlan &plan, const DevBlend &blend, const FrameBatch &batch,
                                                    const StripInfo &si, int f, const float *stage_row, const float *hw,
                                                    const int *hbase, int hrow, int hgroups, int y,
                                                    const AlphaCell *alpha_row, int *flag) {
    const int n_out  = si.ox1 - si.ox0;
    uint8_t *dst_row = batch.dst + (size_t)f * batch.dst_frame_stride + (size_t)y * batch.dst_stride;
    for (int o = threadIdx.x; o < n_out; o += kThreads) {
        const int ox       = si.ox0 + o;
        const AlphaCell ac = alpha_row[ox];  // (requested first: the tap loop hides it)
        // (staged columns and weight groups are float4 slots: said out loud, or the 16-byte reads are split)
        const char *base   = reinterpret_cast<const char *>(
            __builtin_assume_aligned(reinterpret_cast<const char *>(stage_row) + hbase[o], 16));
        const float *wbase = reinterpret_cast<const float *>(__builtin_assume_aligned(hw + (size_t)o * 4, 16));
        float even[3] = {0.0f, 0.0f, 0.0f}, odd[3] = {0.0f, 0.0f, 0.0f};
        if (plan.h_sequential) {
            for (int g = 0; g < hgroups; ++g) {
                const f4v w4 = LdsSlot(wbase + (size_t)g * hrow * 4);
                const float wk[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4v t = LdsSlot(base + (g * 4 + j) * 16);
                    even[0] = even[0] + t.x * wk[j];
                    even[1] = even[1] + t.y * wk[j];
                    even[2] = AddRn(even[2], MulRn(t.z, wk[j]));
                }
            }
        } else {
            for (int g = 0; g < hgroups; ++g) {
                const f4v w4 = LdsSlot(wbase + (size_t)g * hrow * 4);
                const f4v t0 = LdsSlot(base + (g * 4 + 0) * 16);
                const f4v t1 = LdsSlot(base + (g * 4 + 1) * 16);
                const f4v t2 = LdsSlot(base + (g * 4 + 2) * 16);
                const f4v t3 = LdsSlot(base + (g * 4 + 3) * 16);
                // R and G as the packed pair the compiler forms; B through MulRn / AddRn (single instructions the
                // vectoriser cannot regroup: left alone it pairs B of tap k with B of tap k + 2 and pays seven
                // register moves per four taps for it).
                even[0] = even[0] + t0.x * w4.x;
                even[1] = even[1] + t0.y * w4.x;
                even[2] = AddRn(even[2], MulRn(t0.z, w4.x));
                odd[0]  = odd[0] + t1.x * w4.y;
                odd[1]  = odd[1] + t1.y * w4.y;
                odd[2]  = AddRn(odd[2], MulRn(t1.z, w4.y));
                even[0] = even[0] + t2.x * w4.z;
                even[1] = even[1] + t2.y * w4.z;
                even[2] = AddRn(even[2], MulRn(t2.z, w4.z));
                odd[0]  = odd[0] + t3.x * w4.w;
                odd[1]  = odd[1] + t3.y * w4.w;
                odd[2]  = AddRn(odd[2], MulRn(t3.z, w4.w));
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) even[ch] = even[ch] + odd[ch];
        }
        // EncodePx with alpha >= 2^-120 (the host made sure): c * (1.0f / alpha), quantised; alpha's byte from the table
        uint32_t out = ToByte(even[0] * ac.inv) | (ToByte(even[1] * ac.inv) << 8) | (ToByte(even[2] * ac.inv) << 16) | ac.byte_hi;
        if (plan.swap_rb) out = (out & 0xff00ff00u) | ((out & 0xffu) << 16) | ((out >> 16) & 0xffu);
        if ((out >> 24) != 0xffu && y >= blend.start_row) {  // (as FinishStreamPixel)
            if (flag) *flag = 1;
            if (blend.enabled) {
                const bool alt    = CheckerAlt(blend, ox, y);
                const float bg[3] = {alt ? blend.pat[0] : blend.bg[0], alt ? blend.pat[1] : blend.bg[1],
                                     alt ? blend.pat[2] : blend.bg[2]};
                out = BlendOver(out, bg);
            }
        }
        *reinterpret_cast<uint32_t *>(dst_row + (size_t)ox * 4) = out;
    }
}

// kOvf: the plan can have five live output rows (overflow row compiled in); most shrink ratios
// never have more than four.
// (waves per SIMD pinned to exactly 3 -- LDS allows no more: the compiler then uses the 168 registers
// it may have instead of squeezing into 128 for an occupancy the kernel cannot reach)
// The opaque set without the overflow row needs 115 registers: four workgroups per CU with ONE staging
// row (a second barrier per completed row frees it), 2 % faster than three with two staging rows.  The other
// instantiations spill at 128 registers and stay at three -- except the premultiplied set WITH the
// overflow row (rare: alpha frames at a ratio with five live output rows), which spills even at 168 and
// runs two per CU: a spilled kernel is not just slower here, the compiler spilled ring registers whose
// loads were still in flight (see issue_next_row; check_ring_isa.py caught it).
template <int M, bool kOvf> struct MKernelShape {
    // (per SIMD = workgroups per CU)  Both sets without the overflow row run four: the opaque one needs 113 registers; the
    // premultiplied one fits 125 with THREE source rows in flight instead of four (at four it needs 129 and spilled ring
    // registers behind their loads: check_ring_isa.py refused it) -- 0.893 -> 0.868 ms per 64 S-alpha frames.
    static constexpr int kWaves = !kOvf ? 4 : (M == kPremult) ? 2 : 3;
#ifdef TIMG_M_PREMULT_DEPTH
    static constexpr int kDepth = (M == kPremult && !kOvf) ? TIMG_M_PREMULT_DEPTH : 4;  // source rows in flight per lane
#else
    static constexpr int kDepth = (M == kPremult && !kOvf) ? 3 : 4;  // source rows in flight per lane
#endif
    static constexpr int kStage = kWaves == 4 ? 1 : 2;              // staging rows
};
// wave priorities of the two-column horizontal-first kernel's phases: the tap chain, then the vertical sums and the next row's
// decode (the one-column kernel measured 1.4 % SLOWER with the same pair at 52 taps: it has none)
#ifndef TIMG_H_PRIO_T
#define TIMG_H_PRIO_T 1
#endif
#ifndef TIMG_H_PRIO_V
#define TIMG_H_PRIO_V 0
#endif
// wave priorities of the matrix kernel's two phases (s_setprio; see the horizontal pass's call)
#ifndef TIMG_M_PRIO
#define TIMG_M_PRIO 0
#endif
#ifndef TIMG_M_PRIO_H
#define TIMG_M_PRIO_H 1
#endif
template <int M, bool kOvf>
__device__ __forceinline__ void ScaleStreamMBody(const DevPlan &plan, const StreamTables &tab, const MTables &mt,
                                                 const DevBlend &blend, const FrameBatch &batch,
                                                 int *tile_state, int gen, int hrow, int hgroups) {
    static_assert(M == kOpaque || M == kPremult, "three or four channels");
    constexpr int kCh = M == kOpaque ? 3 : 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int fail;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tile_state[tile] == gen) return;  // done by an earlier kernel of this call (uniform: whole workgroup leaves)
    const StripInfo si = LoadConstant(tab.strips + blockIdx.x);
    const BandInfo bi  = LoadConstant(tab.bands + blockIdx.y);
    const int f        = blockIdx.z;
    const int tid      = threadIdx.x;
    // LDS: two staging rows of (strip + zeroed pad for the padded taps) float4 columns, the
    // horizontal weights as [group][output] float4, the byte offset of every output's first tap
    constexpr int kStageM = MKernelShape<M, kOvf>::kStage;
    const int stage_cols  = kStripCols + 4 * hgroups;
    float *stage          = lds;
    float *hw             = stage + (size_t)kStageM * stage_cols * 4;
    int *hbase           = reinterpret_cast<int *>(hw + (size_t)hgroups * hrow * 4);
    for (int i = tid; i < kStageM * 4 * hgroups; i += kThreads) {
        const int row = i / (4 * hgroups), col = kStripCols + i % (4 * hgroups);
        *reinterpret_cast<float4 *>(stage + ((size_t)row * stage_cols + col) * 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    for (int o = tid; o < hrow; o += kThreads) {
        const int ox  = si.ox0 + o;
        const bool in = ox < si.ox1;
        int2 ht       = make_int2(si.cx0, 0);
        if (in) ht = plan.h_taps[ox];
        hbase[o]        = (ht.x - si.cx0) * 16;
        const float *hc = plan.h_coeff + (size_t)ox * plan.h_width;
        for (int g = 0; g < hgroups; ++g) {
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = g * 4 + j;
                w[j]        = (in && k < ht.y && k < plan.h_width) ? hc[k] : 0.0f;
            }
            *reinterpret_cast<float4 *>(hw + ((size_t)g * hrow + o) * 4) = make_float4(w[0], w[1], w[2], w[3]);
        }
    }
    if (tid == 0) fail = 0;
    BlockSync();

    const int col0 = si.cx0 + tid * kPix;
    // (lanes straddling or past the end of the row: see RunTile)
    const int over          = col0 + kPix - plan.in_w;
    const int shift         = (col0 < plan.in_w && over > 0) ? over : 0;
    // (through readfirstlane: in a scalar register, so the per-row test is a scalar branch)
    const int any_shift_s = __builtin_amdgcn_readfirstlane((plan.in_w & 3) != 0 && si.cx0 + kStripCols > plan.in_w);
    const uint32_t lane_off = (uint32_t)min(col0, plan.in_w - kPix) * 4u;
    const uint8_t *frame    = batch.src + (size_t)f * batch.src_frame_stride;
    int *flag               = batch.transparent_flags ? batch.transparent_flags + f : nullptr;
    const int r1            = bi.r1;
    const int r_last        = min(r1, plan.in_h - 1);
    // straight RGB of pixels with (nearly) no alpha only shows when they are not composed
    const bool need_straight = !(blend.enabled && blend.start_row <= bi.oy0);
    // Source rows are fetched in order through a byte offset that just advances (rows past the
    // image's last one -- virtual rows of the schedule -- re-read the last row): no 64-bit
    // multiply per row.  (Frames are far below 4 GB.)
    const uint8_t *frame_lane = frame + lane_off;
    uint32_t next_off         = (uint32_t)((size_t)min(bi.r0, r_last) * batch.src_stride);
    const uint32_t last_off   = (uint32_t)((size_t)r_last * batch.src_stride);
    const uint32_t row_step_b = (uint32_t)batch.src_stride;
    // The loads are inline assembly with hand-placed waits.  Left to the compiler, the waits in front of
    // a ring slot came out as vmcnt(1) where vmcnt(3) is what the ring allows: its count of what is in
    // flight merges pessimistically where the completion path (which issues stores) rejoins, and the
    // prefetch collapsed to one or two rows -- memory time and compute time simply added up (0.40 ms of
    // loads + 0.32 ms of arithmetic = 0.72 ms).  Memory operations complete in order, so "at most
    // kDepth - 1 operations outstanding" means the oldest load of the ring has arrived (stores issued in
    // between only make the wait more conservative).
    // THE HAZARD of this construction: the compiler believes the load's result is there when the asm
    // statement ends.  Any instruction it places between the load and the ring register's s_waitcnt that
    // touches the register (a copy for a tied operand, a live-range split, a back-edge copy into another
    // register set) reads or clobbers data still in flight.  Nothing in the language forbids that, so
    // the BUILD checks the generated code: check_ring_isa.py walks the kernel's assembly and fails the
    // build if any instruction outside the asm statements names a ring register between its load and
    // its wait (two arrangements tried in round 2 -- requesting the next row before a completed row's
    // horizontal pass, and a fifth register set -- compiled into exactly such copies).
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto issue_next_row = [&](u4v &q) __attribute__((always_inline)) {
        const uint8_t *p = frame_lane + next_off;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
        next_off = min(next_off + row_step_b, last_off);  // uniform
    };

    float acc[kPix][kCh][kMSlots];  // the four matrix slots; updated as pairs (0,1) (2,3) by packed fmas
    float ovf[kPix][kCh];   // the overflow row
#pragma unroll
    for (int p = 0; p < kPix; ++p)
#pragma unroll
        for (int ch = 0; ch < kCh; ++ch) {
#pragma unroll
            for (int k = 0; k < kMSlots; ++k) acc[p][ch][k] = 0.0f;
            ovf[p][ch] = 0.0f;
        }
    uint32_t amin = 0xffffffffu;  // kOpaque: minimum over the pixels seen (alpha is the top byte)
    bool ok       = true;
    int ev        = 0;
    int wa        = 0;  // A operand: lanes 0..3 hold the four slot weights (rewritten every row)
#ifdef TIMG_M_TRACE
    unsigned long long tr_wait = 0, tr_vert = 0, tr_stage = 0, tr_horiz = 0, tr_bar2 = 0, tr_pro = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
    unsigned long long t_last = t_begin;
#endif

    const RowRec *rec_ptr = mt.rec + bi.sched;
    RowRec rec_next       = LoadConstant(rec_ptr);
    auto row_step = [&](const uint4 &q_in, int r) __attribute__((always_inline)) -> bool {
        uint4 q = q_in;
        if (any_shift_s) {  // scalar branch (last strip of an image whose width is not a multiple of 4); selects inside
            const uint32_t a = q.x, b2 = q.y, c = q.z, d2 = q.w;
            q.x = shift == 0 ? a : shift == 1 ? b2 : shift == 2 ? c : d2;
            q.y = shift == 0 ? b2 : shift == 1 ? c : d2;
            q.z = shift == 0 ? c : d2;
        }
        const RowW rw    = rec_next.w;
        const RowCtl ctl = rec_next.ctl;
        // (keep0, keep1) (keep2, keep3) as scalar register pairs
        const f2v keep_lo = {rec_next.keep[0], rec_next.keep[1]}, keep_hi = {rec_next.keep[2], rec_next.keep[3]};
        asm volatile("" ::"s"(ctl.flags), "s"(rw.w[0]));  // the record is complete here ...
        __builtin_amdgcn_sched_barrier(0);
        rec_next = LoadConstant(++rec_ptr);  // ... before the next request goes out
        asm volatile("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\t"
                     "v_writelane_b32 %0, %4, 3\n\ts_nop 3"  // (the compiler does not see this VALU write -> MFMA read hazard)
                     : "+v"(wa)
                     : "s"(__builtin_amdgcn_readfirstlane(__float_as_int(rw.w[0]))),
                       "s"(__builtin_amdgcn_readfirstlane(__float_as_int(rw.w[1]))),
                       "s"(__builtin_amdgcn_readfirstlane(__float_as_int(rw.w[2]))),
                       "s"(__builtin_amdgcn_readfirstlane(__float_as_int(rw.w[3]))));  // (uniform: folds away)
        if (M == kOpaque) amin = min(min(min(min(amin, q.x), q.y), q.z), q.w);  // (a chain: two v_min3_u32)
        // (fully transparent pixels announce filtered alphas of zero: see RunTile)
        if (M == kPremult && need_straight)
            ok = ok && MinU32(MinU32(q.x, q.y), MinU32(q.z, q.w)) >= 0x01000000u;  // (alpha is the top byte: every alpha != 0)
#if defined(TIMG_MABL) && TIMG_MABL == 3
        return true;  // (ablation: the loads and the alpha minimum alone)
#endif
        const f4v zero  = {0.0f, 0.0f, 0.0f, 0.0f};
        const float waf = __int_as_float(wa);
        const uint32_t qs[kPix] = {q.x, q.y, q.z, q.w};
        // kBatch pixels at a time: decode, products, sums (the products' registers are the
        // kernel's peak, so only one batch of them is in flight)
#ifdef TIMG_M_BATCH
        constexpr int kBatch = M == kOpaque ? TIMG_M_BATCH : 1;
#else
        constexpr int kBatch = M == kOpaque ? 2 : 1;
#endif
#pragma unroll
        for (int p0 = 0; p0 < kPix; p0 += kBatch) {
            if (p0) __builtin_amdgcn_sched_barrier(0);
            float d[kBatch][kCh];
            if (M == kOpaque) {
                // u8 * (1/255) for the six colour bytes of two pixels as three packed multiplies
                static_assert(M != kOpaque || kBatch <= 2, "one or two pixels");
                const uint32_t pa = qs[p0], pb = qs[p0 + (kBatch > 1 ? 1 : 0)];
                const f2v k2 = {1.0f / 255.0f, 1.0f / 255.0f};
                const f2v m0 = f2v{(float)(pa & 0xffu), (float)((pa >> 8) & 0xffu)} * k2;
                const f2v m1 = f2v{(float)((pa >> 16) & 0xffu), (float)(pb & 0xffu)} * k2;
                const f2v m2 = f2v{(float)((pb >> 8) & 0xffu), (float)((pb >> 16) & 0xffu)} * k2;
                d[0][0] = m0.x;
                d[0][1] = m0.y;
                d[0][kCh > 2 ? 2 : 0] = m1.x;
                d[kBatch > 1 ? 1 : 0][0] = m1.y;
                d[kBatch > 1 ? 1 : 0][1] = m2.x;
                d[kBatch > 1 ? 1 : 0][kCh > 2 ? 2 : 0] = m2.y;
            } else {
#pragma unroll
                for (int b = 0; b < kBatch; ++b) DecodeMode<M>(qs[p0 + b], d[b]);
            }
#if defined(TIMG_MABL) && TIMG_MABL >= 2
            if (p0 == 0) acc[0][0][0] = acc[0][0][0] + d[0][0] + d[0][1];  // (ablation: keep the decode alive)
            continue;
#endif
            // the batch's MFMAs back to back, then its sums: a sum issued right behind its MFMA
            // would wait for it in s_nop states (the scheduler likes to pair them up)
            f4v prod[kBatch][kCh];
#pragma unroll
            for (int b = 0; b < kBatch; ++b)
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch)  // (w0 d, w1 d, w2 d, w3 d), each the correctly rounded product
                    prod[b][ch] = __builtin_amdgcn_mfma_f32_4x4x1f32(waf, d[b][ch], zero, 4, 0, 0);
#ifndef TIMG_M_NOSPLIT
            __builtin_amdgcn_sched_barrier(0);
#endif
            // acc = acc * keep + prod as ONE packed fma per accumulator pair.
            // keep is 1 (acc * 1 is exact, so this is the separately rounded acc + prod of the chain) or,
            // on the source row that starts a new output row in the slot, 0: the sum restarts as 0 + prod
            // == prod.  So a completed slot is never cleared and nothing on the completion path writes an
            // accumulator -- written as C++ adds plus per-slot clearing, the sums landed in fresh
            // registers and the (hot) path without a completed row copied all 24 accumulator pairs back,
            // 24 v_mov_b64 per source row.
#pragma unroll
            for (int b = 0; b < kBatch; ++b)
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch) {
                    const f2v lo = {prod[b][ch].x, prod[b][ch].y}, hi = {prod[b][ch].z, prod[b][ch].w};
                    float *a      = acc[p0 + b][ch];
                    const f2v n01 = __builtin_elementwise_fma(f2v{a[0], a[1]}, keep_lo, lo);
                    const f2v n23 = __builtin_elementwise_fma(f2v{a[2], a[3]}, keep_hi, hi);
                    a[0] = n01.x;
                    a[1] = n01.y;
                    a[2] = n23.x;
                    a[3] = n23.y;
                }
            if (kOvf && (ctl.flags & 1)) {  // wave-uniform: the overflow row
#pragma unroll
                for (int b = 0; b < kBatch; ++b)
#pragma unroll
                    for (int ch = 0; ch < kCh; ++ch) ovf[p0 + b][ch] = ovf[p0 + b][ch] + d[b][ch] * ctl.ovf_w;
            }
        }
#if defined(TIMG_MABL) && TIMG_MABL == 4
        return true;  // (ablation: loads + decode, no completion handling)
#endif
        TIMG_M_MARK(tr_vert)
        const int code = (ctl.flags >> 4) & 7;
        if (code == 0) return true;  // wave- and block-uniform
#ifdef TIMG_M_PRIO_S  // (experiment: the priority of the staging phase in front of the barrier)
        __builtin_amdgcn_s_setprio(TIMG_M_PRIO_S);
#endif

        // an output row is complete: its column sums go to a staging row
        float *row = stage + (size_t)(ev & (kStageM - 1)) * stage_cols * 4;
        float4 *dst = reinterpret_cast<float4 *>(row + (size_t)tid * kPix * 4);
        const bool move = kOvf && (ctl.flags & 0x80) != 0;
        auto finish_slot = [&](auto comp_tag) {
            constexpr int C = decltype(comp_tag)::value;
#pragma unroll
            for (int p = 0; p < kPix; ++p) {
                float4 v;
                v.x = acc[p][0][C];
                v.y = acc[p][1][C];
                v.z = acc[p][2][C];
                v.w = kCh > 3 ? acc[p][kCh > 3 ? 3 : 0][C] : ctl.alpha_sum;
                dst[p] = v;
            }
            if (move) {  // the overflow row continues in the slot that has just been freed
#pragma unroll
                for (int p = 0; p < kPix; ++p)
#pragma unroll
                    for (int ch = 0; ch < kCh; ++ch) {
                        acc[p][ch][C] = ovf[p][ch];
                        ovf[p][ch]                = 0.0f;
                    }
            } else if (kOvf) {  // (without an overflow row nothing is cleared: the next row in this slot starts with keep == 0)
#pragma unroll
                for (int p = 0; p < kPix; ++p)
#pragma unroll
                    for (int ch = 0; ch < kCh; ++ch) acc[p][ch][C] = 0.0f;
            }
        };
        if (!kOvf) {
            // No overflow row: the completed slot is only READ here (it restarts through keep == 0), and it
            // is read through selects on the uniform slot number -- one code path.  (Four specialised
            // paths that only read get merged by the compiler into ONE path indexing a copy of the
            // accumulators in scratch memory, with a store behind every update in the hot loop.  Round 5 tried
            // TWO paths by slot pair with one select per value, the stores inside the branches: 12 selects
            // instead of 36 -- and 4.55 ms instead of 0.67: the same demotion.  profiles/r5/scale_variants.txt.)
            const bool s0 = code == 1, s1 = code == 2, s2 = code == 3;
            auto pick = [&](const float a[kMSlots]) { return s0 ? a[0] : s1 ? a[1] : s2 ? a[2] : a[3]; };
#pragma unroll
            for (int p = 0; p < kPix; ++p) {
                float4 v;
                v.x = pick(acc[p][0]);
                v.y = pick(acc[p][1]);
                v.z = pick(acc[p][2]);
                v.w = kCh > 3 ? pick(acc[p][kCh > 3 ? 3 : 0]) : ctl.alpha_sum;
                dst[p] = v;
            }
        } else if (code == 1) finish_slot(std::integral_constant<int, 0>());
        else if (code == 2) finish_slot(std::integral_constant<int, 1>());
        else if (code == 3) finish_slot(std::integral_constant<int, 2>());
        else if (code == 4) finish_slot(std::integral_constant<int, 3>());
        else {
#pragma unroll
            for (int p = 0; p < kPix; ++p) {
                float4 v;
                v.x = ovf[p][0];
                v.y = ovf[p][1];
                v.z = ovf[p][2];
                v.w = kCh > 3 ? ovf[p][kCh > 3 ? 3 : 0] : ctl.alpha_sum;
                dst[p] = v;
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch) ovf[p][ch] = 0.0f;
            }
        }
        if (M == kOpaque) ok = (amin >> 24) == 0xffu;
        if (__any(!ok) && (tid & 63) == 0) fail = 1;
        BlockSync();
        TIMG_M_MARK(tr_stage)
        if (fail) return false;
        // Wave priority by phase (profiles/r6/scale_prio.txt): a wave in the horizontal pass goes first at its SIMD's
        // issue port -- its workgroup's other waves are parked at the barrier behind this pass, while a wave in the
        // vertical loop that is held up a little only takes its loaded rows later.  2 % of the launch (0.617 -> 0.603-0.608
        // ms, S-alpha 0.869 -> 0.849-0.852); the other way round (vertical phase high) 1 % slower; priority from the staging
        // stores on: the same as from here.
        __builtin_amdgcn_s_setprio(TIMG_M_PRIO_H);
#if !defined(TIMG_MABL) || TIMG_MABL < 1
        if (M == kOpaque)
            HorizontalRowOpaque(plan, blend, batch, si, f, row, hw, hbase, hrow, hgroups, ctl.done_y,
                                mt.alpha_tab + (size_t)((ctl.flags >> 8) & 0xffff) * plan.out_w, flag);
        else
            HorizontalRowM<M>(plan, blend, batch, si, f, row, hw, hbase, hrow, hgroups, ctl.done_y, flag, need_straight, &ok);
#endif
        TIMG_M_MARK(tr_horiz)
        __builtin_amdgcn_s_setprio(TIMG_M_PRIO);
        if (kStageM == 1) BlockSync();  // the single staging row is free again
        TIMG_M_MARK(tr_bar2)
        ++ev;
        return true;
    };

    static_assert(kPrefetch == 4, "the register ring below is written out for 4 rows");
    // kDepth source rows in flight per lane (the register ring is unrolled: "rotating" it is register
    // naming, not moves).  Steady-state ablations (TIMG_MABL, profiles/r2/ablation_matrix_kernel.txt):
    // the loads alone 0.40 ms (HBM-bound: 6.1 TB/s of traffic), + decode 0.40, + staging and barrier
    // 0.43, + vertical products and sums 0.56, + horizontal pass 0.72.
    // (8 rows in flight measure the same as 4 -- 0.722 vs 0.727 ms -- so the smaller code stays the default)
#ifdef TIMG_M_DEPTH
    constexpr int kDepth = TIMG_M_DEPTH;
#else
    constexpr int kDepth = MKernelShape<M, kOvf>::kDepth;
#endif
    static_assert(kDepth == 3 || kDepth == 4 || kDepth == 8, "ring written out for 3, 4 or 8 rows");
    u4v q0, q1, q2, q3 = {0, 0, 0, 0}, q4 = {0, 0, 0, 0}, q5 = q4, q6 = q4, q7 = q4;
    TIMG_M_MARK(tr_pro)
    if (TIMG_M_PRIO != 0) __builtin_amdgcn_s_setprio(TIMG_M_PRIO);
    issue_next_row(q0);
    issue_next_row(q1);
    issue_next_row(q2);
    if (kDepth >= 4) issue_next_row(q3);
    if (kDepth == 8) {
        issue_next_row(q4);
        issue_next_row(q5);
        issue_next_row(q6);
        issue_next_row(q7);
    }
#define TIMG_M_STEP(Q, K)                                                              \
    if (left < K + 1) break;                                                           \
    asm volatile("s_waitcnt vmcnt(%1) ; ring %0" : "+v"(Q) : "n"(kDepth - 1) : "memory"); \
    TIMG_M_MARK(tr_wait)                                                               \
    if (!row_step(make_uint4(Q.x, Q.y, Q.z, Q.w), 0)) return;                          \
    issue_next_row(Q);
    for (int left = r1 - bi.r0 + 1; left > 0; left -= kDepth) {  // (rows still to do)
        TIMG_M_STEP(q0, 0)
        TIMG_M_STEP(q1, 1)
        TIMG_M_STEP(q2, 2)
        if (kDepth >= 4) {
            TIMG_M_STEP(q3, 3)
        }
        if (kDepth == 8) {
            TIMG_M_STEP(q4, 4)
            TIMG_M_STEP(q5, 5)
            TIMG_M_STEP(q6, 6)
            TIMG_M_STEP(q7, 7)
        }
    }
    // (loads still in flight must land before their registers mean anything else)
    asm volatile("s_waitcnt vmcnt(0) ; ring all" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : : "memory");
#undef TIMG_M_STEP
    if (M == kOpaque) ok = ok && (amin >> 24) == 0xffu;
    if (__any(!ok) && (tid & 63) == 0) fail = 1;
    BlockSync();
    if (fail == 0 && tid == 0) tile_state[tile] = gen;
#ifdef TIMG_M_TRACE
    if ((tid & 63) == 0) {
        atomicAdd(&g_mtrace[0], tr_wait);
        atomicAdd(&g_mtrace[1], tr_vert);
        atomicAdd(&g_mtrace[2], tr_stage);
        atomicAdd(&g_mtrace[3], tr_horiz);
        atomicAdd(&g_mtrace[4], tr_bar2);
        atomicAdd(&g_mtrace[5], 1ull);
        atomicAdd(&g_mtrace[6], tr_pro);
        atomicAdd(&g_mtrace[7], (unsigned long long)__builtin_readcyclecounter() - t_begin);
    }
#endif
}

template <int M, bool kOvf>
__global__ void __launch_bounds__(kThreads)
    __attribute__((amdgpu_waves_per_eu(MKernelShape<M, kOvf>::kWaves, MKernelShape<M, kOvf>::kWaves)))
ScaleStreamMKernel(DevPlan plan, StreamTables tab, MTables mt, DevBlend blend, FrameBatch batch,
                   int *tile_state, int gen, int hrow, int hgroups) {
    ScaleStreamMBody<M, kOvf>(plan, tab, mt, blend, batch, tile_state, gen, hrow, hgroups);
}

// ===================================================================================
// Horizontal-first plans (what stb picks e.g. for 8K -> 800x450 and 640x480 -> 67x50):
// every source row is first gathered horizontally to the output width, the vertical
// filter then runs over those rows in source-row order (stb's scatter / gather loops).
//
// One workgroup = ONE WAVE = one (strip of <= 32 output columns, band of output rows, frame),
// a lane pair per output column.  Per source row:
//   * the lanes decode the strip's source window (raw RGBA8 rows are prefetched two rows
//     ahead with 16-byte loads) into the wave's LDS row buffer as float4 pixels;
//   * no barrier (a wave's LDS operations execute in order: the gather's reads see the
//     decoder's writes, and the next row's writes land behind this row's reads); the two
//     lanes of a pair gather stb's even and odd tap chains (weights live in registers, taps
//     past the column's own count have weight 0 and read finite data);
//   * the row's value feeds the <= kSlots output rows whose vertical filter covers it
//     (running sums in registers, RowSched as in the vertical-first kernel); a completed
//     output pixel is un-weighted, composed and stored straight from registers.
// No staging of results, no barriers -- waves never wait for each other, only for their own
// loads -- and the expensive part, taps x channels multiply-adds, is spread over all lanes.
// Channel sets as above; the opaque set carries A == 1.0f as a fourth channel so that the
// alpha chain is computed by the same packed arithmetic (three channels would cost the
// same number of instructions).
constexpr int kColsH    = 32;    // output columns per workgroup = ONE WAVE
constexpr int kThreadsH = 64;    // ... two lanes per column: stb's even and odd tap chains
constexpr int kWinMaxH  = 1024;  // source columns of a strip's window (multiple of 4)
#ifndef TIMG_H2_WAVES
#define TIMG_H2_WAVES 3
#endif

// One float4 per pixel in the row buffer: kOpaque (R, G, B, 1), kPremult (A, RA, GA, BA),
// kFull (R, G, B, A) -- its weighted channels RA GA BA are formed while gathering, by the
// same single multiplication the decoder would do.
template <int M>
__device__ __forceinline__ void DecodeToLds(uint32_t px, float *dst) {
    constexpr int kCh = ModeTraits<M>::kCh;
    float d[kCh];
    DecodeMode<M>(px, d);
    // (premultiplied set: RA GA BA A -- the colour pair comes out of one packed multiply into an even-aligned
    // register pair of the 16-byte store; with A in front the compiler moved every pair by one register)
    if (M == kPremult)
        *reinterpret_cast<float4 *>(dst) = make_float4(d[1], d[2], d[3], d[0]);
    else
        *reinterpret_cast<float4 *>(dst) = make_float4(d[0], d[1], d[2], M == kOpaque ? 1.0f : d[kCh > 3 ? 3 : 0]);
}

// value of the other lane of the pair (lanes 2i and 2i+1)
__device__ __forceinline__ float FromPartner(float v) {
    // (mov_dpp: no 'old' value -- every lane has a source under quad_perm -- so no register is zeroed first and the
    // exchange folds into the add that consumes it)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
}

// TAPS = taps per LANE (the column's taps 2j + parity): the loop is unrolled with its
// weights in registers.
// (three waves per SIMD for the opaque and the premultiplied set: the opaque one came out at 170 registers --
// two over the budget of three waves -- and ran at two; the 40-tap instantiations spill at 168 and stay at two)
// LOADS: 16-byte loads per lane and source row -- 2 for windows of up to 512 source columns (8K -> 800: 350), 4 up to
// kWinMaxH.  The raw rows travel through a register ring filled by inline assembly with hand-placed waits, as in the
// matrix kernel above and for the same reason, only worse: written as C++ (a conditional 16-byte load per chunk) every
// load came out behind an s_waitcnt vmcnt(0) -- no row was ever in flight while another was processed, a wave paid
// the whole memory round trip twice per source row (8K -> 800x450: 3.3 us per row and wave, 130 us per frame, 13 % of
// the bus).  Now every lane issues its LOADS loads unconditionally (a chunk outside the strip's window re-reads the
// window's last chunk: same cache line, no extra traffic; only the DECODE is predicated), kDepthH rows ahead, and a
// row is released by s_waitcnt vmcnt((kDepthH - 1) * LOADS): loads return in order.  check_ring_isa.py proves on the
// generated code that nothing touches a register set between its load and its wait.
template <int M, int TAPS, int LOADS>
__global__ void __launch_bounds__(kThreadsH) __attribute__((amdgpu_waves_per_eu((M == kFull || TAPS > 20) ? 2 : 3, (M == kFull || TAPS > 20) ? 2 : 3)))
ScaleStreamHKernel(DevPlan plan, StreamTables tab, DevBlend blend, FrameBatch batch, int *tile_state,
                   int gen, int win, int w4, int tile_pair) {
    static_assert(LOADS == 2 || LOADS == 4, "two or four 16-byte loads per lane and row");
    // One row buffer of 4 planes x w4 pixels: pixel n of the window lives in plane n & 3 at
    // index n >> 2.  The decoder's lanes hold 4 consecutive pixels each, so plane q is written
    // by consecutive lanes at consecutive 16-byte slots (no bank conflicts; a linear layout
    // made every write 4-way conflicted and the LDS the bottleneck); w4 = 4 mod 16 keeps the
    // planes 16 banks apart for the gather's reads.  Pixels >= win stay zero (padded taps).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kStride = 4;  // floats per pixel in the row buffer
    constexpr int kHc     = M == kFull ? 7 : 4;   // channels of the horizontal gather
    constexpr int kVc     = M == kFull ? 4 : 2;   // channels a lane carries through the vertical pass
    // tile_pair: this launch is the LAST of a chain whose earlier kernels (ScaleStreamH2Kernel) tiled the frame in strips
    // of twice the width -- strips 2k and 2k + 1 here are the halves of their strip k: their tile's state is read, none written
    const int tile = tile_pair ? (blockIdx.z * gridDim.y + blockIdx.y) * (gridDim.x >> 1) + (blockIdx.x >> 1)
                               : (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tile_state[tile] == gen) return;  // done by an earlier kernel of this call (uniform: whole workgroup leaves)
    const StripInfo si    = LoadConstant(tab.strips + blockIdx.x);
    if (si.ox0 >= si.ox1) return;  // (the empty second half of a narrow last strip)
    const BandInfo bi     = LoadConstant(tab.bands + blockIdx.y);
    const RowSched *sched = tab.sched + bi.sched;
    const int f           = blockIdx.z;
    const int tid         = threadIdx.x;
    const int par         = tid & 1;             // 0: even taps / low channels, 1: odd taps / high channels
    const int buf_floats  = 4 * w4 * kStride;    // floats of the row buffer
    for (int i = tid; i < buf_floats / 4; i += kThreadsH)
        reinterpret_cast<float4 *>(lds)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto slot = [&](int n) -> int { return ((n & 3) * w4 + (n >> 2)) * kStride; };  // float index of pixel n

    // this lane's output column: tap window and its half of the weights (in registers)
    const int ox   = si.ox0 + (tid >> 1);
    const bool has = ox < si.ox1;
    int2 ht        = make_int2(si.cx0, 0);
    if (has) ht = plan.h_taps[ox];
    const int n0l = ht.x - si.cx0;
    float hw[TAPS];
    {
        // <= 3 taps: stb runs ONE chain over them -- lane 0 of the pair takes them all
        const float *hc = plan.h_coeff + (size_t)ox * plan.h_width;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const int k = plan.h_sequential ? j : 2 * j + par;
            hw[j]       = (has && k < ht.y && !(plan.h_sequential && par)) ? hc[k] : 0.0f;
        }
    }
    const int tap_step = plan.h_sequential ? 1 : 2;
    const int tap_0    = plan.h_sequential ? 0 : par;
    int *flag = batch.transparent_flags ? batch.transparent_flags + f : nullptr;
    uint8_t *dst_frame = batch.dst + (size_t)f * batch.dst_frame_stride;

    // raw source rows: lane t owns the 4-pixel chunks t, t + 256, ... of the window
    const uint8_t *frame = batch.src + (size_t)f * batch.src_frame_stride;
    const int r_last     = min(bi.r1, plan.in_h - 1);
    const uint8_t *chunk_ptr[LOADS];  // this lane's chunks in row 0 of the frame
    bool chunk_in[LOADS];
    int chunk_shift[LOADS];  // a chunk straddling the end of the row: see the vertical-first kernel
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int c    = tid + j * kThreadsH;  // chunk index inside the window
        const int col  = si.cx0 + 4 * c;
        chunk_in[j]    = 4 * c < win;
        // (outside the window: the window's last chunk -- a cache line the wave reads anyway)
        chunk_ptr[j]   = frame + (size_t)min(min(col, si.cx0 + win - 4), plan.in_w - 4) * 4u;
        chunk_shift[j] = (col < plan.in_w && col + 4 > plan.in_w) ? col + 4 - plan.in_w : 0;
    }
    // (wave-uniform: only a wave of the last strip of a frame whose width is no multiple of 4 shuffles anything; as a
    // per-lane test the three-way shuffle cost every wave some 35 scalar and branch instructions per chunk and row)
    bool any_shift = false;
#pragma unroll
    for (int j = 0; j < LOADS; ++j) any_shift = any_shift || __any(chunk_shift[j] != 0);
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    // rows are fetched in order through a byte offset that just advances (rows past the image's last one --
    // virtual rows of the schedule -- re-read the last row); 64-bit: 8K frames with padded strides pass 4 GB... never,
    // but nothing here depends on it
    size_t next_off       = (size_t)min(bi.r0, r_last) * batch.src_stride;
    const size_t last_off = (size_t)r_last * batch.src_stride;
    // (the register sets are named variables handed over one by one, never arrays: an array of ring registers is an
    // object the compiler copies around -- check_ring_isa.py caught exactly such copies behind the first loads)
    auto issue_one = [&](u4v &q, int j) __attribute__((always_inline)) {
        const uint8_t *p = chunk_ptr[j] + next_off;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
    };
    auto advance_row = [&]() __attribute__((always_inline)) { next_off = min(next_off + batch.src_stride, last_off); };  // uniform

    // vertical sums of this lane's channels: gather channels par * kVc ... (kFull: 0-3 / 4-6)
    float acc[kSlots][kVc];
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int ch = 0; ch < kVc; ++ch) acc[s][ch] = 0.0f;
    uint32_t amin = 0xffffffffu;
    bool tiny     = false;
    // straight RGB of pixels with (nearly) no alpha only shows when they are not composed (as in the matrix kernel:
    // a composed pixel whose filtered alpha is below 2^-120 has alpha byte 0 and becomes the background alone)
    const bool need_straight = !(blend.enabled && blend.start_row <= bi.oy0);

    const uint32_t amin_limit = M == kOpaque ? 0xff000000u : need_straight ? 0x01000000u : 0u;

    RowSched rs_next = LoadConstant(sched);
    // one source row: decode -> LDS, barrier, gather, vertical update
    const RowSched *sched_ptr = sched;
    auto row_step = [&](const uint4 &r0, const uint4 &r1, const uint4 &r2, const uint4 &r3) __attribute__((always_inline)) -> bool {
        const uint4 raw[4] = {r0, r1, r2, r3};
        const RowSched rs = rs_next;
        asm volatile("" ::"s"(rs.flags[0]), "s"(rs.weight[0]));
        __builtin_amdgcn_sched_barrier(0);
        rs_next = LoadConstant(++sched_ptr);
        float *buf = lds;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            if (!chunk_in[j]) continue;
            uint4 q = raw[j];
            if (any_shift && chunk_shift[j]) {
                if (chunk_shift[j] == 1) q = make_uint4(q.y, q.z, q.w, q.w);
                else if (chunk_shift[j] == 2) q = make_uint4(q.z, q.w, q.w, q.w);
                else q = make_uint4(q.w, q.w, q.w, q.w);
            }
            // smallest pixel word so far (alpha is the top byte: its top byte is the smallest alpha)
            if (M != kFull) amin = MinU32(MinU32(MinU32(amin, q.x), q.y), MinU32(q.z, q.w));
            float *dst = buf + (size_t)(tid + j * kThreadsH) * kStride;  // index (chunk) in plane 0
            DecodeToLds<M>(q.x, dst);
            DecodeToLds<M>(q.y, dst + (size_t)w4 * kStride);
            DecodeToLds<M>(q.z, dst + (size_t)2 * w4 * kStride);
            DecodeToLds<M>(q.w, dst + (size_t)3 * w4 * kStride);
        }
        // opaque set: every alpha 0xff; premultiplied set: no alpha of 0 (unless composed anyway: need_straight)
        // (tiny: a filtered alpha below 2^-120, found by the row before this one -- see below.  ONE test and one
        // branch per row whatever the channel set: a second branch on need_straight made the compiler merge the
        // unrolled steps of the ring into a rotating loop that copies registers whose loads are still in flight)
        if (M != kFull && __any(amin < amin_limit || tiny)) return false;  // (the workgroup is this one wave)

        // this lane's chain of the horizontal gather: taps n = nb + 2j alternate between two
        // planes, each advancing one slot every second tap
        const int nb      = n0l + tap_0;
        const float *b_ev = buf + slot(nb), *b_od = buf + slot(nb + 2);
        float sum[kHc];
#pragma unroll
        for (int ch = 0; ch < kHc; ++ch) sum[ch] = 0.0f;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            float v[kHc];
            const float *src = tap_step == 2 ? ((j & 1) ? b_od : b_ev) + (size_t)(j >> 1) * kStride
                                             : buf + slot(nb + j);  // (<= 3 taps: one chain)
            const float4 t0 = *reinterpret_cast<const float4 *>(src);
            v[0] = t0.x;
            v[1] = t0.y;
            v[2] = t0.z;
            v[3] = t0.w;
            if (kHc > 4) {  // kFull: R*A, G*A, B*A as the decoder forms them
                v[kHc > 4 ? 4 : 0] = t0.x * t0.w;
                v[kHc > 5 ? 5 : 0] = t0.y * t0.w;
                v[kHc > 6 ? 6 : 0] = t0.z * t0.w;
            }
#pragma unroll
            for (int ch = 0; ch < kHc; ++ch) sum[ch] = sum[ch] + v[ch] * hw[j];
        }
        // even chain + odd chain (one of them is all zeros for <= 3 taps); every lane keeps
        // only the channels it carries through the vertical pass
        float h[kVc];
#pragma unroll
        for (int ch = 0; ch < kVc; ++ch) {
            // channel index in the gather: lane 0 carries 0..kVc-1, lane 1 the rest
            const int lo = ch, hi = kVc + ch;
            const float mine_lo = sum[lo], mine_hi = sum[hi < kHc ? hi : 0];
            const float oth_lo = FromPartner(mine_lo), oth_hi = FromPartner(mine_hi);
            // (sum of the even-tap lane's chain and the odd-tap lane's chain: commutative)
            h[ch] = par ? (hi < kHc ? mine_hi + oth_hi : 0.0f) : mine_lo + oth_lo;
        }

        // vertical: feed the active output rows, finish the one that completes
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int fl = rs.flags[s];
            if (!(fl & 1)) continue;  // wave-uniform
            const float w = rs.weight[s];
#pragma unroll
            for (int ch = 0; ch < kVc; ++ch) acc[s][ch] = acc[s][ch] + h[ch] * w;
            if (fl & 4) {
                // lane 0 of the pair assembles the pixel: its own channels + the partner's
                float all[2 * kVc];
#pragma unroll
                for (int ch = 0; ch < kVc; ++ch) {
                    all[ch]       = acc[s][ch];
                    all[kVc + ch] = FromPartner(acc[s][ch]);
                }
                Px7 px;
                if (M == kOpaque) {  // with alpha == 1 the straight and the weighted sums coincide
                    px.c[0] = all[0];
                    px.c[1] = all[1];
                    px.c[2] = all[2];
                    px.c[3] = all[3];
                    px.c[4] = all[0];
                    px.c[5] = all[1];
                    px.c[6] = all[2];
                } else if (M == kPremult) {
                    px.c[0] = px.c[1] = px.c[2] = 0.0f;
                    px.c[3] = all[3];  // (row buffer order: RA GA BA A)
                    px.c[4] = all[0];
                    px.c[5] = all[1];
                    px.c[6] = all[2];
                    // a filtered alpha below 2^-120 needs the straight RGB sums (noted here, acted upon by the next
                    // row's check)
                    if (need_straight && has && !par && px.c[3] < TIMG_TINY_F32) tiny = true;
                } else {
#pragma unroll
                    for (int ch = 0; ch < 7; ++ch) px.c[ch] = all[ch];
                }
                const int y = fl >> 8;
                if (has && !par) {
                    const uint32_t out = FinishStreamPixel(px, ox, y, plan.swap_rb, blend, flag);
                    *reinterpret_cast<uint32_t *>(dst_frame + (size_t)y * batch.dst_stride + (size_t)ox * 4) = out;
                }
#pragma unroll
                for (int ch = 0; ch < kVc; ++ch) acc[s][ch] = 0.0f;
            }
        }
        return true;
    };

    // kDepthH rows in flight per lane (register sets named, not rotated: see the matrix kernel)
    constexpr int kDepthH = LOADS == 2 ? 4 : 2;
    const u4v z4 = {0, 0, 0, 0};
    // set k = (qka, qkb) for LOADS == 2, sets 0..3; (qka, qkb, qkc, qkd) for LOADS == 4, sets 0..1
    u4v q0a, q0b, q1a, q1b, q2a = z4, q2b = z4, q3a = z4, q3b = z4, q0c = z4, q0d = z4, q1c = z4, q1d = z4;
    u4v q2c = z4, q2d = z4, q3c = z4, q3d = z4;  // (named by the macros' discarded branches only)
#define TIMG_H_ISSUE(K)                                   \
    issue_one(q##K##a, 0);                                \
    issue_one(q##K##b, 1);                                \
    if constexpr (LOADS == 4) {                           \
        issue_one(q##K##c, 2);                            \
        issue_one(q##K##d, 3);                            \
    }                                                     \
    advance_row();
#define TIMG_H_STEP(K)                                                                                                   \
    if (left < K + 1) break;                                                                                             \
    if constexpr (LOADS == 2)                                                                                            \
        asm volatile("s_waitcnt vmcnt(%2) ; ring %0 %1" : "+v"(q##K##a), "+v"(q##K##b) : "n"((kDepthH - 1) * LOADS) : "memory"); \
    else                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(%4) ; ring %0 %1 %2 %3"                                                            \
                     : "+v"(q##K##a), "+v"(q##K##b), "+v"(q##K##c), "+v"(q##K##d)                                        \
                     : "n"((kDepthH - 1) * LOADS)                                                                        \
                     : "memory");                                                                                        \
    if (!row_step(make_uint4(q##K##a.x, q##K##a.y, q##K##a.z, q##K##a.w), make_uint4(q##K##b.x, q##K##b.y, q##K##b.z, q##K##b.w), \
                  make_uint4(q##K##c.x, q##K##c.y, q##K##c.z, q##K##c.w), make_uint4(q##K##d.x, q##K##d.y, q##K##d.z, q##K##d.w))) \
        return;                                                                                                          \
    TIMG_H_ISSUE(K)
    TIMG_H_ISSUE(0)
    TIMG_H_ISSUE(1)
    if constexpr (kDepthH == 4) {
        TIMG_H_ISSUE(2)
        TIMG_H_ISSUE(3)
    }
    for (int left = bi.r1 - bi.r0 + 1; left > 0; left -= kDepthH) {  // (rows still to do)
        TIMG_H_STEP(0)
        TIMG_H_STEP(1)
        if constexpr (kDepthH == 4) {
            // (sets 2 and 3 exist for LOADS == 2 only: their c / d halves are the zero constants)
            if (left < 3) break;
            asm volatile("s_waitcnt vmcnt(%2) ; ring %0 %1" : "+v"(q2a), "+v"(q2b) : "n"((kDepthH - 1) * LOADS) : "memory");
            if (!row_step(make_uint4(q2a.x, q2a.y, q2a.z, q2a.w), make_uint4(q2b.x, q2b.y, q2b.z, q2b.w), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0))) return;
            issue_one(q2a, 0);
            issue_one(q2b, 1);
            advance_row();
            if (left < 4) break;
            asm volatile("s_waitcnt vmcnt(%2) ; ring %0 %1" : "+v"(q3a), "+v"(q3b) : "n"((kDepthH - 1) * LOADS) : "memory");
            if (!row_step(make_uint4(q3a.x, q3a.y, q3a.z, q3a.w), make_uint4(q3b.x, q3b.y, q3b.z, q3b.w), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0))) return;
            issue_one(q3a, 0);
            issue_one(q3b, 1);
            advance_row();
        }
    }
    // (loads still in flight must land before their registers mean anything else)
    asm volatile("s_waitcnt vmcnt(0) ; ring all"
                 : "+v"(q0a), "+v"(q0b), "+v"(q1a), "+v"(q1b), "+v"(q2a), "+v"(q2b), "+v"(q3a), "+v"(q3b), "+v"(q0c), "+v"(q0d), "+v"(q1c), "+v"(q1d)
                 :
                 : "memory");
#undef TIMG_H_STEP
#undef TIMG_H_ISSUE
    if (M == kPremult && __any(tiny)) return;  // (found by the last row)
    if (tid == 0 && !tile_pair) tile_state[tile] = gen;  // (any other row that broke the channel set's assumption has left through row_step)
}


// ===================================================================================
// Horizontal-first plans whose neighbouring columns share most of their taps (round 6): TWO output columns per lane
// pair.  The kernel above reads 16 bytes of LDS per (column, tap) -- at 8K -> 800x450 (39 taps, columns 9.6 source
// pixels apart) 20 KB a wave and source row for a window of 5.5 KB, and its LDS pipe is busy 78 % of the time
// (profiles/r5/sq_counters_c5.txt); no layout of the row buffer takes the bank conflicts of that gather below a factor
// of 1.7 (48 source pixels are exactly 5 columns: scratch/sim_h_gather_banks.py, profiles/r6/h2_kernel.txt).  Here a
// lane of pair i walks ONE chain parity p of source pixels  n0(A) + p + 2j,  j = 0 .. JS + TAPS,  for column A = 2i
// AND for column B = 2i + 1, whose window starts d = n0(B) - n0(A) pixels further on: the pixel a step reads is tap
// 2j + p of A (steps 0 .. TAPS - 1) and tap 2j + p - d of B (steps JS .. JS + TAPS: TAPS + 1 slots, the first or the
// last of them padded with weight 0 according to the lane's own start (d + q - p) / 2 in {JS, JS + 1}).  B's taps met
// by this lane form ONE chain of B (even or odd, by the parity of d), in stb's order; the partner lane meets the
// other.  25 reads of 16 bytes serve what took 40 (two waves x 20); a wave carries 64 columns, its window 653 source
// pixels instead of 2 x 346: half the horizontal halo as well.  Per column the multiply-adds are those of the kernel
// above, operation for operation (stb_image_resize2.h:5801-6009).  After the gather the pair exchanges what the
// other lane needs: lane 0 finishes column A, lane 1 column B -- all four channels of its column through the vertical
// pass, a completed pixel straight from its own registers.
// Applicability (host: H2Applicable): more than 3 taps, every pair's d gives starts in {JS, JS + 1}, windows of 64
// columns within kWinMaxH.  The seven-channel set falls back to the kernel above on the two halves of every strip.
constexpr int kColsH2 = 64;  // output columns per workgroup = ONE WAVE: 32 lane pairs x 2 columns

template <int M, int TAPS, int JS, int LOADS>
__global__ void __launch_bounds__(kThreadsH) __attribute__((amdgpu_waves_per_eu(TIMG_H2_WAVES, TIMG_H2_WAVES)))
ScaleStreamH2Kernel(DevPlan plan, StreamTables tab, const int2 *pairs, DevBlend blend, FrameBatch batch, int *tile_state,
                    int gen, int win, int w4) {
    static_assert(M == kOpaque || M == kPremult, "the four-channel sets");
    static_assert(LOADS == 2 || LOADS == 3 || LOADS == 4, "two to four 16-byte loads per lane and row");
    static_assert(JS >= 1 && JS < TAPS, "the second column's window overlaps the first's");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kStride = 4;        // floats per pixel in the row buffer
    constexpr int kSteps  = JS + TAPS + 1;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tile_state[tile] == gen) return;  // done by an earlier kernel of this call (uniform: whole workgroup leaves)
    const StripInfo si    = LoadConstant(tab.strips + blockIdx.x);
    const BandInfo bi     = LoadConstant(tab.bands + blockIdx.y);
    const RowSched *sched = tab.sched + bi.sched;
    const int f           = blockIdx.z;
    const int tid         = threadIdx.x;
    const int par         = tid & 1;  // the chain parity this lane walks -- and: 0 finishes column A, 1 column B
    const int buf_floats  = 4 * w4 * kStride;
    for (int i = tid; i < buf_floats / 4; i += kThreadsH)
        reinterpret_cast<float4 *>(lds)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto slot = [&](int n) -> int { return ((n & 3) * w4 + (n >> 2)) * kStride; };  // float index of pixel n

    // the pair's columns (host table, 32 entries a strip): B = A + 1 where the two windows sit as the steps assume, or none
    // (-1: a column at a clamped edge, whose window starts where its neighbour's does, has the pair to itself)
    const int2 pr    = pairs[(size_t)blockIdx.x * (kColsH2 / 2) + (tid >> 1)];
    const int ox_a   = pr.x, ox_b = pr.y;
    const bool has_a = ox_a >= 0, has_b = ox_b >= 0;
    int2 ht_a        = make_int2(si.cx0, 0);
    if (has_a) ht_a = plan.h_taps[ox_a];
    int2 ht_b = make_int2(ht_a.x + 2 * JS, 0);
    if (has_b) ht_b = plan.h_taps[ox_b];
    const int d   = ht_b.x - ht_a.x;
    const int n0l = ht_a.x - si.cx0;
    float hw_a[TAPS], hw_b[TAPS + 1];
    {
        const float *hc_a = plan.h_coeff + (size_t)(has_a ? ox_a : 0) * plan.h_width;
        const float *hc_b = plan.h_coeff + (size_t)(has_b ? ox_b : 0) * plan.h_width;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const int k = 2 * j + par;
            hw_a[j]     = (has_a && k < ht_a.y) ? hc_a[k] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t <= TAPS; ++t) {
            const int k = par + 2 * (JS + t) - d;  // B's tap under the pixel of step JS + t
            hw_b[t]     = (has_b && k >= 0 && k < ht_b.y) ? hc_b[k] : 0.0f;
        }
    }
    // the column this lane finishes
    const int ox   = par ? ox_b : ox_a;
    const bool has = par ? has_b : has_a;
    int *flag = batch.transparent_flags ? batch.transparent_flags + f : nullptr;
    uint8_t *dst_frame = batch.dst + (size_t)f * batch.dst_frame_stride;

    // raw source rows: lane t owns the 4-pixel chunks t, t + 64, ... of the window (as in the kernel above)
    const uint8_t *frame = batch.src + (size_t)f * batch.src_frame_stride;
    const int r_last     = min(bi.r1, plan.in_h - 1);
    const uint8_t *chunk_ptr[LOADS];
    bool chunk_in[LOADS];
    int chunk_shift[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int c    = tid + j * kThreadsH;
        const int col  = si.cx0 + 4 * c;
        chunk_in[j]    = 4 * c < win;
        chunk_ptr[j]   = frame + (size_t)min(min(col, si.cx0 + win - 4), plan.in_w - 4) * 4u;
        chunk_shift[j] = (col < plan.in_w && col + 4 > plan.in_w) ? col + 4 - plan.in_w : 0;
    }
    bool any_shift = false;
#pragma unroll
    for (int j = 0; j < LOADS; ++j) any_shift = any_shift || __any(chunk_shift[j] != 0);
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    size_t next_off       = (size_t)min(bi.r0, r_last) * batch.src_stride;
    const size_t last_off = (size_t)r_last * batch.src_stride;
    auto issue_one = [&](u4v &q, int j) __attribute__((always_inline)) {
        const uint8_t *p = chunk_ptr[j] + next_off;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
    };
    auto advance_row = [&]() __attribute__((always_inline)) { next_off = min(next_off + batch.src_stride, last_off); };  // uniform

    float acc[kSlots][4];  // vertical sums of this lane's column, row-buffer channel order
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) acc[s][ch] = 0.0f;
    uint32_t amin = 0xffffffffu;
    bool tiny     = false;
    const bool need_straight  = !(blend.enabled && blend.start_row <= bi.oy0);
    const uint32_t amin_limit = M == kOpaque ? 0xff000000u : need_straight ? 0x01000000u : 0u;

    RowSched rs_next          = LoadConstant(sched);
    const RowSched *sched_ptr = sched;
    auto row_step = [&](const uint4 &r0, const uint4 &r1, const uint4 &r2, const uint4 &r3) __attribute__((always_inline)) -> bool {
        const uint4 raw[4] = {r0, r1, r2, r3};
        const RowSched rs  = rs_next;
        asm volatile("" ::"s"(rs.flags[0]), "s"(rs.weight[0]));
        __builtin_amdgcn_sched_barrier(0);
        rs_next    = LoadConstant(++sched_ptr);
        float *buf = lds;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            if (!chunk_in[j]) continue;
            uint4 q = raw[j];
            if (any_shift && chunk_shift[j]) {
                if (chunk_shift[j] == 1) q = make_uint4(q.y, q.z, q.w, q.w);
                else if (chunk_shift[j] == 2) q = make_uint4(q.z, q.w, q.w, q.w);
                else q = make_uint4(q.w, q.w, q.w, q.w);
            }
            amin = MinU32(MinU32(MinU32(amin, q.x), q.y), MinU32(q.z, q.w));
            float *dst = buf + (size_t)(tid + j * kThreadsH) * kStride;
            DecodeToLds<M>(q.x, dst);
            DecodeToLds<M>(q.y, dst + (size_t)w4 * kStride);
            DecodeToLds<M>(q.z, dst + (size_t)2 * w4 * kStride);
            DecodeToLds<M>(q.w, dst + (size_t)3 * w4 * kStride);
        }
        if (__any(amin < amin_limit || tiny)) return false;  // (the workgroup is this one wave)

        // Wave priority by phase (profiles/r6/scale_prio.txt): a wave in its tap chain -- dependent LDS reads and sums --
        // goes first at the SIMD's issue port, one that decodes its next row or runs the vertical sums yields: 2-2.5 % of
        // the launch (8K S-alpha: 3.83-3.87 -> 3.73-3.79 ms); decode high 1.6 %, three levels (chain 2, vertical 1) slower.
        __builtin_amdgcn_s_setprio(TIMG_H_PRIO_T);
        // the lane's chain over the window of BOTH columns: steps alternate between two planes of the row buffer
        const int nb      = n0l + par;
        const float *b_ev = buf + slot(nb), *b_od = buf + slot(nb + 2);
        f2v a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f}, b01 = {0.0f, 0.0f}, b23 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < kSteps; ++j) {
            const f4v t = LdsSlot(((j & 1) ? b_od : b_ev) + (size_t)(j >> 1) * kStride);
            if (j < TAPS) {
                a01 = a01 + f2v{t.x, t.y} * f2v{hw_a[j < TAPS ? j : 0], hw_a[j < TAPS ? j : 0]};
                a23 = a23 + f2v{t.z, t.w} * f2v{hw_a[j < TAPS ? j : 0], hw_a[j < TAPS ? j : 0]};
            }
            if (j >= JS) {
                b01 = b01 + f2v{t.x, t.y} * f2v{hw_b[j >= JS ? j - JS : 0], hw_b[j >= JS ? j - JS : 0]};
                b23 = b23 + f2v{t.z, t.w} * f2v{hw_b[j >= JS ? j - JS : 0], hw_b[j >= JS ? j - JS : 0]};
            }
        }
        // even chain + odd chain of a column = this lane's sum + the partner's (commutative): lane 0 takes A, lane 1 B --
        // each hands the other what it has of the other's column
        const f2v give01 = par ? a01 : b01, give23 = par ? a23 : b23;
        const f2v keep01 = par ? b01 : a01, keep23 = par ? b23 : a23;
        float h[4];
        h[0] = keep01.x + FromPartner(give01.x);
        h[1] = keep01.y + FromPartner(give01.y);
        h[2] = keep23.x + FromPartner(give23.x);
        h[3] = keep23.y + FromPartner(give23.y);

        __builtin_amdgcn_s_setprio(TIMG_H_PRIO_V);
        // vertical: feed the active output rows of this lane's column, finish the one that completes
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int fl = rs.flags[s];
            if (!(fl & 1)) continue;  // wave-uniform
            const float w = rs.weight[s];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) acc[s][ch] = acc[s][ch] + h[ch] * w;
            if (fl & 4) {
                Px7 px;
                if (M == kOpaque) {  // (R, G, B, 1): with alpha == 1 the straight and the weighted sums coincide
                    px.c[0] = acc[s][0];
                    px.c[1] = acc[s][1];
                    px.c[2] = acc[s][2];
                    px.c[3] = acc[s][3];
                    px.c[4] = acc[s][0];
                    px.c[5] = acc[s][1];
                    px.c[6] = acc[s][2];
                } else {  // (RA, GA, BA, A)
                    px.c[0] = px.c[1] = px.c[2] = 0.0f;
                    px.c[3] = acc[s][3];
                    px.c[4] = acc[s][0];
                    px.c[5] = acc[s][1];
                    px.c[6] = acc[s][2];
                    if (need_straight && has && px.c[3] < TIMG_TINY_F32) tiny = true;  // (acted upon by the next row's check)
                }
                const int y = fl >> 8;
                if (has) {
                    const uint32_t out = FinishStreamPixel(px, ox, y, plan.swap_rb, blend, flag);
                    *reinterpret_cast<uint32_t *>(dst_frame + (size_t)y * batch.dst_stride + (size_t)ox * 4) = out;
                }
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) acc[s][ch] = 0.0f;
            }
        }
        return true;
    };

    // rows in flight per lane: named register sets (see the kernels above), set k = (qka .. qk[LOADS])
#ifndef TIMG_H2_DEPTH
#define TIMG_H2_DEPTH 2  // (three rows in flight need 12 more registers than three waves a SIMD leave: spills behind vmcnt(0))
#endif
    constexpr int kDepth2 = LOADS <= 3 ? TIMG_H2_DEPTH : 2;  // (four rows of two loads: the compiler rotated the sets through copies -- check_ring_isa.py refused it)
    const u4v z4 = {0, 0, 0, 0};
    u4v q0a, q0b, q0c = z4, q0d = z4, q1a, q1b, q1c = z4, q1d = z4, q2a = z4, q2b = z4, q2c = z4, q2d = z4, q3a = z4, q3b = z4,
        q3c = z4, q3d = z4;
#define TIMG_H2_ISSUE(K)                                  \
    issue_one(q##K##a, 0);                                \
    issue_one(q##K##b, 1);                                \
    if constexpr (LOADS >= 3) issue_one(q##K##c, 2);      \
    if constexpr (LOADS >= 4) issue_one(q##K##d, 3);      \
    advance_row();
#define TIMG_H2_STEP(K)                                                                                                  \
    if (left < K + 1) break;                                                                                             \
    if constexpr (LOADS == 2)                                                                                            \
        asm volatile("s_waitcnt vmcnt(%2) ; ring %0 %1" : "+v"(q##K##a), "+v"(q##K##b) : "n"((kDepth2 - 1) * LOADS) : "memory"); \
    else if constexpr (LOADS == 3)                                                                                       \
        asm volatile("s_waitcnt vmcnt(%3) ; ring %0 %1 %2" : "+v"(q##K##a), "+v"(q##K##b), "+v"(q##K##c)                 \
                     : "n"((kDepth2 - 1) * LOADS) : "memory");                                                           \
    else                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(%4) ; ring %0 %1 %2 %3"                                                            \
                     : "+v"(q##K##a), "+v"(q##K##b), "+v"(q##K##c), "+v"(q##K##d)                                        \
                     : "n"((kDepth2 - 1) * LOADS) : "memory");                                                           \
    if (!row_step(make_uint4(q##K##a.x, q##K##a.y, q##K##a.z, q##K##a.w), make_uint4(q##K##b.x, q##K##b.y, q##K##b.z, q##K##b.w), \
                  make_uint4(q##K##c.x, q##K##c.y, q##K##c.z, q##K##c.w), make_uint4(q##K##d.x, q##K##d.y, q##K##d.z, q##K##d.w))) \
        return;                                                                                                          \
    TIMG_H2_ISSUE(K)
    TIMG_H2_ISSUE(0)
    TIMG_H2_ISSUE(1)
    if constexpr (kDepth2 >= 3) {
        TIMG_H2_ISSUE(2)
    }
    if constexpr (kDepth2 >= 4) {
        TIMG_H2_ISSUE(3)
    }
    for (int left = bi.r1 - bi.r0 + 1; left > 0; left -= kDepth2) {  // (rows still to do)
        TIMG_H2_STEP(0)
        TIMG_H2_STEP(1)
        if constexpr (kDepth2 >= 3) {
            TIMG_H2_STEP(2)
        }
        if constexpr (kDepth2 >= 4) {
            TIMG_H2_STEP(3)
        }
    }
    // (loads still in flight must land before their registers mean anything else)
    asm volatile("s_waitcnt vmcnt(0) ; ring all"
                 : "+v"(q0a), "+v"(q0b), "+v"(q0c), "+v"(q0d), "+v"(q1a), "+v"(q1b), "+v"(q1c), "+v"(q1d), "+v"(q2a), "+v"(q2b),
                   "+v"(q2c), "+v"(q2d), "+v"(q3a), "+v"(q3b), "+v"(q3c), "+v"(q3d)
                 :
                 : "memory");
#undef TIMG_H2_STEP
#undef TIMG_H2_ISSUE
    if (M == kPremult && __any(tiny)) return;  // (found by the last row)
    if (tid == 0) tile_state[tile] = gen;
}

}  // namespace

// ---- host side: applicability + schedule ------------------------------------------
static bool H2Instantiated(int taps_lane, int js);
struct StreamVariant {
    void *device   = nullptr;
    StreamTables t = {};
    MTables m      = {};     // matrix-slot schedule (ScaleStreamMKernel), same indexing as t.sched
    bool m_ok      = false;  // ... exists for this plan
    bool m_ovf     = false;  // ... and ever uses the overflow row
    int band_rows  = 0;
};

struct StreamSchedule {
    // [0]: tall bands (little halo) for batches that fill the chip anyway,
    // [1]: short bands so that a single frame still yields ~500 workgroups
    StreamVariant v[2];
    int hrow        = 0;        // widest strip, in output columns
    bool hfirst     = false;    // horizontal-first plan: ScaleStreamHKernel
    int hwin        = 0;        // ... widest source window of a strip (multiple of 4)
    // ScaleStreamH2Kernel (two columns per lane pair) serves the plan's opaque and premultiplied channel sets: its strips
    // (up to 64 columns; the strips of the tables above are then their halves, for the seven-channel fallback), the
    // widest window and the lanes' common first step of the second column
    bool h2               = false;
    StripInfo *h2_strips  = nullptr;  // device
    int2 *h2_pairs        = nullptr;  // device: [strip][32] {column A, column B or -1} ({-1, -1}: an idle pair)
    int h2_n_strips       = 0;
    int h2_win            = 0;
    int h2_js             = 0;
    // Tile bookkeeping of a scale call, per SLOT: calls of one scaler that may be in flight at the same time (the
    // pieces of timg_hip_scale_sixel_encode, each on its own stream) use different slots.
    static constexpr int kTileSlots = 4;
    int *tile_state[kTileSlots] = {nullptr, nullptr, nullptr, nullptr};  // device, grown on demand
    int gen[kTileSlots]         = {0, 0, 0, 0};  // generation of the slot's current call (tile_state holds generations)
    size_t tile_cap[kTileSlots] = {0, 0, 0, 0};
    int slot                    = 0;             // the slot of the call being launched (host-side plumbing)
    // Further band heights, built when a batch size asks for one (LaunchScaleStream): a batch should fill the chip's
    // workgroup slots ONCE with equally tall bands -- 64 frames of 3840x2160 -> 200x56 as two bands of 45 + 11 rows a
    // frame were 512 tiles of which the 256 tall ones set the time, on 1024 slots (BASELINE config 3: 0.83 ms).
    std::vector<StripInfo> strips_host;
    std::vector<int> first_host, last_host;
    std::map<int, StreamVariant> more;           // by band_rows
    static constexpr size_t kMoreVariants = 6;
    // slot, gen and the tile tables are written while a call is being LAUNCHED: two host threads that launch on the
    // same scaler take turns here (include/timg_hip.h: what concurrent use of one scaler means)
    std::mutex launch_mu;
};

static bool BuildVariant(const ResamplePlan &p, const std::vector<StripInfo> &strips,
                         const std::vector<int> &first, const std::vector<int> &last,
                         int band_rows, StreamVariant *out) {
    std::vector<BandInfo> bands;
    std::vector<RowSched> sched;
    std::vector<RowW> wrows;    // matrix-slot schedule, indexed like sched
    std::vector<RowCtl> crows;
    std::vector<uint32_t> alpha_vals;  // distinct vertical alpha sums (bit patterns), in order of first use
    std::vector<float> keeps;  // [row][kMSlots]
    bool m_ok = p.vertical_first, uses_ovf = false;
    for (int oy = 0; oy < p.out_h; oy += band_rows) {
        BandInfo b;
        memset(&b, 0, sizeof(b));
        b.oy0   = oy;
        b.oy1   = std::min(p.out_h, oy + band_rows);
        b.r0    = first[b.oy0];
        // The kernel stages at most one completed output row per source row.  Where the
        // clamped edge makes two rows end on the same source row (the last two rows of a
        // frame), the later one is completed one row further down by a tap of weight 0 on a
        // virtual source row (it re-reads the last real row): acc + d * 0 == acc, so the sum
        // is untouched.  comp[y] = source row on which output row y is staged.
        std::vector<int> comp(b.oy1 - b.oy0);
        for (int y = b.oy0; y < b.oy1; ++y) {
            int c = last[y];
            if (y > b.oy0) c = std::max(c, comp[y - 1 - b.oy0] + 1);
            comp[y - b.oy0] = c;
        }
        b.r1    = comp.back();
        b.sched = (int)sched.size();
        RowSched blank;
        memset(&blank, 0, sizeof(blank));
        sched.resize(sched.size() + (size_t)(b.r1 - b.r0 + 1), blank);
        for (int y = b.oy0; y < b.oy1; ++y) {
            const VRun &r = p.v_runs[y];
            // vertical chain of an all-opaque column: 1.0f * w accumulated in order
            float alpha = 0.0f;
            for (int j = 0; j < r.count; ++j) {
                const float w = p.v_coeff[r.first + j];
                alpha         = j == 0 ? 1.0f * w : alpha + 1.0f * w;
            }
            const int s      = y % kSlots;
            const int done_r = comp[y - b.oy0];
            for (int j = 0; j < r.count; ++j) {
                const int row = p.v_rows[r.first + j];
                RowSched &e   = sched[(size_t)b.sched + (size_t)(row - b.r0)];
                e.weight[s]   = p.v_coeff[r.first + j];
                e.flags[s]    = 1 | (y << 8);
            }
            RowSched &e = sched[(size_t)b.sched + (size_t)(done_r - b.r0)];
            if (!(e.flags[s] & 1)) {  // virtual tap of weight 0
                e.weight[s] = 0.0f;
                e.flags[s]  = 1 | (y << 8);
            }
            e.flags[s] |= 4;
            e.alpha_sum[s] = alpha;
        }
        // The same band for the matrix-slot kernel: four slots whose products come from one
        // MFMA plus one overflow row on the VALU.  A row starts in a free matrix slot, or in
        // the overflow slot when all four are taken, and moves from there into the slot of
        // the next row that completes.
        {
            RowW wblank;
            RowCtl cblank;
            memset(&wblank, 0, sizeof(wblank));
            memset(&cblank, 0, sizeof(cblank));
            const size_t base = wrows.size();
            wrows.resize(base + (size_t)(b.r1 - b.r0 + 1), wblank);
            crows.resize(base + (size_t)(b.r1 - b.r0 + 1), cblank);
            keeps.resize((base + (size_t)(b.r1 - b.r0 + 1)) * kMSlots, 1.0f);
            int slot_y[kMSlots + 1];
            for (int &v : slot_y) v = -1;
            int next_y = b.oy0;
            for (int r = b.r0; r <= b.r1 && m_ok; ++r) {
                RowW &we   = wrows[base + (size_t)(r - b.r0)];
                RowCtl &ce = crows[base + (size_t)(r - b.r0)];
                while (next_y < b.oy1 && first[next_y] <= r) {
                    int k = 0;
                    while (k <= kMSlots && slot_y[k] >= 0) ++k;
                    if (k > kMSlots) {
                        m_ok = false;
                        break;
                    }
                    if (k < kMSlots) keeps[(base + (size_t)(r - b.r0)) * kMSlots + k] = 0.0f;  // the sum restarts here
                    slot_y[k] = next_y++;
                }
                for (int k = 0; k <= kMSlots && m_ok; ++k) {
                    const int y = slot_y[k];
                    if (y < 0) continue;
                    const VRun &run = p.v_runs[y];
                    float w         = 0.0f;  // (past the row's last tap: the virtual tap of weight 0)
                    for (int j = 0; j < run.count; ++j)
                        if (p.v_rows[run.first + j] == r) w = p.v_coeff[run.first + j];
                    if (k < kMSlots) {
                        we.w[k] = w;
                    } else {
                        ce.ovf_w = w;
                        ce.flags |= 1;
                        uses_ovf = true;
                    }
                }
                for (int k = 0; k <= kMSlots && m_ok; ++k) {
                    const int y = slot_y[k];
                    if (y < 0 || comp[y - b.oy0] != r) continue;
                    if (ce.flags & 0x70) {  // (never: comp[] is strictly increasing)
                        m_ok = false;
                        break;
                    }
                    const VRun &run = p.v_runs[y];
                    float alpha     = 0.0f;
                    for (int j = 0; j < run.count; ++j) {
                        const float w = p.v_coeff[run.first + j];
                        alpha         = j == 0 ? 1.0f * w : alpha + 1.0f * w;
                    }
                    ce.flags |= (k + 1) << 4;
                    ce.done_y    = y;
                    ce.alpha_sum = alpha;
                    {   // which row of the opaque set's alpha table (AlphaCell) this output row uses
                        uint32_t bits;
                        memcpy(&bits, &alpha, 4);
                        size_t at = 0;
                        while (at < alpha_vals.size() && alpha_vals[at] != bits) ++at;
                        if (at == alpha_vals.size()) alpha_vals.push_back(bits);
                        if (at > 0xffff) m_ok = false;
                        ce.flags |= (int)(at & 0xffff) << 8;
                    }
                    slot_y[k]    = -1;
                    if (k < kMSlots && slot_y[kMSlots] >= 0) {
                        ce.flags |= 0x80;
                        slot_y[k]       = slot_y[kMSlots];
                        slot_y[kMSlots] = -1;
                    }
                }
            }
            for (int v : slot_y)
                if (v >= 0) m_ok = false;  // (every row of the band completes inside the band)
            if (next_y != b.oy1) m_ok = false;
        }
        bands.push_back(b);
    }
    {
        RowSched blank;  // row_step reads one entry ahead
        memset(&blank, 0, sizeof(blank));
        sched.push_back(blank);
        RowW wblank;
        RowCtl cblank;
        memset(&wblank, 0, sizeof(wblank));
        memset(&cblank, 0, sizeof(cblank));
        wrows.push_back(wblank);
        crows.push_back(cblank);
        keeps.resize(keeps.size() + kMSlots, 1.0f);
    }
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_strips = 0;
    const size_t o_bands  = align(o_strips + strips.size() * sizeof(StripInfo));
    const size_t o_sched  = align(o_bands + bands.size() * sizeof(BandInfo));
    std::vector<RowRec> recs(wrows.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        recs[i].w   = wrows[i];
        recs[i].ctl = crows[i];
        for (int k = 0; k < kMSlots; ++k) recs[i].keep[k] = keeps[i * kMSlots + k];
    }
    const size_t o_recs   = align(o_sched + sched.size() * sizeof(RowSched));
    // The opaque set's alpha table: per distinct vertical alpha sum and output column, stb's horizontal chain of
    // HorizontalRowM evaluated on a row whose every column holds that sum -- even taps into one chain, odd taps
    // into the other (padded taps add v * 0: nothing), their sum at the end; one chain for <= 3 taps.
    std::vector<AlphaCell> atab;
    if (m_ok && alpha_vals.size() * (size_t)p.out_w * sizeof(AlphaCell) > ((size_t)32 << 20)) m_ok = false;
    if (m_ok) {
        atab.resize(alpha_vals.size() * (size_t)p.out_w);
        for (size_t d = 0; d < alpha_vals.size() && m_ok; ++d) {
            float a;
            memcpy(&a, &alpha_vals[d], 4);
            for (int x = 0; x < p.out_w; ++x) {
                const float *hc = &p.h_coeff[(size_t)x * p.h_width];
                const int n     = std::min((int)p.h_taps[x].count, p.h_width);
                float even = 0.0f, odd = 0.0f;
                for (int k = 0; k < n; ++k) {
                    const float prod = a * hc[k];
                    if (p.h_sequential || !(k & 1)) even = even + prod;
                    else odd = odd + prod;
                }
                const float alpha_h = p.h_sequential ? even : even + odd;
                if (!(alpha_h >= TIMG_TINY_F32)) {  // (the kernel's table path assumes stb's "alpha present" branch)
                    m_ok = false;
                    break;
                }
                AlphaCell c;
                c.inv     = 1.0f / alpha_h;
                float q   = alpha_h * 255.0f + 0.5f;
                q         = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
                c.byte_hi = (uint32_t)q << 24;
                atab[d * (size_t)p.out_w + x] = c;
            }
        }
    }
    const size_t o_atab   = align(o_recs + recs.size() * sizeof(RowRec));
    const size_t total    = align(o_atab + atab.size() * sizeof(AlphaCell) + 16);
    std::vector<char> host(total, 0);
    memcpy(&host[o_strips], strips.data(), strips.size() * sizeof(StripInfo));
    memcpy(&host[o_bands], bands.data(), bands.size() * sizeof(BandInfo));
    memcpy(&host[o_sched], sched.data(), sched.size() * sizeof(RowSched));
    memcpy(&host[o_recs], recs.data(), recs.size() * sizeof(RowRec));
    if (!atab.empty()) memcpy(&host[o_atab], atab.data(), atab.size() * sizeof(AlphaCell));
    void *dev = nullptr;
    if (DevMalloc(&dev, total) != hipSuccess) return false;
    if (hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        (void)DevFree(dev);
        return false;
    }
    out->device     = dev;
    out->t.strips   = (const StripInfo *)((char *)dev + o_strips);
    out->t.bands    = (const BandInfo *)((char *)dev + o_bands);
    out->t.sched    = (const RowSched *)((char *)dev + o_sched);
    out->t.n_strips = (int)strips.size();
    out->t.n_bands  = (int)bands.size();
    out->m.rec      = (const RowRec *)((char *)dev + o_recs);
    out->m.alpha_tab = (const AlphaCell *)((char *)dev + o_atab);
    out->m_ok       = m_ok && wrows.size() == sched.size();
    out->m_ovf      = uses_ovf;
    out->band_rows  = band_rows;
    return true;
}

bool PrepareStreamSchedule(timg_hip_scaler *s, std::string *why_not) {
    const ResamplePlan &p = s->plan;
    auto no = [&](const char *m) {
        if (why_not) *why_not = m;
        return false;
    };
    if (p.identity) return no("identity plan");
    if (!p.vertical_first && p.h_width > 80) return no("horizontal filter wider than 80 taps");
    if (p.max_active_rows > kSlots) return no("too many output rows per source row");
    if (p.in_w < kPix) return no("source narrower than one load");
    // slot = y % kSlots must be free again when row y + kSlots starts
    std::vector<int> first(p.out_h), last(p.out_h);
    for (int y = 0; y < p.out_h; ++y) {
        const VRun &r = p.v_runs[y];
        if (r.count <= 0) return no("empty vertical run");
        first[y] = p.v_rows[r.first];
        last[y]  = p.v_rows[r.first + r.count - 1];
        for (int j = 1; j < r.count; ++j)
            if (p.v_rows[r.first + j] <= p.v_rows[r.first + j - 1]) return no("rows not ascending");
    }
    {
        // the row an output row is staged on (BuildVariant spreads rows that end together;
        // per band it can only be earlier than this whole-frame bound)
        int comp = -1;
        for (int y = 0; y < p.out_h; ++y) {
            comp = std::max(last[y], comp + 1);
            if (y + kSlots < p.out_h && comp >= first[y + kSlots]) return no("slot reuse conflict");
        }
    }
    for (int y = 1; y < p.out_h; ++y)
        if (last[y] < last[y - 1] || first[y] < first[y - 1]) return no("non-monotonic rows");

    std::vector<StripInfo> strips, h2_wide;
    std::vector<int2> h2_pairs_host;
    int hwin = 0, h2_wide_win = 0, h2_js_min = 0;
    if (p.vertical_first) {
        // strips: as many output columns as fit with all their taps in kStripCols source columns
        for (int ox = 0; ox < p.out_w;) {
            StripInfo si;
            si.ox0  = ox;
            si.cx0  = p.h_taps[ox].n0 & ~(kPix - 1);
            si.pad  = 0;
            int end = ox;
            // (LDS keeps {taps, weights} per output column: bound the strip's outputs too)
            const int max_out = kLdsCoeffFloats / std::max(1, p.h_width);
            if (max_out < 1) return no("horizontal filter too wide for the LDS weight table");
            while (end < p.out_w && end - ox < max_out) {
                const HTaps &t = p.h_taps[end];
                if (t.n0 < si.cx0 || t.n0 + t.count > si.cx0 + kStripCols) break;
                ++end;
            }
            if (end == ox) return no("horizontal window wider than a strip");
            si.ox1 = end;
            strips.push_back(si);
            ox = end;
        }
    } else {
        // horizontal-first: a lane pair per output column, the strip's source window in LDS;
        // equally wide strips (a narrow last strip would cost a full walk over the rows)
        auto cut = [&](int cols_max, std::vector<StripInfo> *out, int *win_out) -> bool {
            const int n_even = (p.out_w + cols_max - 1) / cols_max;
            const int cols   = (p.out_w + n_even - 1) / n_even;
            for (int ox = 0; ox < p.out_w;) {
                StripInfo si;
                si.ox0  = ox;
                si.cx0  = p.h_taps[ox].n0 & ~3;
                si.pad  = 0;
                int end = ox, reach = si.cx0;
                while (end < p.out_w && end - ox < cols) {
                    const HTaps &t = p.h_taps[end];
                    if (t.n0 < si.cx0 || t.n0 + t.count > si.cx0 + kWinMaxH) break;
                    reach = std::max(reach, t.n0 + t.count);
                    ++end;
                }
                if (end == ox) return false;
                si.ox1   = end;
                *win_out = std::max(*win_out, (reach - si.cx0 + 3) & ~3);
                out->push_back(si);
                ox = end;
            }
            return true;
        };
        // Two columns per lane pair (ScaleStreamH2Kernel) where the plan allows: h2_strips.h
        const char *h2_env  = getenv("TIMG_HIP_H2");
        H2Tiling tiling;
        if (!(h2_env && h2_env[0] == '0')) tiling = BuildH2Tiling(p, kColsH2, kColsH, kWinMaxH, H2Instantiated);
        std::vector<StripInfo> wide;
        std::vector<int2> pairs;
        int wide_win = 0, js = 0;
        if (tiling.ok) {
            for (const H2Strip &w : tiling.strips) wide.push_back(StripInfo{w.ox0, w.ox1, w.cx0, 0});
            for (const H2Pair &e : tiling.pairs) pairs.push_back(make_int2(e.a, e.b));
            for (const H2Strip &w : tiling.halves) strips.push_back(StripInfo{w.ox0, w.ox1, w.cx0, 0});
            wide_win = tiling.win;
            hwin     = tiling.half_win;
            js       = tiling.js;
        } else if (!cut(kColsH, &strips, &hwin)) {
            return no("horizontal window wider than the row buffer");
        }
        h2_pairs_host = pairs;
        const int js_lo = js;
        h2_wide     = wide;
        h2_wide_win = wide_win;
        h2_js_min   = js_lo;
    }
    StreamSchedule *ss = new StreamSchedule();
    ss->hfirst = !p.vertical_first;
    ss->hwin   = hwin;
    for (const StripInfo &si : strips) ss->hrow = std::max(ss->hrow, si.ox1 - si.ox0);
    // Tall bands: 45 output rows for vertical-first plans (flat between 45 and 150: profiles/r5/band_rows.txt).  A
    // horizontal-first plan pays every band's vertical halo with a full horizontal pass over the halo's source rows (39
    // rows of 8K per band at 9.6:1): 90 rows -- 4.20 ms per 64 8K frames against 4.44 at 45 (profiles/r5/band_rows_c5.txt;
    // 113 and more leave the 768 workgroup slots of that kernel half empty in the last round).
    int tall = p.vertical_first ? 45 : !h2_wide.empty() ? 75 : 90;  // (two columns a lane pair: half as many, longer tiles -- 75 rows fill the slots better: profiles/r6/h2_kernel.txt)
    if (const char *e = getenv("TIMG_HIP_BAND_ROWS")) tall = atoi(e) > 0 ? atoi(e) : tall;  // tuning
    tall = std::max(1, std::min(p.out_h, tall));
    tall = (p.out_h + (p.out_h + tall - 1) / tall - 1) / ((p.out_h + tall - 1) / tall);  // (equally tall bands)
    ss->strips_host = strips;
    ss->first_host  = first;
    ss->last_host   = last;
    int fine           = tall;
    while (fine > 6 && (size_t)strips.size() * ((p.out_h + fine - 1) / fine) < 512)
        fine = std::max(6, fine / 2);
    bool uploaded = BuildVariant(p, strips, first, last, tall, &ss->v[0]) && BuildVariant(p, strips, first, last, fine, &ss->v[1]);
    if (uploaded && !h2_wide.empty()) {
        const size_t bytes = h2_wide.size() * sizeof(StripInfo);
        const size_t pbytes = h2_pairs_host.size() * sizeof(int2);
        uploaded = DevMalloc((void **)&ss->h2_strips, bytes) == hipSuccess &&
                   hipMemcpy(ss->h2_strips, h2_wide.data(), bytes, hipMemcpyHostToDevice) == hipSuccess &&
                   DevMalloc((void **)&ss->h2_pairs, pbytes) == hipSuccess &&
                   hipMemcpy(ss->h2_pairs, h2_pairs_host.data(), pbytes, hipMemcpyHostToDevice) == hipSuccess;
        ss->h2          = uploaded;
        ss->h2_n_strips = (int)h2_wide.size();
        ss->h2_win      = h2_wide_win;
        ss->h2_js       = h2_js_min;
    }
    if (!uploaded) {
        for (auto &v : ss->v)
            if (v.device) (void)DevFree(v.device);
        if (ss->h2_strips) (void)DevFree(ss->h2_strips);
        if (ss->h2_pairs) (void)DevFree(ss->h2_pairs);
        delete ss;
        return no("uploading the schedule failed");
    }
    if (getenv("TIMG_HIP_TRACE_SCHEDULE"))
        fprintf(stderr, "timg_hip schedule %dx%d -> %dx%d: %s, %zu strips, h2 %d (js %d, %d strips, window %d), hwin %d\n", p.in_w, p.in_h,
                p.out_w, p.out_h, p.vertical_first ? "vertical-first" : "horizontal-first", strips.size(), (int)ss->h2, ss->h2_js,
                ss->h2_n_strips, ss->h2_win, ss->hwin);
    s->stream_tables = ss;
    if (const char *e = getenv("TIMG_HIP_NO_MATRIX")) s->stream_cfg[4] = atoi(e) != 0;  // tuning: all-VALU kernels
    s->stream_cfg[0] = ss->v[0].t.n_strips;
    s->stream_cfg[1] = ss->v[0].t.n_bands;
    s->stream_cfg[2] = ss->v[1].t.n_bands;
    return true;
}

// bit 0: the matrix-core kernel serves this plan, bit 1: with the overflow row compiled in, bit 2: a horizontal-first plan
// served by the kernel with two columns a lane pair (ScaleStreamH2Kernel)
int StreamShapeBits(const timg_hip_scaler *s) {
    const StreamSchedule *ss = (const StreamSchedule *)s->stream_tables;
    if (!ss) return 0;
    return (ss->v[0].m_ok ? 1 : 0) | (ss->v[0].m_ovf || ss->v[1].m_ovf ? 2 : 0) | (ss->h2 ? 4 : 0);
}

void ReleaseStreamSchedule(timg_hip_scaler *s) {
    StreamSchedule *ss = (StreamSchedule *)s->stream_tables;
    if (!ss) return;
    for (auto &v : ss->v)
        if (v.device) (void)DevFree(v.device);
    for (auto &kv : ss->more)
        if (kv.second.device) (void)DevFree(kv.second.device);
    for (int *t : ss->tile_state)
        if (t) (void)DevFree(t);
    if (ss->h2_strips) (void)DevFree(ss->h2_strips);
    if (ss->h2_pairs) (void)DevFree(ss->h2_pairs);
    delete ss;
    s->stream_tables = nullptr;
}

template <int M>
static hipError_t LaunchMode(const timg_hip_scaler *s, const StreamSchedule *ss,
                             const StreamVariant &v, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream) {
    const size_t lds = ((size_t)ModeTraits<M>::kStage * kStripCols * ModeTraits<M>::kStride + 2 * (size_t)ss->hrow +
                        (size_t)s->plan.h_width * ss->hrow) * sizeof(float);
    // (once per instantiation and process; loader threads may get here concurrently: timg_hip.h promises them that)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, []() {
        attr_err = hipFuncSetAttribute((const void *)ScaleStreamKernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const dim3 grid(v.t.n_strips, v.t.n_bands, batch.n_frames);
    hipLaunchKernelGGL(ScaleStreamKernel<M>, grid, dim3(kThreads), lds, stream, s->dev, v.t, blend,
                       batch, ss->tile_state[ss->slot], ss->gen[ss->slot], ss->hrow);
    return hipGetLastError();
}

template <int M, bool kOvf>
static hipError_t LaunchModeMO(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                               const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    const int hgroups = (s->plan.h_width + 3) / 4;
    const size_t lds  = ((size_t)MKernelShape<M, kOvf>::kStage * (kStripCols + 4 * hgroups) * 4 + (size_t)hgroups * ss->hrow * 4) * sizeof(float) +
                       (size_t)ss->hrow * sizeof(int);
    // (once per instantiation and process; loader threads may get here concurrently: timg_hip.h promises them that)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, []() {
        attr_err = hipFuncSetAttribute((const void *)ScaleStreamMKernel<M, kOvf>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const dim3 grid(v.t.n_strips, v.t.n_bands, batch.n_frames);
    hipLaunchKernelGGL((ScaleStreamMKernel<M, kOvf>), grid, dim3(kThreads), lds, stream, s->dev, v.t, v.m, blend, batch,
                       ss->tile_state[ss->slot], ss->gen[ss->slot], ss->hrow, hgroups);
    return hipGetLastError();
}

// What the matrix kernel needs beyond the VALU kernel's own bounds: its staging rows and float4 weight groups in at
// most 128 KB of LDS (a plan with one- or two-tap rows and thousands of outputs per strip -- large upscales -- fits
// the VALU layout but not this one), and 32-bit row byte offsets (the kernel advances a uint32 through the frame).
template <int M, bool kOvf>
static bool MatrixKernelFits(const timg_hip_scaler *s, const StreamSchedule *ss, const FrameBatch &batch) {
    const int hgroups = (s->plan.h_width + 3) / 4;
    const size_t lds  = ((size_t)MKernelShape<M, kOvf>::kStage * (kStripCols + 4 * hgroups) * 4 + (size_t)hgroups * ss->hrow * 4) * sizeof(float) +
                       (size_t)ss->hrow * sizeof(int);
    return lds + 64 <= (size_t)128 * 1024 && (size_t)s->plan.in_h * batch.src_stride < ((size_t)1 << 32);
}

template <int M>
static hipError_t LaunchModeM(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                              const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    if (!(v.m_ovf ? MatrixKernelFits<M, true>(s, ss, batch) : MatrixKernelFits<M, false>(s, ss, batch)))
        return LaunchMode<M>(s, ss, v, blend, batch, stream);  // same results, all-VALU
    return v.m_ovf ? LaunchModeMO<M, true>(s, ss, v, blend, batch, stream)
                   : LaunchModeMO<M, false>(s, ss, v, blend, batch, stream);
}

// which (taps per lane, first step of the second column) ScaleStreamH2Kernel exists for
// (17 to 40 taps, ratios 4.25 to 10; with THREE rows in flight first steps 2 and 3 compiled into rings whose sets the compiler
// rotated through copies -- check_ring_isa.py refused them; with two rows in flight every instantiation passes)
static bool H2Instantiated(int taps_lane, int js) { return taps_lane == 20 && js >= 2 && js <= 5; }

template <int M, int TAPS, int JS, int LOADS>
static hipError_t LaunchModeH2TJL(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                                  const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    // pixels per plane: a quarter of (window + the steps past it), rounded up to 4 mod 16
    int w4 = (ss->h2_win + 2 * (TAPS + JS + 1) + 3) / 4 + 1;
    while ((w4 & 15) != 4) ++w4;
    const size_t lds = (size_t)4 * w4 * 4 * sizeof(float);
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, []() {
        attr_err = hipFuncSetAttribute((const void *)ScaleStreamH2Kernel<M, TAPS, JS, LOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    StreamTables t = v.t;
    t.strips       = ss->h2_strips;
    t.n_strips     = ss->h2_n_strips;
    const dim3 grid(t.n_strips, t.n_bands, batch.n_frames);
    hipLaunchKernelGGL((ScaleStreamH2Kernel<M, TAPS, JS, LOADS>), grid, dim3(kThreadsH), lds, stream, s->dev, t, ss->h2_pairs, blend,
                       batch, ss->tile_state[ss->slot], ss->gen[ss->slot], ss->h2_win, w4);
    return hipGetLastError();
}
template <int M, int TAPS, int JS>
static hipError_t LaunchModeH2TJ(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                                 const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    const int chunks = (ss->h2_win / 4 + kThreadsH - 1) / kThreadsH;  // 16-byte loads per lane and row
    if (chunks <= 2) return LaunchModeH2TJL<M, TAPS, JS, 2>(s, ss, v, blend, batch, stream);
    if (chunks == 3) return LaunchModeH2TJL<M, TAPS, JS, 3>(s, ss, v, blend, batch, stream);
    return LaunchModeH2TJL<M, TAPS, JS, 4>(s, ss, v, blend, batch, stream);
}
template <int M>
static hipError_t LaunchModeH2(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                               const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    switch (ss->h2_js) {  // (H2Instantiated)
    case 2: return LaunchModeH2TJ<M, 20, 2>(s, ss, v, blend, batch, stream);
    case 3: return LaunchModeH2TJ<M, 20, 3>(s, ss, v, blend, batch, stream);
    case 4: return LaunchModeH2TJ<M, 20, 4>(s, ss, v, blend, batch, stream);
    default: return LaunchModeH2TJ<M, 20, 5>(s, ss, v, blend, batch, stream);
    }
}

template <int M, int TAPS, int LOADS>
static hipError_t LaunchModeHTL(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                                const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    // pixels per plane: a quarter of (window + padded taps), rounded up to 4 mod 16
    int w4 = (ss->hwin + 2 * TAPS + 3) / 4 + 1;
    while ((w4 & 15) != 4) ++w4;
    const size_t lds = (size_t)4 * w4 * 4 * sizeof(float);
    // (once per instantiation and process; loader threads may get here concurrently: timg_hip.h promises them that)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, []() {
        attr_err = hipFuncSetAttribute((const void *)ScaleStreamHKernel<M, TAPS, LOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const dim3 grid(v.t.n_strips, v.t.n_bands, batch.n_frames);
    hipLaunchKernelGGL((ScaleStreamHKernel<M, TAPS, LOADS>), grid, dim3(kThreadsH), lds, stream, s->dev, v.t, blend,
                       batch, ss->tile_state[ss->slot], ss->gen[ss->slot], ss->hwin, w4, ss->h2 ? 1 : 0);
    return hipGetLastError();
}

template <int M, int TAPS>
static hipError_t LaunchModeHT(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                               const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    // windows of up to 512 source columns need two 16-byte loads per lane and row (and keep four rows in flight)
    return ss->hwin <= 2 * 4 * kThreadsH ? LaunchModeHTL<M, TAPS, 2>(s, ss, v, blend, batch, stream)
                                         : LaunchModeHTL<M, TAPS, 4>(s, ss, v, blend, batch, stream);
}

template <int M>
static hipError_t LaunchModeH(const timg_hip_scaler *s, const StreamSchedule *ss, const StreamVariant &v,
                              const DevBlend &blend, const FrameBatch &batch, hipStream_t stream) {
    // the tap loop is unrolled with its weights in registers: smallest instantiation that fits
    // (a lane runs every second tap; <= 3 taps form one chain that lane 0 of the pair runs alone)
    const int taps = s->plan.h_width;
    if (taps <= 16) return LaunchModeHT<M, 8>(s, ss, v, blend, batch, stream);
    if (taps <= 40) return LaunchModeHT<M, 20>(s, ss, v, blend, batch, stream);
    // (ratios 10 to 14: 28 weights a lane fit two waves a SIMD without the spills of the 40-tap instantiation; TIMG_HIP_H28=0: comparison)
    static const bool h28 = !(getenv("TIMG_HIP_H28") && getenv("TIMG_HIP_H28")[0] == '0');
    if (taps <= 56 && h28) return LaunchModeHT<M, 28>(s, ss, v, blend, batch, stream);
    return LaunchModeHT<M, 40>(s, ss, v, blend, batch, stream);
}

hipError_t LaunchScaleStream(const timg_hip_scaler *s, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream, int slot) {
    StreamSchedule *ss = (StreamSchedule *)s->stream_tables;
    if (!ss || slot < 0 || slot >= StreamSchedule::kTileSlots) return hipErrorNotSupported;
    std::lock_guard<std::mutex> launching(ss->launch_mu);
    ss->slot = slot;
    // rows of whole pixels (the 16-byte loads only need 4-byte alignment); otherwise the
    // generic kernel runs
    if (((uintptr_t)batch.src & 3) || (batch.src_stride & 3) || (batch.src_frame_stride & 3))
        return LaunchScaleGeneric(s->dev, blend, batch, stream);
    // Band height: the tallest whose tiles fill the workgroup slots of the chip (four a CU), between the tall variant
    // (little halo: batches that fill the chip anyway) and 6-row bands (a single frame); equally tall bands
    const StreamVariant &tall = ss->v[0];
    const StreamVariant *pick = &tall;
    {
        const int out_h   = s->plan.out_h;
        const long slots  = 4L * std::max(1, s->ctx ? s->ctx->cu_count : 256);
        const long per_b  = (long)tall.t.n_strips * std::max(1, batch.n_frames);
        const long b_max  = std::max<long>(tall.t.n_bands, (out_h + 5) / 6);
        const long bands  = std::min(b_max, std::max<long>(tall.t.n_bands, (slots + per_b - 1) / per_b));
        const int rows    = (int)((out_h + bands - 1) / bands);
        if (rows >= tall.band_rows) {
            pick = &tall;
        } else if (rows <= ss->v[1].band_rows) {
            pick = &ss->v[1];
        } else {
            auto it = ss->more.find(rows);
            if (it == ss->more.end() && ss->more.size() < StreamSchedule::kMoreVariants) {
                StreamVariant nv;
                if (BuildVariant(s->plan, ss->strips_host, ss->first_host, ss->last_host, rows, &nv))
                    it = ss->more.emplace(rows, nv).first;
                else
                    (void)hipGetLastError();  // (a failed allocation of the EXTRA variant is not this launch's error: the
                                              // launch goes on with a variant that exists, and LaunchMode* below read
                                              // hipGetLastError() right behind their kernel -- ADVICE r4)
            }
            if (it != ss->more.end()) {
                pick = &it->second;
            } else {  // (the cache is full or the tables did not fit: the nearest of the two that always exist)
                pick = (size_t)tall.t.n_strips * tall.t.n_bands * batch.n_frames >= 512 ? &tall : &ss->v[1];
            }
        }
    }
    const StreamVariant &v = *pick;
    const size_t tiles = (size_t)v.t.n_strips * v.t.n_bands * batch.n_frames;
    if (tiles > ss->tile_cap[slot]) {
        // (a slot's previous call may still be running on ANOTHER stream: the device is idle before its table goes)
        if (ss->tile_state[slot]) {
            (void)hipDeviceSynchronize();
            (void)DevFree(ss->tile_state[slot]);
        }
        ss->tile_state[slot] = nullptr;
        ss->tile_cap[slot]   = 0;
        hipError_t e         = DevMalloc((void **)&ss->tile_state[slot], tiles * sizeof(int));
        if (e != hipSuccess) return e;
        // (once per allocation, ordered in front of the kernels on this stream)
        if ((e = hipMemsetAsync(ss->tile_state[slot], 0, tiles * sizeof(int), stream)) != hipSuccess) return e;
        ss->tile_cap[slot] = tiles;
        ss->gen[slot]      = 0;
    }
    // a tile is done when it carries this call's generation: nothing to clear per call
    if (++ss->gen[slot] == 0x7fffffff) {
        hipError_t e0 = hipMemsetAsync(ss->tile_state[slot], 0, ss->tile_cap[slot] * sizeof(int), stream);
        if (e0 != hipSuccess) return e0;
        ss->gen[slot] = 1;
    }
    hipError_t e = hipSuccess;
    // cheapest channel set first; tiles whose data breaks its assumption stay
    // open for the next kernel (stream_cfg[3] can skip the optimistic passes)
    const int first_mode = s->stream_cfg[3];
    if (ss->hfirst && ss->h2) {
        // two columns per lane pair for the four-channel sets; the seven-channel set on the halves of their strips
        if (first_mode <= kOpaque && (e = LaunchModeH2<kOpaque>(s, ss, v, blend, batch, stream)) != hipSuccess)
            return e;
        if (first_mode <= kPremult && (e = LaunchModeH2<kPremult>(s, ss, v, blend, batch, stream)) != hipSuccess)
            return e;
        return LaunchModeH<kFull>(s, ss, v, blend, batch, stream);
    }
    if (ss->hfirst) {
        if (first_mode <= kOpaque && (e = LaunchModeH<kOpaque>(s, ss, v, blend, batch, stream)) != hipSuccess)
            return e;
        if (first_mode <= kPremult && (e = LaunchModeH<kPremult>(s, ss, v, blend, batch, stream)) != hipSuccess)
            return e;
        return LaunchModeH<kFull>(s, ss, v, blend, batch, stream);
    }
    // the three- and four-channel sets with their vertical products on the matrix cores where
    // the plan allows (stream_cfg[4]: 1 forces the all-VALU kernels, for comparison)
    const bool matrix = v.m_ok && s->stream_cfg[4] == 0;
    if (first_mode <= kOpaque &&
        (e = matrix ? LaunchModeM<kOpaque>(s, ss, v, blend, batch, stream)
                    : LaunchMode<kOpaque>(s, ss, v, blend, batch, stream)) != hipSuccess)
        return e;
    if (first_mode <= kPremult &&
        (e = matrix ? LaunchModeM<kPremult>(s, ss, v, blend, batch, stream)
                    : LaunchMode<kPremult>(s, ss, v, blend, batch, stream)) != hipSuccess)
        return e;
    return LaunchMode<kFull>(s, ss, v, blend, batch, stream);
}

}  // namespace timg_amd

#ifdef TIMG_M_TRACE
extern "C" int timg_hip_debug_mtrace(unsigned long long *out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(timg_amd::g_mtrace), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(timg_amd::g_mtrace), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
