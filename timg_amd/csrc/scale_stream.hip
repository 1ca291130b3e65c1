// timg_amd/csrc/scale_stream.hip -- streaming scale(+blend) kernel for
// vertical-first shrink plans (what stb picks for 4K -> 800x450 and for
// 4K -> 200x56, see resample_plan.cc VerticalFirst).
//
// Data flow per workgroup = one (column strip, output-row band, frame):
//   * every lane owns kPix adjacent source columns and walks the band's source
//     rows top to bottom ONCE (coalesced 8-byte loads, next row prefetched);
//   * a source row feeds the <= kSlots output rows whose vertical filter
//     covers it; their running sums live in registers (7 channels each) and
//     are updated in source-row order -- exactly stb's vertical chain;
//   * when an output row has seen its last source row its column sums go to
//     LDS and the strip's output pixels are produced by the horizontal
//     even/odd-chain gather, un-weighted, blended and stored as RGBA8.
// Which slot an output row uses, and the per-row weights, come from a small
// host-built schedule, so the kernel has no data-dependent control flow beyond
// wave-uniform branches.
//
// HBM traffic: each source byte is read once per band that needs it (band
// height trades halo re-reads against grid size); same-strip bands land on
// the same XCD (block id mod 8), so the halo rows are L2 hits.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "context.h"
#include "pixel_math.h"

namespace timg_amd {

namespace {

constexpr int kPix       = 2;    // source columns per lane
constexpr int kSlots     = 5;    // output rows in flight per lane
constexpr int kThreads   = 256;
constexpr int kStripCols = kPix * kThreads;
constexpr int kStage     = 2;    // output rows that may complete on one source row
constexpr int kLdsCoeffs = 32;   // horizontal weights kept in LDS up to this tap count

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
    int pad;
};
static_assert(sizeof(RowSched) == 64, "RowSched must stay one cache line");

struct StreamTables {
    const StripInfo *strips;
    const BandInfo *bands;
    const RowSched *sched;
    int n_strips, n_bands;
};

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
template <> struct ModeTraits<kOpaque>  { static constexpr int kCh = 3, kStride = 4; };
template <> struct ModeTraits<kPremult> { static constexpr int kCh = 4, kStride = 4; };
template <> struct ModeTraits<kFull>    { static constexpr int kCh = 7, kStride = 8; };

template <int M>
__device__ __forceinline__ void DecodeMode(uint32_t px, int swap_rb, float out[ModeTraits<M>::kCh]) {
    const float k = 1.0f / 255.0f;
    float r       = (float)(px & 0xffu) * k;
    const float g = (float)((px >> 8) & 0xffu) * k;
    float b       = (float)((px >> 16) & 0xffu) * k;
    if (swap_rb) {
        const float t = r;
        r             = b;
        b             = t;
    }
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

__device__ __forceinline__ uint32_t FinishStreamPixel(const Px7 &acc, int x, int y,
                                                      const DevBlend &blend, int *flag) {
    uint32_t out = EncodePx(acc);
    if ((out >> 24) != 0xffu && y >= blend.start_row) {
        if (flag) *flag = 1;
        if (blend.enabled) {
            const bool alt = blend.checker && (((x / blend.pw) + (y / blend.ph)) & 1);
            out            = BlendOver(out, alt ? blend.pat : blend.bg);
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
    float *stage;    // kStage * kStripCols * 8 floats
    float *hcoef;    // h_width * hrow floats (k-major), valid if lds_coeffs
    int hrow;        // outputs per strip rounded up (row length of hcoef)
    int *fail;       // LDS flag
    bool lds_coeffs;
};

// Runs one tile with channel set M.  Returns false (uniformly) when M's
// assumption did not hold; the caller then retries with the next set.
template <int M>
__device__ bool RunTile(const TileCtx &c) {
    constexpr int kCh = ModeTraits<M>::kCh, kStride = ModeTraits<M>::kStride;
    const DevPlan &plan     = *c.plan;
    const FrameBatch &batch = *c.batch;
    const int tid           = threadIdx.x;
    const int col0          = c.si.cx0 + tid * kPix;
    const bool in0 = col0 < plan.in_w, in1 = col0 + 1 < plan.in_w;
    const uint8_t *src = batch.src + (size_t)c.f * batch.src_frame_stride + (size_t)col0 * 4;
    int *flag          = batch.transparent_flags ? batch.transparent_flags + c.f : nullptr;

    const int ox      = c.si.ox0 + tid;
    const bool has_ox = ox < c.si.ox1;
    int2 ht           = make_int2(c.si.cx0, 1);
    const float *hc   = plan.h_coeff;
    if (has_ox) {
        ht = plan.h_taps[ox];
        hc = plan.h_coeff + (size_t)ox * plan.h_width;
    }

    auto load_row = [&](int r) -> uint2 {
        // lanes past the right edge read nothing and count as opaque
        uint2 v = make_uint2(0xff000000u, 0xff000000u);
        if (r > c.bi.r1) return v;
        const uint8_t *p = src + (size_t)r * batch.src_stride;
        if (in1)
            v = *reinterpret_cast<const uint2 *>(p);
        else if (in0)
            v.x = *reinterpret_cast<const uint32_t *>(p);
        return v;
    };

    float acc[kSlots][kPix][kCh];
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int p = 0; p < kPix; ++p)
#pragma unroll
            for (int ch = 0; ch < kCh; ++ch) acc[s][p][ch] = 0.0f;

    bool ok   = true;  // this lane has seen nothing that breaks M's assumption
    uint2 cur = load_row(c.bi.r0), nx1 = load_row(c.bi.r0 + 1);
    for (int r = c.bi.r0; r <= c.bi.r1; ++r) {
        const uint2 nx2   = load_row(r + 2);
        const RowSched rs = c.sched[r - c.bi.r0];
        if (M == kOpaque) ok = ok && ((cur.x & cur.y) >> 24) == 0xffu;
        float d0[kCh], d1[kCh];
        DecodeMode<M>(cur.x, plan.swap_rb, d0);
        DecodeMode<M>(cur.y, plan.swap_rb, d1);

        int n_done = 0;
        int done_y0 = 0, done_y1 = 0;
        float done_a0 = 0.0f, done_a1 = 0.0f;
        bool completing = false;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) completing = completing || (rs.flags[s] & 4);
        if (completing) {
            // the previous horizontal pass must be done with the staging rows
            if (__any(!ok) && (tid & 63) == 0) *c.fail = 1;
            __syncthreads();
            if (*c.fail) return false;
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int fl = rs.flags[s];
            if (!(fl & 1)) continue;  // wave-uniform
            const float w = rs.weight[s];
            if (fl & 2) {
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch) {
                    acc[s][0][ch] = d0[ch] * w;
                    acc[s][1][ch] = d1[ch] * w;
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < kCh; ++ch) {
                    acc[s][0][ch] = acc[s][0][ch] + d0[ch] * w;
                    acc[s][1][ch] = acc[s][1][ch] + d1[ch] * w;
                }
            }
            if (fl & 4) {
                float *row = c.stage + (size_t)n_done * kStripCols * kStride;
#pragma unroll
                for (int p = 0; p < kPix; ++p) {
                    float *dst = row + (size_t)(tid * kPix + p) * kStride;
                    if (kStride == 4) {
                        float4 v;
                        v.x = acc[s][p][0];
                        v.y = acc[s][p][1];
                        v.z = acc[s][p][2];
                        v.w = kCh > 3 ? acc[s][p][kCh > 3 ? 3 : 0] : 0.0f;
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
                if (n_done == 0) {
                    done_y0 = fl >> 8;
                    done_a0 = rs.alpha_sum[s];
                } else {
                    done_y1 = fl >> 8;
                    done_a1 = rs.alpha_sum[s];
                }
                ++n_done;
            }
        }
        if (n_done) {
            __syncthreads();
            for (int j = 0; j < n_done; ++j) {
                const int y = j == 0 ? done_y0 : done_y1;
                if (has_ox) {
                    const float *base = c.stage + (size_t)j * kStripCols * kStride +
                                        (size_t)(ht.x - c.si.cx0) * kStride;
                    float even[kCh], odd[kCh];
                    float a_even = 0.0f, a_odd = 0.0f;  // kOpaque: alpha chain
                    const float av = j == 0 ? done_a0 : done_a1;
#pragma unroll
                    for (int ch = 0; ch < kCh; ++ch) even[ch] = odd[ch] = 0.0f;
                    auto tap = [&](int k, float v[kCh]) {
                        if (kStride == 4) {
                            const float4 t = *reinterpret_cast<const float4 *>(base + k * 4);
                            v[0] = t.x;
                            v[1] = t.y;
                            v[2] = t.z;
                            if (kCh > 3) v[kCh > 3 ? 3 : 0] = t.w;
                        } else {
                            const float4 t0 = *reinterpret_cast<const float4 *>(base + k * 8);
                            const float4 t1 = *reinterpret_cast<const float4 *>(base + k * 8 + 4);
                            v[0] = t0.x;
                            v[1] = t0.y;
                            v[2] = t0.z;
                            v[kCh > 3 ? 3 : 0] = t0.w;
                            v[kCh > 4 ? 4 : 0] = t1.x;
                            v[kCh > 5 ? 5 : 0] = t1.y;
                            v[kCh > 6 ? 6 : 0] = t1.z;
                        }
                    };
                    auto weight = [&](int k) -> float {
                        return c.lds_coeffs ? c.hcoef[k * c.hrow + tid] : hc[k];
                    };
                    float v[kCh];
                    if (plan.h_sequential) {
                        for (int k = 0; k < ht.y; ++k) {
                            const float hw = weight(k);
                            tap(k, v);
                            if (k == 0) {
#pragma unroll
                                for (int ch = 0; ch < kCh; ++ch) even[ch] = v[ch] * hw;
                                a_even = av * hw;
                            } else {
#pragma unroll
                                for (int ch = 0; ch < kCh; ++ch) even[ch] = even[ch] + v[ch] * hw;
                                a_even = a_even + av * hw;
                            }
                        }
                    } else {
                        {
                            const float hw = weight(0);
                            tap(0, v);
#pragma unroll
                            for (int ch = 0; ch < kCh; ++ch) even[ch] = v[ch] * hw;
                            a_even = av * hw;
                        }
                        if (ht.y > 1) {
                            const float hw = weight(1);
                            tap(1, v);
#pragma unroll
                            for (int ch = 0; ch < kCh; ++ch) odd[ch] = v[ch] * hw;
                            a_odd = av * hw;
                        }
                        for (int k = 2; k < ht.y; ++k) {
                            const float hw = weight(k);
                            tap(k, v);
                            if (k & 1) {
#pragma unroll
                                for (int ch = 0; ch < kCh; ++ch) odd[ch] = odd[ch] + v[ch] * hw;
                                a_odd = a_odd + av * hw;
                            } else {
#pragma unroll
                                for (int ch = 0; ch < kCh; ++ch) even[ch] = even[ch] + v[ch] * hw;
                                a_even = a_even + av * hw;
                            }
                        }
#pragma unroll
                        for (int ch = 0; ch < kCh; ++ch) even[ch] = even[ch] + odd[ch];
                        a_even = a_even + a_odd;
                    }
                    Px7 px;
                    if (M == kOpaque) {
                        px.c[0] = px.c[1] = px.c[2] = 0.0f;
                        px.c[3] = a_even;
                        px.c[4] = even[0];
                        px.c[5] = even[1];
                        px.c[6] = even[2];
                        // with alpha == 1 the straight and the weighted sums coincide
                        px.c[0] = even[0];
                        px.c[1] = even[1];
                        px.c[2] = even[2];
                    } else if (M == kPremult) {
                        px.c[0] = px.c[1] = px.c[2] = 0.0f;
                        px.c[3] = even[0];
                        px.c[4] = even[1];
                        px.c[5] = even[2];
                        px.c[6] = even[kCh > 3 ? 3 : 0];
                        if (px.c[3] < TIMG_TINY_F32) ok = false;  // needs the straight RGB sums
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 7; ++ch) px.c[ch] = even[ch < kCh ? ch : 0];
                    }
                    const uint32_t out = FinishStreamPixel(px, ox, y, *c.blend, flag);
                    *reinterpret_cast<uint32_t *>(batch.dst + (size_t)c.f * batch.dst_frame_stride +
                                                  (size_t)y * batch.dst_stride + (size_t)ox * 4) = out;
                }
            }
        }
        cur = nx1;
        nx1 = nx2;
    }
    if (M == kFull) return true;
    if (__any(!ok) && (tid & 63) == 0) *c.fail = 1;
    __syncthreads();
    return *c.fail == 0;
}

// One kernel per channel set (each gets its own register budget).  They run
// back to back on the stream; tile_state[tile] says which tiles are still open:
// 0 = not produced yet, 1 = done.  A kernel skips tiles that are done and marks
// the ones it completes.
template <int M>
__global__ void __launch_bounds__(kThreads)
ScaleStreamKernel(DevPlan plan, StreamTables tab, DevBlend blend, FrameBatch batch,
                  int *tile_state, int hrow) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int fail;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tile_state[tile] != 0) return;  // uniform: whole workgroup leaves
    TileCtx c;
    c.plan       = &plan;
    c.blend      = &blend;
    c.batch      = &batch;
    c.si         = tab.strips[blockIdx.x];
    c.bi         = tab.bands[blockIdx.y];
    c.sched      = tab.sched + c.bi.sched;
    c.f          = blockIdx.z;
    c.stage      = lds;
    c.hcoef      = lds + kStage * kStripCols * ModeTraits<M>::kStride;
    c.hrow       = hrow;
    c.fail       = &fail;
    c.lds_coeffs = plan.h_width <= kLdsCoeffs;
    if (c.lds_coeffs && (int)threadIdx.x < hrow) {
        const int ox    = c.si.ox0 + threadIdx.x;
        const float *hc = plan.h_coeff + (size_t)ox * plan.h_width;
        for (int k = 0; k < plan.h_width; ++k)
            c.hcoef[k * hrow + threadIdx.x] = ox < c.si.ox1 ? hc[k] : 0.0f;
    }
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    if (RunTile<M>(c) && threadIdx.x == 0) tile_state[tile] = 1;
}

}  // namespace

// ---- host side: applicability + schedule ------------------------------------------
struct StreamVariant {
    void *device   = nullptr;
    StreamTables t = {};
    int band_rows  = 0;
};

struct StreamSchedule {
    // [0]: tall bands (little halo) for batches that fill the chip anyway,
    // [1]: short bands so that a single frame still yields ~500 workgroups
    StreamVariant v[2];
    int hrow        = 0;        // widest strip, in output columns
    int *tile_state = nullptr;  // device, grown on demand
    size_t tile_cap = 0;
};

static bool BuildVariant(const ResamplePlan &p, const std::vector<StripInfo> &strips,
                         const std::vector<int> &first, const std::vector<int> &last,
                         int band_rows, StreamVariant *out) {
    std::vector<BandInfo> bands;
    std::vector<RowSched> sched;
    for (int oy = 0; oy < p.out_h; oy += band_rows) {
        BandInfo b;
        memset(&b, 0, sizeof(b));
        b.oy0   = oy;
        b.oy1   = std::min(p.out_h, oy + band_rows);
        b.r0    = first[b.oy0];
        b.r1    = last[b.oy1 - 1];
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
            for (int j = 0; j < r.count; ++j) {
                RowSched &e = sched[(size_t)b.sched + (size_t)(p.v_rows[r.first + j] - b.r0)];
                const int s = y % kSlots;
                e.weight[s] = p.v_coeff[r.first + j];
                e.flags[s]  = 1 | (j == 0 ? 2 : 0) | (j == r.count - 1 ? 4 : 0) | (y << 8);
                if (j == r.count - 1) e.alpha_sum[s] = alpha;
            }
        }
        bands.push_back(b);
    }
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_strips = 0;
    const size_t o_bands  = align(o_strips + strips.size() * sizeof(StripInfo));
    const size_t o_sched  = align(o_bands + bands.size() * sizeof(BandInfo));
    const size_t total    = align(o_sched + sched.size() * sizeof(RowSched));
    std::vector<char> host(total, 0);
    memcpy(&host[o_strips], strips.data(), strips.size() * sizeof(StripInfo));
    memcpy(&host[o_bands], bands.data(), bands.size() * sizeof(BandInfo));
    memcpy(&host[o_sched], sched.data(), sched.size() * sizeof(RowSched));
    void *dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) return false;
    if (hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev);
        return false;
    }
    out->device     = dev;
    out->t.strips   = (const StripInfo *)((char *)dev + o_strips);
    out->t.bands    = (const BandInfo *)((char *)dev + o_bands);
    out->t.sched    = (const RowSched *)((char *)dev + o_sched);
    out->t.n_strips = (int)strips.size();
    out->t.n_bands  = (int)bands.size();
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
    if (!p.vertical_first) return no("horizontal-first plan");
    if (p.max_active_rows > kSlots) return no("too many output rows per source row");
    if (p.in_w & 1) return no("odd source width");
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
    for (int y = 0; y + kSlots < p.out_h; ++y)
        if (last[y] >= first[y + kSlots]) return no("slot reuse conflict");
    for (int y = 1; y < p.out_h; ++y)
        if (last[y] < last[y - 1] || first[y] < first[y - 1]) return no("non-monotonic rows");
    for (int y = 0; y + kStage < p.out_h; ++y)
        if (last[y] == last[y + kStage]) return no("too many rows complete together");

    // strips: as many output columns as fit with all their taps in kStripCols source columns
    std::vector<StripInfo> strips;
    for (int ox = 0; ox < p.out_w;) {
        StripInfo si;
        si.ox0  = ox;
        si.cx0  = p.h_taps[ox].n0 & ~(kPix - 1);
        si.pad  = 0;
        int end = ox;
        while (end < p.out_w && end - ox < kThreads) {
            const HTaps &t = p.h_taps[end];
            if (t.n0 < si.cx0 || t.n0 + t.count > si.cx0 + kStripCols) break;
            ++end;
        }
        if (end == ox) return no("horizontal window wider than a strip");
        si.ox1 = end;
        strips.push_back(si);
        ox = end;
    }
    StreamSchedule *ss = new StreamSchedule();
    for (const StripInfo &si : strips) ss->hrow = std::max(ss->hrow, si.ox1 - si.ox0);
    const int tall     = std::max(1, std::min(p.out_h, 45));
    int fine           = tall;
    while (fine > 6 && (size_t)strips.size() * ((p.out_h + fine - 1) / fine) < 512)
        fine = std::max(6, fine / 2);
    if (!BuildVariant(p, strips, first, last, tall, &ss->v[0]) ||
        !BuildVariant(p, strips, first, last, fine, &ss->v[1])) {
        for (auto &v : ss->v)
            if (v.device) (void)hipFree(v.device);
        delete ss;
        return no("uploading the schedule failed");
    }
    s->stream_tables = ss;
    s->stream_cfg[0] = ss->v[0].t.n_strips;
    s->stream_cfg[1] = ss->v[0].t.n_bands;
    s->stream_cfg[2] = ss->v[1].t.n_bands;
    return true;
}

void ReleaseStreamSchedule(timg_hip_scaler *s) {
    StreamSchedule *ss = (StreamSchedule *)s->stream_tables;
    if (!ss) return;
    for (auto &v : ss->v)
        if (v.device) (void)hipFree(v.device);
    if (ss->tile_state) (void)hipFree(ss->tile_state);
    delete ss;
    s->stream_tables = nullptr;
}

template <int M>
static hipError_t LaunchMode(const timg_hip_scaler *s, const StreamSchedule *ss,
                             const StreamVariant &v, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream) {
    const bool lds_coeffs = s->plan.h_width <= kLdsCoeffs;
    const size_t lds = ((size_t)kStage * kStripCols * ModeTraits<M>::kStride +
                        (lds_coeffs ? (size_t)s->plan.h_width * ss->hrow : 0)) * sizeof(float);
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)ScaleStreamKernel<M>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const dim3 grid(v.t.n_strips, v.t.n_bands, batch.n_frames);
    hipLaunchKernelGGL(ScaleStreamKernel<M>, grid, dim3(kThreads), lds, stream, s->dev, v.t, blend,
                       batch, ss->tile_state, ss->hrow);
    return hipGetLastError();
}

hipError_t LaunchScaleStream(const timg_hip_scaler *s, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream) {
    StreamSchedule *ss = (StreamSchedule *)s->stream_tables;
    if (!ss) return hipErrorNotSupported;
    // 8-byte loads need 8-byte aligned rows; otherwise the generic kernel runs
    if (((uintptr_t)batch.src & 7) || (batch.src_stride & 7) || (batch.src_frame_stride & 7))
        return LaunchScaleGeneric(s->dev, blend, batch, stream);
    const StreamVariant &tall = ss->v[0];
    const bool enough = (size_t)tall.t.n_strips * tall.t.n_bands * batch.n_frames >= 512;
    const StreamVariant &v = enough ? tall : ss->v[1];
    const size_t tiles = (size_t)v.t.n_strips * v.t.n_bands * batch.n_frames;
    if (tiles > ss->tile_cap) {
        if (ss->tile_state) (void)hipFree(ss->tile_state);
        ss->tile_state = nullptr;
        ss->tile_cap   = 0;
        hipError_t e   = hipMalloc((void **)&ss->tile_state, tiles * sizeof(int));
        if (e != hipSuccess) return e;
        ss->tile_cap = tiles;
    }
    hipError_t e = hipMemsetAsync(ss->tile_state, 0, tiles * sizeof(int), stream);
    if (e != hipSuccess) return e;
    // cheapest channel set first; tiles whose data breaks its assumption stay
    // open for the next kernel (stream_cfg[3] can skip the optimistic passes)
    const int first_mode = s->stream_cfg[3];
    if (first_mode <= kOpaque && (e = LaunchMode<kOpaque>(s, ss, v, blend, batch, stream)) != hipSuccess)
        return e;
    if (first_mode <= kPremult && (e = LaunchMode<kPremult>(s, ss, v, blend, batch, stream)) != hipSuccess)
        return e;
    return LaunchMode<kFull>(s, ss, v, blend, batch, stream);
}

}  // namespace timg_amd
