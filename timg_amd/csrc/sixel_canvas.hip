// timg_amd/csrc/sixel_canvas.hip -- device twin of the pixel work behind
// timg::SixelCanvas::Send (src/sixel-canvas.cc:100-155), i.e. of the two
// libsixel calls it makes: sixel_dither_initialize (adaptive 256-colour
// palette) and sixel_encode (nearest colour + Floyd-Steinberg + band RLE).
//
// libsixel is not part of the reference tree (parity unpinned); the
// kernels below implement the algorithm written down in oracle/sixel.c
// (lookup_mode 1) bit for bit: all arithmetic is integer.
//
//   K1 HistSample / MarkFirst   sparse-sampled 15-bit histogram, first-seen order
//   K2 MedianCut                one wave per frame: median cut -> palette
//   K3 BuildLut                 15-bit cell -> nearest palette entry (+ its rgb)
//   K4 Dither                   one wave per frame, rows skewed by 4 columns:
//                               lookup + Floyd-Steinberg, errors passed between
//                               rows through DPP shuffles (no memory traffic)
//   K5 EncodeBand               one workgroup per 6-row band: colour runs ->
//                               nodes -> libsixel's greedy packing -> RLE bytes
//   K6 AssembleBands             header, palette, band offsets, compaction (one kernel, a workgroup per band)
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <cstring>

#include "context.h"
#include "sixel_launch.h"
#include "wave_ops.h"
#include "pixel_math.h"

namespace timg_amd {
namespace {

constexpr int kMaxColors     = 256;
// (kLdsEntries, kMaxSixelWidth, the diffusion's and the band kernels' launch geometry: sixel_launch.h)

struct SixelGeom {
    int w, h, h6;        // frame, padded height
    int bands;
    size_t stride, frame_stride;
    uint32_t pad[2];     // RGBA of the pad rows: background / pattern colour
    int pad_checker, pad_pw, pad_ph;
    int broken_cursor;
    uint32_t sample_stride_px, n_samples;
    size_t band_cap;     // scratch bytes per band
    int band_ne;         // entry / node slots per band (6 * w rounded up to 64)
    // The palette-index image: idx_stride bytes a row (a multiple of 8, >= w + 6), pixel x of row r at byte
    // r * idx_stride + 2 * (r & 3) + x -- IndexRow().  The diffusion advances a row two columns behind the row above it
    // and stores eight indices at a time; with the rows shifted by two bytes per row (mod 8) every row of a wave
    // completes an aligned group of eight in the SAME step: one 8-byte store per lane every eighth step (K4).
    int idx_stride;        // bytes of an index row
    int idx_shift;         // 0: the index image holds palette indices (a byte a pixel); 1: 15-bit biased CELLS (two bytes a
                           // pixel: the one-trip diffusion leaves the cell -> index lookup to the band kernel, see K4)
    // the diffusion's helper waves (K4): columns a boundary row is moved by at a time, naps of 128 clocks between polls
    int helper_batch, helper_naps;
};

// per-frame device scratch
struct SixelFrameScratch {
    uint32_t *entries;     // [n_samples]  cnt<<15 | hash for first-seen samples, else 0
    uint32_t *tab_a;       // [32768]
    uint32_t *tab_b;       // [32768]
    uint32_t *lut;         // [32768] idx | r<<8 | g<<16 | b<<24
    uint8_t *lut8;         // [32768] idx alone, indexed by the BIASED cell (cell ^ kCellBias): what the diffusion looks up
    uint8_t *palette;      // [768]
    int *meta;             // [0]=ncolors [1]=dither
    uint8_t *index;        // [h6 * w]
    char *band_bytes;      // [bands * band_cap]
    int *band_meta;        // [bands * 4]: len, first colour, last colour, first tag len
    uint32_t *band_off;    // [bands * 2]: output offset, elided bytes
    // per band, band_ne slots each (see the K5 kernels)
    uint32_t *band_ent;    // (colour, x, mask) entries sorted by colour, x
    uint32_t *band_nkey;   // nodes sorted by sx asc, mx desc, colour asc
    uint16_t *band_nfirst; // node id (colour, start order) of each sorted node
    uint32_t *band_pi;     // per node: pass << 16 | index inside the pass
    uint16_t *band_xs;     // per node: pen position when it is put
    uint2 *band_rec;       // per OUTPUT SLOT: {node key, node id | pen << 15 | '$' << 27}
    // per sorted entry: byte offset of what it emits inside its band's node bodies (exclusive
    // prefix over the band), length of the run that ends with it (0: none), its node id;
    // per node id (colour, start order): its first entry
    uint32_t *band_ep;
    uint16_t *band_erl, *band_enode, *band_nf;
    int *band_cnt;         // [bands * 4]: entries, nodes
};

struct SixelBatch {
    const uint8_t *fb;
    uint32_t *entries, *tab_a, *tab_b, *lut;
    uint8_t *lut8;
    uint8_t *palette;
    int *meta;
    uint8_t *index;
    char *band_bytes;
    int *band_meta;
    uint32_t *band_off;
    uint32_t *band_ent, *band_nkey, *band_pi;
    uint16_t *band_nfirst, *band_xs;
    uint2 *band_rec;
    uint32_t *band_ep;
    uint16_t *band_erl, *band_enode, *band_nf;
    int *band_cnt;
    uint32_t *pad_rows;          // [frames][5][w] the rows appended below the frame, as pixels (K4)
    uint32_t *xwg;               // [frames][kDitherMaxParts - 1][XwgStride(w)] boundary rows handed from CU to CU (K4)
    int *error;                  // [1] set when a device-side wait gives up
    char *out;
    size_t out_cap;
    unsigned long long *out_len;
    // One group of frames (the default): the step's first kernel clears the error word and its last one writes the
    // frames' byte counts and the error word straight into the caller's PINNED words -- no memset in front of the chain,
    // no copy behind it (two launches of ~4 us and their gaps, 1.3 % of a 64-frame step).  Null / 0: hipMemsetAsync and
    // hipMemcpyAsync as before (several groups on side streams: a group's kernels must not clear or report for another's).
    int clears_error;
    unsigned long long *len_host;  // [frames] mirror of out_len (this group's first frame at 0)
    unsigned long long *err_host;  // [1]
};

__device__ __forceinline__ uint32_t IndexRow(const SixelGeom &g, int row) {  // byte offset of pixel 0 of `row`
    return (uint32_t)row * (uint32_t)g.idx_stride + ((2u * ((uint32_t)row & 3u)) << g.idx_shift);
}

__device__ __forceinline__ SixelFrameScratch FrameScratch(const SixelBatch &b, const SixelGeom &g,
                                                          int f) {
    SixelFrameScratch s;
    s.entries    = b.entries + (size_t)f * g.n_samples;
    s.tab_a      = b.tab_a + (size_t)f * 32768;
    s.tab_b      = b.tab_b + (size_t)f * 32768;
    s.lut        = b.lut + (size_t)f * 32768;
    s.lut8       = b.lut8 + (size_t)f * 32768;
    s.palette    = b.palette + (size_t)f * 768;
    s.meta       = b.meta + (size_t)f * 4;
    s.index      = b.index + (size_t)f * g.h6 * g.idx_stride;
    s.band_bytes = b.band_bytes + (size_t)f * g.bands * g.band_cap;
    s.band_meta  = b.band_meta + (size_t)f * g.bands * 4;
    s.band_off   = b.band_off + (size_t)f * g.bands * 2;
    const size_t fb = (size_t)f * g.bands;
    s.band_ent    = b.band_ent + fb * g.band_ne;
    s.band_nkey   = b.band_nkey + fb * g.band_ne;
    s.band_nfirst = b.band_nfirst + fb * g.band_ne;
    s.band_pi     = b.band_pi + fb * g.band_ne;
    s.band_xs     = b.band_xs + fb * g.band_ne;
    s.band_rec    = b.band_rec + fb * g.band_ne;
    s.band_ep     = b.band_ep + fb * g.band_ne;
    s.band_erl    = b.band_erl + fb * g.band_ne;
    s.band_enode  = b.band_enode + fb * g.band_ne;
    s.band_nf     = b.band_nf + fb * g.band_ne;
    s.band_cnt    = b.band_cnt + fb * 4;
    return s;
}

// Pixel of the padded frame SixelCanvas::Send builds (src/sixel-canvas.cc:111-120):
// the original rows, then rows of background (or checkerboard) colour.
__device__ __forceinline__ uint32_t PaddedPixel(const uint8_t *frame, const SixelGeom &g, int x,
                                                int y) {
    if (y < g.h) return *reinterpret_cast<const uint32_t *>(frame + (size_t)y * g.stride + (size_t)x * 4);
    const int alt = g.pad_checker && (((x / g.pad_pw) + (y / g.pad_ph)) & 1);
    return g.pad[alt];
}

__device__ __forceinline__ uint32_t Hash555(uint32_t px) {  // r,g,b in the low 3 bytes
    return ((px & 0xf8u) << 7) | (((px >> 8) & 0xf8u) << 2) | ((px >> 19) & 0x1fu);
}

// decimal digits of v, and the bytes of palette entry n in the output ("#n;2;r;g;b" in percent)
__device__ __forceinline__ int NumLen(uint32_t v) {
    return v >= 10000 ? 5 : v >= 1000 ? 4 : v >= 100 ? 3 : v >= 10 ? 2 : 1;
}
__device__ __forceinline__ int PaletteEntryLen(int n, uint32_t r, uint32_t g, uint32_t b) {
    return 1 + NumLen((uint32_t)n) + 3 + NumLen((r * 100u + 127u) / 255u) + 1 + NumLen((g * 100u + 127u) / 255u) + 1 +
           NumLen((b * 100u + 127u) / 255u);
}
__device__ __forceinline__ int PaletteEntryLen(int n, const uint8_t *rgb) { return PaletteEntryLen(n, rgb[0], rgb[1], rgb[2]); }

// ---- K1: sampled 15-bit histogram, one workgroup per frame --------------------------------
// libsixel's computeHistogram walks the samples in order, counts every 5:5:5 colour and
// lists the colours in first-seen order.  Both answers (first sample index, count) live in
// ONE word per bin of a 128 KB LDS table; the result is entries[k] = count<<15 | hash
// for the sample that saw its colour first, 0 for every other sample -- the global 32768-bin
// tables of the first version (and their 16 MB memset per batch) are gone.
constexpr int kHistThreads = 1024;
constexpr size_t kHistLdsBytes = 32768 * sizeof(uint32_t);

constexpr int kHistPerThread = 36;  // n_samples <= 2 * 18383 (libsixel's sampling budget)

__global__ void __launch_bounds__(kHistThreads) HistKernel(SixelGeom g, SixelBatch b) {
    extern __shared__ uint32_t hist_lds[];
    const int f   = blockIdx.x;
    const int tid = threadIdx.x;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const uint8_t *frame      = b.fb + (size_t)f * g.frame_stride;

    if (b.clears_error && f == 0 && tid == 0) *b.error = 0;  // (only the diffusion, three kernels on, ever sets it)
    for (int i = tid; i < 32768; i += kHistThreads) hist_lds[i] = 0xffffffffu;
    // the thread's samples k = tid, tid + 1024, ...: all loads in flight together, hashes
    // kept in registers for the passes below; (x, y) advance without divisions
    uint32_t hash[kHistPerThread];
    {
        const uint32_t step = kHistThreads * g.sample_stride_px;
        const int dx = (int)(step % g.w), dy = (int)(step / g.w);
        const uint32_t p0 = (uint32_t)tid * g.sample_stride_px;
        int x = (int)(p0 % g.w), y = (int)(p0 / g.w);
#pragma unroll
        for (int j = 0; j < kHistPerThread; ++j) {
            const bool in = (uint32_t)(tid + j * kHistThreads) < g.n_samples;
            hash[j]       = in ? Hash555(PaddedPixel(frame, g, x, y)) : 0xffffffffu;
            x += dx;
            y += dy;
            if (x >= g.w) {
                x -= g.w;
                ++y;
            }
        }
    }
    __syncthreads();
    // One table serves both questions, one after the other WITHOUT a second clear: a bin first takes the smallest
    // sample index that hits it (atomicMin; an index is below 2^16: n_samples <= 36 864), then every hit adds 1 << 16 --
    // the low half keeps the first index, the high half counts (<= n_samples, no carry out of the word).  (Until
    // round 4: min, read back, clear to zero, count, read -- one more pass of 32 stores and 36 reads a thread and
    // a barrier.)
#pragma unroll
    for (int j = 0; j < kHistPerThread; ++j)
        if (hash[j] != 0xffffffffu) atomicMin(&hist_lds[hash[j]], (uint32_t)(tid + j * kHistThreads));
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kHistPerThread; ++j)
        if (hash[j] != 0xffffffffu) atomicAdd(&hist_lds[hash[j]], 1u << 16);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kHistPerThread; ++j) {
        if (hash[j] == 0xffffffffu) continue;
        const uint32_t w = hist_lds[hash[j]];
        uint32_t e       = 0;
        if ((w & 0xffffu) == (uint32_t)(tid + j * kHistThreads)) {
            const uint32_t c = w >> 16;  // (libsixel's histogram is unsigned short, saturating: never reached here)
            e = (c << 15) | hash[j];
        }
        s.entries[tid + j * kHistThreads] = e;
    }
}

// ---- K2: median cut, one workgroup per frame -------------------------------------------
// libsixel's median cut always splits the largest remaining box, which reads as 255
// dependent steps.  But what a split DOES (plane choice, stable sort of the box's colours,
// median) depends on that box alone -- only WHICH boxes get split depends on the order.
// So every round the kCutWaves waves split, concurrently and speculatively, the first
// kCutWaves splittable boxes of the list (one wave each, into the other half of the
// ping-pong colour table, leaving the source intact), and one wave then replays
// libsixel's list bookkeeping with the prepared results for as long as the box it
// needs next has been prepared.  Boxes near the head of the sum-ordered list are exactly
// the ones the serial algorithm takes next, so almost no speculation is wasted.
// Inside a split: boxes of <= 64 colours are sorted in registers (ds_permute), up to 256
// colours four entries per lane, larger ones by a stable per-lane-segment counting sort on
// the 5-bit key.  The list itself is never materialised: see "The box list" in the kernel.
// (12 waves = 12 speculative splits per round: 320 us per 64 frames against 346 with 8; the per-wave scratch of a
// 13th would not fit beside the 96 KB colour table and the box list any more)
#ifndef TIMG_CUT_WAVES
#define TIMG_CUT_WAVES 12
#endif
constexpr int kCutWaves      = TIMG_CUT_WAVES;
#ifndef TIMG_CUT_PICK_LANES
#define TIMG_CUT_PICK_LANES 40
#endif
constexpr int kCutPickLanes  = TIMG_CUT_PICK_LANES;  // lane maxima a round's picks start from (see pick() in the kernel)
constexpr int kCutLdsEntries = 12288;             // colour table kept in LDS up to this size
// words of per-wave scratch: [64][33] 16-bit counters (a lane's segment of a box and every
// exclusive prefix stay below 65536: a box has at most 32768 colours) plus 32 + 32 key totals /
// bases; the medium path uses the first 256 words as its permutation buffer instead
constexpr int kCutScratch    = 64 * 17;  // per wave: 64 rows of 16 words (32 packed 16-bit counters) + 1 spare word
constexpr size_t kCutLdsBytes =
    ((size_t)2 * kCutLdsEntries + (size_t)kCutWaves * kCutScratch) * sizeof(uint32_t);

__device__ __forceinline__ uint32_t PlaneKey(uint32_t entry, int plane) {
    return (entry >> (10 - 5 * plane)) & 0x1fu;  // plane 0=r 1=g 2=b
}

struct CutBox {  // what a split needs to know of its box (registers)
    uint32_t ind, colors, sum, buf;  // buf: which half of the ping-pong table holds it
};
// A box of the list in LDS, 16 bytes: the bookkeeping reads (w0, w2) of every box it is about to split in one
// ds_read_b64 and writes a box with one ds_write_b128.  ind < 32768, colors <= 32768, median < colors and every pixel
// sum <= the number of samples (<= 36766) -- sixteen bits each.
struct alignas(16) CutRec {
    uint32_t w0;  // ind | buf << 15 | colors << 16
    uint32_t w2;  // the prepared split: median | lowersum << 16
    uint32_t w1;  // sum | tie << 16   (tie = 256 -+ the step that made the box: see "The box list")
    uint32_t spare;
};
__device__ __forceinline__ CutBox BoxOf(const CutRec &r) {
    return CutBox{r.w0 & 0x7fffu, r.w0 >> 16, r.w1 & 0xffffu, (r.w0 >> 15) & 1u};
}

// Memory traffic of ONE wave needs no barrier (its operations are issued in order);
// this drains them and keeps the compiler from moving or caching accesses across the point.
#define TIMG_WAVE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")

// Prepares the split of `box` by one wave: sorts its colours (stable, by the plane
// with the largest luminosity-weighted spread) into the other table half and finds
// the median.  scratch: kCutScratch words owned by this wave.
// kLds: the colour table lives in LDS (up to kCutLdsEntries colours: every frame of a photograph).  Which of the two
// it is is decided once per frame, but a pointer that may be either is a FLAT pointer: every load of a colour went
// down the vector-memory path to find out it was LDS (flat_load_dword, s_waitcnt vmcnt(0) lgkmcnt(0)) -- several times
// the latency of a ds_read in loops that are chains of such loads.  Typed per case, the table is read with ds_read.
typedef __attribute__((address_space(3))) uint32_t CutLdsWord;
template <bool kLds>
__device__ void SplitBox(const CutBox &box, uint32_t *const tab[2], uint32_t *scratch, int lane,
                         uint32_t *median_out, uint32_t *lowersum_out) {
    typedef typename std::conditional<kLds, CutLdsWord, uint32_t>::type Word;
    // large path: lane l counts its segment's keys in row l -- 32 sixteen-bit counters packed into 16 words, rows 17
    // words apart (an odd stride: the lanes' rows start on different banks); the spare 17th word of row k holds the
    // base of key k in the sorted box
    uint16_t *lane_cnt    = reinterpret_cast<uint16_t *>(scratch);  // [64][34]
    constexpr int kRow16  = 34;
    auto key_base  = [&](uint32_t k) -> uint32_t & { return scratch[k * 17 + 16]; };
    uint32_t *perm        = scratch;  // [256]: the medium path's permutation buffer (never together with lane_cnt)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const Word *src     = (const Word *)tab[box.buf] + box.ind;
    Word *dst           = (Word *)tab[box.buf ^ 1u] + box.ind;
    const uint32_t half = box.sum / 2;
    uint32_t median, lowersum;

    // per-plane extent of the box (5-bit keys): which key values occur, OR-ed over the wave
    uint32_t seen[3] = {0, 0, 0};
    uint32_t e_own   = 0;  // (small box: the lane's colour, the last one again past the end)
    if (box.colors <= 64) {
        e_own = src[min((uint32_t)lane, box.colors - 1)];
#pragma unroll
        for (int p = 0; p < 3; ++p) seen[p] = 1u << PlaneKey(e_own, p);
    } else {
        // (eight loads in flight: one load per iteration makes a box of thousands of colours a chain of LDS round trips)
        for (uint32_t i = lane; i < box.colors; i += 8 * 64) {
            uint32_t e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = src[min(i + q * 64, box.colors - 1)];  // (past the end: the last colour again)
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) seen[p] |= 1u << PlaneKey(e[q], p);
        }
    }
    uint32_t mn[3], mx[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const uint32_t m = WaveOr(seen[p]);  // (a box has at least one colour)
        mn[p] = (uint32_t)__ffs((int)m) - 1u;
        mx[p] = 31u - (uint32_t)__clz((int)m);
    }
    // SIXEL_LARGE_LUM: plane with the largest luminosity-weighted spread, first of equals.  libsixel compares
    // lum[p] * ((max - min) << 3) in double with lum = 0.2989, 0.5866, 0.1145; the ranges are multiples of 8 below 256
    // and no two of 2989 i, 5866 j, 1145 k (0 < i, j, k < 32) are equal (gcd(2989, 5866) = 7 and 427 does not divide
    // j < 32; 1145 is coprime to both), so products that differ differ by >= 8e-4 -- thirteen orders of magnitude
    // above the rounding of a double: the integer comparison decides exactly as the double one does.
    int plane = 0;
    uint32_t key_lo = mn[0], key_span = mx[0] - mn[0];  // the chosen plane's smallest key, and largest minus smallest
    {
        const uint32_t lum[3] = {2989u, 5866u, 1145u};
        uint32_t best         = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint32_t spread = lum[p] * (mx[p] - mn[p]);
            if (spread > best) {
                plane    = p;
                best     = spread;
                key_lo   = mn[p];
                key_span = mx[p] - mn[p];
            }
        }
    }
    // The radix sorts below sort by key - key_lo, which has only as many bits as key_span: the boxes of the late rounds
    // span a few key values in their widest plane, two or three passes instead of five.
    const int key_bits         = 32 - __clz((int)key_span);  // (0 when all keys are equal: nothing to sort)
    const uint32_t key_of_dead = (1u << key_bits) - 1u;

    if (box.colors <= 64) {
        // ---- small box: stable LSD radix sort on the key, in registers -----
        const bool live = (uint32_t)lane < box.colors;
        uint32_t e      = live ? e_own : 0xffffffffu;
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) {
            if (bit >= key_bits) break;
            // dead lanes stand behind the live ones and carry the largest key: stable passes keep them there
            const uint32_t key = e == 0xffffffffu ? key_of_dead : PlaneKey(e, plane) - key_lo;
            const bool one     = (key >> bit) & 1u;
            const unsigned long long ones = __ballot(one);
            const int n_zero   = 64 - __popcll(ones);
            const int dest     = one ? n_zero + __popcll(ones & lt_mask) : __popcll(~ones & lt_mask);
            // forward permutation through the LDS crossbar (ds_permute_b32: lane -> lane dest)
            e = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)e);
        }
        if (live) dst[lane] = e;
        const uint32_t c = live ? (e >> 15) : 0u;
        const uint32_t incl = WaveInclusiveAdd(c);
        const uint32_t pre  = incl - c;  // P(lane): pixels in front of entry `lane`
        const unsigned long long hit = __ballot(live && lane >= 1 && pre >= half);
        median = hit ? (uint32_t)__ffsll((long long)hit) - 1 : box.colors - 1;
        if (median > box.colors - 1) median = box.colors - 1;
        lowersum = ReadLane(pre, (int)median);
    } else if (box.colors <= 256) {
        // ---- medium box: the same radix sort with four consecutive entries per lane (entry
        // lane * 4 + r), the permutation through LDS.  A counting sort pays ~2 us of fixed cost
        // (its per-key prefix over the lanes is a chain of dependent LDS updates) whatever the size.
        uint32_t e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = (uint32_t)lane * 4 + r;
            e[r]             = i < box.colors ? src[i] : 0xffffffffu;
        }
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) {
            if (bit >= key_bits) break;
            bool one[4];
            uint32_t nz = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t key = e[r] == 0xffffffffu ? key_of_dead : PlaneKey(e[r], plane) - key_lo;
                one[r]             = (key >> bit) & 1u;
                nz += one[r] ? 0u : 1u;
            }
            const uint32_t incl  = WaveInclusiveAdd(nz);
            const uint32_t zeros = ReadLane(incl, 63);
            uint32_t zb          = incl - nz;  // zeros in front of this lane's first entry
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t i = (uint32_t)lane * 4 + r;
                perm[one[r] ? zeros + (i - zb) : zb] = e[r];
                zb += one[r] ? 0u : 1u;
            }
            TIMG_WAVE_SYNC();
            const uint4 v = *reinterpret_cast<const uint4 *>(perm + lane * 4);
            e[0] = v.x, e[1] = v.y, e[2] = v.z, e[3] = v.w;
            TIMG_WAVE_SYNC();
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = (uint32_t)lane * 4 + r;
            if (i < box.colors) dst[i] = e[r];
            cnt += i < box.colors ? e[r] >> 15 : 0u;
        }
        uint32_t run  = WaveInclusiveAdd(cnt) - cnt;  // pixels in front of entry lane * 4
        uint32_t cand = 0xffffffffu, cand_sum = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = (uint32_t)lane * 4 + r;
            if (cand == 0xffffffffu && i >= 1 && i < box.colors && run >= half) {
                cand     = i;
                cand_sum = run;
            }
            run += i < box.colors ? e[r] >> 15 : 0u;
        }
        const unsigned long long hit = __ballot(cand != 0xffffffffu);
        if (hit) {
            const int l = __ffsll((long long)hit) - 1;  // lowest lane = lowest index
            median      = ReadLane(cand, l);
            lowersum    = ReadLane(cand_sum, l);
        } else {
            median   = box.colors - 1;
            lowersum = 0;
        }
        if (median >= box.colors - 1) {  // (as in the large path: everything but the last colour below)
            median = box.colors - 1;
            __threadfence_block();
            TIMG_WAVE_SYNC();
            lowersum = box.sum - (dst[box.colors - 1] >> 15);
        }
    } else {
        // ---- large box: stable counting sort, one contiguous segment per lane -
        // (odd segment length keeps the lanes on different LDS banks)
        // A chain of "load an entry, read-modify-write its counter" per entry made the root box (6 000 colours, 92
        // entries a lane) cost 33 us and the first six rounds 78 of the kernel's 290: every step waited for two LDS
        // round trips.  Now the entries are fetched four at a time, counting is a fire-and-forget ds_add on the packed
        // counters (no result, nothing to wait for), the prefix over the lanes reads its 32 counters into registers and
        // writes them back once, and the scatter's four fetch-and-adds are in flight together.
        const uint32_t seg = (((box.colors + 63) / 64) | 1u);
        const uint32_t a   = min(box.colors, (uint32_t)lane * seg);
        const uint32_t z   = min(box.colors, a + seg);
        uint32_t *mine_w   = scratch + lane * 17;  // counters of keys 2w and 2w + 1 in the halves of word w
        for (int k = 0; k < 16; ++k) mine_w[k] = 0;
        for (uint32_t i = a; i < z; i += 4) {
            uint32_t e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = src[min(i + q, z - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // (an entry past the segment adds 0: no branches)
                const uint32_t k = PlaneKey(e[q], plane);
                (void)__hip_atomic_fetch_add(&mine_w[k >> 1], i + q < z ? 1u << ((k & 1u) * 16) : 0u, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        TIMG_WAVE_SYNC();
        // exclusive prefix over the lanes, per key: lanes 0-31 take key = lane for the
        // lower half of the lanes, lanes 32-63 the same key for the upper half
        {
            const int key = lane & 31, l0 = (lane >> 5) * 32;
            uint32_t t[32], sum = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) t[q] = lane_cnt[(l0 + q) * kRow16 + key];
#pragma unroll
            for (int q = 0; q < 32; ++q) sum += t[q];
            const uint32_t other = __shfl_xor(sum, 32);  // the other half's total for this key
            uint32_t run         = lane < 32 ? 0u : other;  // (the lower lanes' entries stand in front)
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                lane_cnt[(l0 + q) * kRow16 + key] = (uint16_t)run;
                run += t[q];
            }
            const uint32_t total = sum + other;
            const uint32_t incl = WaveInclusiveAdd(lane < 32 ? total : 0u);
            if (lane < 32) key_base(lane) = incl - total;
        }
        TIMG_WAVE_SYNC();
        for (uint32_t i = a; i < z; i += 4) {
            uint32_t e[4], k[4], was[4], kb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = src[min(i + q, z - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                k[q]   = PlaneKey(e[q], plane);
                kb[q]  = key_base(k[q]);
                // (issued in order: two entries of one key take consecutive places; an entry past the segment adds 0)
                was[q] = __hip_atomic_fetch_add(&mine_w[k[q] >> 1], i + q < z ? 1u << ((k[q] & 1u) * 16) : 0u,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i + q < z) dst[kb[q] + ((was[q] >> ((k[q] & 1u) * 16)) & 0xffffu)] = e[q];
        }
        __threadfence_block();
        TIMG_WAVE_SYNC();
        // median: the lane whose segment of the sorted box holds the crossing finds it (no early exit: the loads
        // of a loop that may leave cannot be issued ahead)
        uint32_t seg_sum = 0;
        for (uint32_t i = a; i < z; i += 8) {
            uint32_t c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) c[q] = dst[min(i + q, z - 1)] >> 15;
#pragma unroll
            for (int q = 0; q < 8; ++q) seg_sum += i + q < z ? c[q] : 0u;
        }
        uint32_t run  = WaveInclusiveAdd(seg_sum) - seg_sum;  // P(a)
        uint32_t cand = 0xffffffffu, cand_sum = 0;
        // (a lane whose segment ends below half has no candidate; the lanes behind the crossing find theirs at their
        // first entry, the lane of the crossing inside its segment -- the walk ends, wave-uniformly, when they all have)
        const bool mine = run + seg_sum >= half;
        for (uint32_t i = a; i < z; i += 8) {
            if (!__any(mine && cand == 0xffffffffu)) break;
            uint32_t c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) c[q] = dst[min(i + q, z - 1)] >> 15;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (i + q < z && cand == 0xffffffffu && i + q >= 1 && run >= half) {
                    cand     = i + q;
                    cand_sum = run;
                }
                run += i + q < z ? c[q] : 0u;
            }
        }
        const unsigned long long hit = __ballot(cand != 0xffffffffu);
        if (hit) {
            const int l = __ffsll((long long)hit) - 1;  // lowest lane = lowest index
            median      = ReadLane(cand, l);
            lowersum    = ReadLane(cand_sum, l);
        } else {
            median   = box.colors - 1;
            lowersum = 0;
        }
        if (median >= box.colors - 1) {
            median   = box.colors - 1;
            lowersum = box.sum - (dst[box.colors - 1] >> 15);
        }
    }
    *median_out   = median;
    *lowersum_out = lowersum;
}

// The first rounds of a frame pick one, two, four boxes of thousands of colours: one wave each sorted them while the
// others idled (19 + 11 + 7 of the kernel's ~120 us on the traced frame; a wave issues an instruction per four clocks,
// ~60 of them per colour and lane).  When a round picks one or two boxes and one of them is large, TEAMS of waves
// split them instead -- twelve or six waves per box, 64 x T virtual lanes with a segment each -- by the same
// stable counting sort as SplitBox's large path, its phases separated by workgroup barriers that EVERY wave of the
// workgroup executes (a team without a box walks empty ranges): extent; count; group sums; bases and write-back;
// scatter; segment sums; median.  The counter rows of a team are the scratch areas of its waves, contiguous.
constexpr uint32_t kCutTeamMin = 1024;  // colours of the largest pick from which teams pay (seven barriers ~ 3 us)
struct CutTeamShared {
    uint32_t seen[kCutWaves][3];      // per wave: key values seen per plane
    uint16_t grp[2 * kCutWaves * 32]; // per team, group of 32 segments, key: entries (<= 32768)
    uint32_t wsum[kCutWaves];         // per wave: pixels of its lanes' segments of the sorted box
    uint2 wcand[kCutWaves];           // per wave: the median candidate of its lowest lane that has one, or ~0
};
template <bool kLds>
__device__ void TeamSplit(bool has_box, const CutBox &box, int T, int team, int tw, int wave, int lane,
                          uint32_t *const tab[2], uint32_t *team_scratch, CutTeamShared &sh, uint32_t *median_out,
                          uint32_t *lowersum_out) {
    typedef typename std::conditional<kLds, CutLdsWord, uint32_t>::type Word;
    const uint32_t VL = 64u * (uint32_t)T, vl = (uint32_t)tw * 64u + (uint32_t)lane;
    const uint32_t colors = has_box ? box.colors : 0u;
    const Word *src       = (const Word *)tab[box.buf] + box.ind;
    Word *dst             = (Word *)tab[box.buf ^ 1u] + box.ind;
    const uint32_t half   = box.sum / 2;
    uint16_t *cnt16       = reinterpret_cast<uint16_t *>(team_scratch);  // [64 T][34]: row v at word 17 v
    constexpr int kRow16  = 34;
    auto key_base = [&](uint32_t k) -> uint32_t & { return team_scratch[k * 17 + 16]; };

    // ---- extent of the box per plane
    uint32_t seen[3] = {0, 0, 0};
    for (uint32_t i = vl; i < colors; i += 4 * VL) {
        uint32_t e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = src[min(i + q * VL, colors - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 3; ++p) seen[p] |= 1u << PlaneKey(e[q], p);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const uint32_t m = WaveOr(seen[p]);
        if (lane == 0) sh.seen[wave][p] = m;
    }
    __threadfence_block();
    __syncthreads();
    int plane = 0;
    {
        uint32_t m3[3] = {0, 0, 0};
        for (int w = 0; w < T; ++w)
#pragma unroll
            for (int p = 0; p < 3; ++p) m3[p] |= sh.seen[team * T + w][p];
        const uint32_t lum[3] = {2989u, 5866u, 1145u};  // (see SplitBox: the integer comparison decides as the double one)
        uint32_t best         = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint32_t m      = m3[p] | (has_box ? 0u : 1u);
            const uint32_t spread = lum[p] * ((31u - (uint32_t)__clz((int)m)) - ((uint32_t)__ffs((int)m) - 1u));
            if (spread > best) {
                plane = p;
                best  = spread;
            }
        }
    }
    // ---- count: virtual lane v takes the segment [a, z) and counts its keys in row v
    const uint32_t seg = (((colors + VL - 1) / VL) | 1u);
    const uint32_t a   = min(colors, vl * seg);
    const uint32_t z   = min(colors, a + seg);
    uint32_t *mine_w   = team_scratch + vl * 17;
    for (int k = 0; k < 16; ++k) mine_w[k] = 0;
    for (uint32_t i = a; i < z; i += 4) {
        uint32_t e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = src[min(i + q, z - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t k = PlaneKey(e[q], plane);
            (void)__hip_atomic_fetch_add(&mine_w[k >> 1], i + q < z ? 1u << ((k & 1u) * 16) : 0u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- exclusive prefix over the virtual lanes, per key: thread (key, group of 32 rows)
    const uint32_t key = vl & 31u, grp = vl >> 5, n_grp = 2u * (uint32_t)T;
    uint16_t *my_grp   = sh.grp + (uint32_t)team * n_grp * 32u;
    uint32_t t[32], sum = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) t[q] = cnt16[(grp * 32 + q) * kRow16 + key];
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += t[q];
    my_grp[grp * 32 + key] = (uint16_t)sum;
    __threadfence_block();
    __syncthreads();
    {
        uint32_t base = 0, total = 0;
        for (uint32_t g2 = 0; g2 < n_grp; ++g2) {
            const uint32_t v = my_grp[g2 * 32 + key];
            total += v;
            base += g2 < grp ? v : 0u;
        }
        const uint32_t incl = WaveInclusiveAdd(lane < 32 ? total : 0u);
        if (tw == 0 && lane < 32) key_base((uint32_t)lane) = incl - total;
        uint32_t run = base;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            cnt16[(grp * 32 + q) * kRow16 + key] = (uint16_t)run;
            run += t[q];
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- scatter
    for (uint32_t i = a; i < z; i += 4) {
        uint32_t e[4], k[4], was[4], kb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = src[min(i + q, z - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            k[q]   = PlaneKey(e[q], plane);
            kb[q]  = key_base(k[q]);
            was[q] = __hip_atomic_fetch_add(&mine_w[k[q] >> 1], i + q < z ? 1u << ((k[q] & 1u) * 16) : 0u,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i + q < z) dst[kb[q] + ((was[q] >> ((k[q] & 1u) * 16)) & 0xffffu)] = e[q];
    }
    __threadfence_block();
    __syncthreads();
    // ---- median: pixels in front of every segment of the sorted box, then the crossing
    uint32_t seg_sum = 0;
    for (uint32_t i = a; i < z; i += 8) {
        uint32_t c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = dst[min(i + q, z - 1)] >> 15;
#pragma unroll
        for (int q = 0; q < 8; ++q) seg_sum += i + q < z ? c[q] : 0u;
    }
    const uint32_t incl2 = WaveInclusiveAdd(seg_sum);
    if (lane == 63) sh.wsum[wave] = incl2;
    __threadfence_block();
    __syncthreads();
    uint32_t run = incl2 - seg_sum;
    for (int w = 0; w < tw; ++w) run += sh.wsum[team * T + w];
    uint32_t cand = 0xffffffffu, cand_sum = 0;
    const bool mine = run + seg_sum >= half;
    for (uint32_t i = a; i < z; i += 8) {
        if (!__any(mine && cand == 0xffffffffu)) break;
        uint32_t c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = dst[min(i + q, z - 1)] >> 15;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i + q < z && cand == 0xffffffffu && i + q >= 1 && run >= half) {
                cand     = i + q;
                cand_sum = run;
            }
            run += i + q < z ? c[q] : 0u;
        }
    }
    {
        const unsigned long long hit = __ballot(cand != 0xffffffffu);
        const int l                  = hit ? __builtin_ctzll(hit) : 0;
        const uint32_t wc = ReadLane(cand, l), ws = ReadLane(cand_sum, l);
        if (lane == 0) sh.wcand[wave] = hit ? make_uint2(wc, ws) : make_uint2(0xffffffffu, 0u);
    }
    __threadfence_block();
    __syncthreads();
    uint32_t median = colors ? colors - 1 : 0u, lowersum = 0;
    if (has_box) {
        for (int w = T - 1; w >= 0; --w) {  // the lowest wave that has a candidate
            const uint2 c = sh.wcand[team * T + w];
            if (c.x != 0xffffffffu) {
                median   = c.x;
                lowersum = c.y;
            }
        }
        if (median >= colors - 1) {
            median   = colors - 1;
            lowersum = box.sum - (dst[colors - 1] >> 15);
        }
    }
    *median_out   = median;
    *lowersum_out = lowersum;
}

// channel sums (as 8-bit values) of a box's colours, for its palette entry
template <class Word>
__device__ __forceinline__ void BoxColourSums(const Word *src, uint32_t colors, uint32_t sum[3]) {
    sum[0] = sum[1] = sum[2] = 0;
    for (uint32_t i = 0; i < colors; ++i) {
        const uint32_t e = src[i];
        sum[0] += ((e >> 10) & 0x1f) << 3;
        sum[1] += ((e >> 5) & 0x1f) << 3;
        sum[2] += (e & 0x1f) << 3;
    }
}

__global__ void __launch_bounds__(kCutWaves * 64) MedianCutKernel(SixelGeom g, SixelBatch b) {
    extern __shared__ uint32_t cut_lds[];
#ifdef TIMG_CUT_TRACE
    const long long t_kernel = wall_clock64();
#endif
    __shared__ CutRec pool[kMaxColors];
    __shared__ uint32_t s_n, s_total, s_nboxes, s_done;
    const int f    = blockIdx.x;
    const int tid  = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: wave 0's branches are)
    const SixelFrameScratch s = FrameScratch(b, g, f);
    uint32_t *scratch = cut_lds + 2 * kCutLdsEntries + wave * kCutScratch;

    // compact the first-seen entries (stable: sample order) into tab_a: every wave takes a
    // contiguous range of samples in coalesced chunks of 64, counts with ballots, and the
    // ranges are stitched together by a scan over the waves
    {
        __shared__ uint32_t s_wave_cnt[kCutWaves], s_wave_sum[kCutWaves];
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        const uint32_t chunks = (g.n_samples + 63) / 64;
        const uint32_t per    = (chunks + kCutWaves - 1) / kCutWaves;
        const uint32_t c0 = min(chunks, (uint32_t)wave * per), c1 = min(chunks, c0 + per);
        // (every chunk of the wave is loaded ONCE, all loads in flight together, and kept in registers for the second
        // pass: two passes of eight loads at a time were twelve global round trips, 5 of the kernel's 130 us)
        constexpr int kPer = (kHistPerThread * kHistThreads / 64 + kCutWaves - 1) / kCutWaves;  // chunks a wave
        uint32_t e[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) e[j] = s.entries[min((c0 + j) * 64 + lane, g.n_samples - 1)];
#pragma unroll
        for (int j = 0; j < kPer; ++j) e[j] = c0 + j < c1 && (c0 + j) * 64 + lane < g.n_samples ? e[j] : 0u;
        uint32_t cnt = 0, sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            cnt += (uint32_t)__popcll(__ballot(e[j] != 0));
            sum += e[j] >> 15;
        }
        sum = WaveSum(sum);
        if (lane == 0) {
            s_wave_cnt[wave] = cnt;
            s_wave_sum[wave] = sum;
        }
        __syncthreads();
        uint32_t at = 0, n_all = 0, sum_all = 0;
        for (int w = 0; w < kCutWaves; ++w) {
            if (w < wave) at += s_wave_cnt[w];
            n_all += s_wave_cnt[w];
            sum_all += s_wave_sum[w];
        }
        // straight into the LDS table when the colours fit there (and are not few enough to skip
        // the median cut altogether)
        const bool to_lds = n_all > (uint32_t)kMaxColors && n_all <= (uint32_t)kCutLdsEntries;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned long long m = __ballot(e[j] != 0);
            const uint32_t to          = at + (uint32_t)__popcll(m & lt_mask);
            if (e[j]) {
                if (to_lds)
                    cut_lds[to] = e[j];
                else
                    s.tab_a[to] = e[j];
            }
            at += (uint32_t)__popcll(m);
        }
        if (tid == 0) {
            s_n     = n_all;
            s_total = sum_all;
        }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t n = s_n;

    if (n <= (uint32_t)kMaxColors) {
        // few enough colours: palette = the histogram colours, no diffusion
        __shared__ uint32_t s_pal_bytes0;
        if (tid == 0) s_pal_bytes0 = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += blockDim.x) {
            const uint32_t e     = s.tab_a[i];
            const uint32_t r = ((e >> 10) & 0x1f) << 3, gg = ((e >> 5) & 0x1f) << 3, bb = (e & 0x1f) << 3;
            s.palette[i * 3 + 0] = (uint8_t)r;
            s.palette[i * 3 + 1] = (uint8_t)gg;
            s.palette[i * 3 + 2] = (uint8_t)bb;
            atomicAdd(&s_pal_bytes0, (uint32_t)PaletteEntryLen((int)i, r, gg, bb));
        }
        __syncthreads();
        if (tid == 0) {
            s.meta[0] = (int)n;
            s.meta[1] = 0;
            s.meta[2] = (int)s_pal_bytes0;  // bytes of the palette in the output (K6)
        }
        return;
    }
    uint32_t *tab[2];
    const bool in_lds = n <= (uint32_t)kCutLdsEntries;
    if (in_lds) {
        tab[0] = cut_lds;  // (filled by the compaction above)
        tab[1] = cut_lds + kCutLdsEntries;
    } else {
        tab[0] = s.tab_a;
        tab[1] = s.tab_b;
    }
    // The box list.  libsixel keeps a vector sorted by pixel sum (descending, stable), replaces
    // the split box by its low half and appends the high half, then sorts again.  In the sorted
    // result the low half comes FIRST among boxes of equal sum (everything of its sum stood
    // behind the parent) and the high half LAST (it was appended), and nothing else changes
    // its relative order -- so the whole order is that of the key (sum, tie) with tie = -step
    // for the low half and +step for the high half of the step-th split.  Boxes therefore never
    // move: the low half takes the parent's slot, the high half the next free one, "the first
    // box of the list with >= 2 colours" is a maximum over keys, and the list positions are
    // only needed once, at the end, for the palette order.
    // Per slot one word in s_S: 0 for a box that cannot be split (one colour), else
    // order key << 3 | prepared << 2 | slot >> 6; the word of slot q * 64 + lane is s_S[lane * 4 + q]: wave 0 reads its
    // four slots at once, and boxes made one after the other stand in different lanes.
    //
    // The bookkeeping of a round, by wave 0, is NOT a loop over splits (it was: 24 rounds of twelve dependent
    // steps of ~110 instructions and an LDS round trip, 139 of the kernel's 250 us).  What libsixel does next is
    // decided by keys alone: it splits the prepared boxes in descending key order for as long as the next one's key
    // is above every box that is not prepared -- those of the list, and the halves made on the way, whose sums the
    // prepared records hold and whose ties follow from the position in that order.  So: the prepared boxes (s_ready,
    // at most 64) one per lane, ranked by all-pairs comparison, permuted into key order; the halves' keys per lane;
    // an exclusive prefix maximum over them; the splits that happen are the leading lanes whose key is above that
    // maximum and above the list's unprepared maximum -- and those lanes write their two halves at once.
    constexpr uint32_t kReady = 4u;
    __shared__ uint32_t s_pick[64], s_npick, s_pal_bytes;
    __shared__ CutTeamShared s_team;
    __shared__ alignas(16) uint32_t s_key[kMaxColors];
    __shared__ uint32_t s_rank[kMaxColors];
    __shared__ alignas(16) uint32_t s_S[kMaxColors];
    __shared__ alignas(16) uint32_t s_sortkey[64 + 8];  // (read eight at a time: a tail of zeros)
    __shared__ uint32_t s_ready[64 + 4];
    auto order_key = [](uint32_t sum, uint32_t tie /* 256 -+ step */) { return (sum << 9) | (511u - tie); };
    // wave 0: choose the boxes the next round prepares -- the first boxes of the list that can be split and have not
    // been prepared, as many as the ready list still holds -- mark them and append them to the list; returns its new
    // length.  Not a loop of wave maxima (34 instructions a pick): every lane's largest unprepared word is ranked
    // among the 64 of them (all-pairs, broadcast reads), the word of rank P - 1 is a threshold, and EVERY unprepared
    // box at or above it is picked -- the lane maxima of rank < P and whatever stands behind them in their lanes:
    // exactly the first boxes of the list, whatever their number (<= 4 P).
    uint32_t nb = 1, n_ready = 0;  // (wave 0's: boxes of the list, prepared boxes not yet split)
    auto pick = [&]() {
        const uint4 sv = *reinterpret_cast<const uint4 *>(&s_S[lane * 4]);
        uint32_t U[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) U[q] = (U[q] & kReady) ? 0u : U[q];
        const uint32_t best = max(max(U[0], U[1]), max(U[2], U[3]));
        s_sortkey[lane]     = best;
        TIMG_WAVE_SYNC();
        uint32_t rank = 0;
#pragma unroll
        for (int j = 0; j < 64; j += 4) {
            const uint4 k = *reinterpret_cast<const uint4 *>(&s_sortkey[j]);
            rank += (k.x > best ? 1u : 0u) + (k.y > best ? 1u : 0u) + (k.z > best ? 1u : 0u) + (k.w > best ? 1u : 0u);
        }
        const uint32_t room = 64u - n_ready;  // (>= 1: a round splits at least one box)
        // boxes at or above the lane maximum of rank P - 1 (every box that can be split, if there are fewer maxima)
        uint32_t c = 0, total = 0, thr = 0;
        auto at_or_above = [&](uint32_t P) {
            const unsigned long long hit = __ballot(best != 0 && rank == P - 1u);
            thr   = hit ? ReadLane(best, __builtin_ctzll(hit)) : 1u;
            c     = 0;
            total = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c += U[q] >= thr ? 1u : 0u;
                total += (uint32_t)__popcll(__ballot(U[q] >= thr));
            }
        };
        // (never more than the splits that remain to be made: what is prepared beyond them is thrown away)
        at_or_above(max(1u, min(min((uint32_t)kCutPickLanes, room), (uint32_t)kMaxColors - nb - min(n_ready, (uint32_t)kMaxColors - nb))));
        if (total > room) at_or_above(max(1u, room / 4u));  // (<= 4 boxes a lane: this one fits)
        const uint32_t incl = WaveInclusiveAdd(c);
        uint32_t at         = incl - c;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (U[q] >= thr) {
                const uint32_t slot = (uint32_t)q * 64u + (uint32_t)lane;
                s_pick[at]          = slot;
                s_ready[n_ready + at] = slot;
                s_S[lane * 4 + q]   = U[q] | kReady;
                ++at;
            }
        if (lane == 0) s_npick = total;
        n_ready += total;
    };
    auto s_index = [](uint32_t slot) { return (slot & 63u) * 4u + (slot >> 6); };
    if (tid < kMaxColors) s_S[tid] = tid == 0 && n >= 2u ? order_key(s_total, 256) << 3 : 0u;
    if (tid < 8) s_sortkey[64 + tid] = 0;
    if (tid == 0) {
        pool[0]     = CutRec{n << 16, 0, s_total | (256u << 16), 0};
        s_done      = 0;
        s_pal_bytes = 0;
    }
    __threadfence_block();
    __syncthreads();
    if (wave == 0) pick();
    __threadfence_block();
    __syncthreads();

#ifdef TIMG_CUT_TRACE
    // (kept in LDS and printed once at the end: a printf inside the loop costs more than a round)
    __shared__ long long s_tr_split[64], s_tr_book[64];
    __shared__ uint32_t s_tr_nb[64], s_tr_np[64];
    long long t_mark = wall_clock64();
    const long long t_setup = t_mark - t_kernel;
    int n_rounds = 0;
#endif
    for (;;) {
#ifdef TIMG_CUT_TRACE
        t_mark = wall_clock64();
#endif
        // ---- speculative splits: teams of waves for the few large boxes of the first rounds (TeamSplit), else wave w
        // prepares the picked boxes w, w + 12, ...
        const uint32_t n_picked = s_npick;
        uint32_t largest = 0;
        if (n_picked <= 2)  // (teams of three for four boxes came out at 7.3 us a round against 7.0: profiles/r4/cut_trace.txt)
            for (uint32_t p = 0; p < n_picked; ++p) largest = max(largest, pool[s_pick[p]].w0 >> 16);
        if (kCutWaves == 12 && largest >= kCutTeamMin) {  // (the same for every thread of the workgroup: barriers inside)
            const int T = n_picked == 1 ? 12 : 6, team = wave / T, tw = wave % T;
            const bool has_box  = (uint32_t)team < n_picked;
            const uint32_t slot = has_box ? s_pick[team] : 0u;
            const CutBox box    = has_box ? BoxOf(pool[slot]) : CutBox{0, 0, 0, 0};
            uint32_t *team_scratch = cut_lds + 2 * kCutLdsEntries + team * T * kCutScratch;
            uint32_t median, lowersum;
            if (in_lds)
                TeamSplit<true>(has_box, box, T, team, tw, wave, lane, tab, team_scratch, s_team, &median, &lowersum);
            else
                TeamSplit<false>(has_box, box, T, team, tw, wave, lane, tab, team_scratch, s_team, &median, &lowersum);
            if (has_box && tw == 0 && lane == 0) pool[slot].w2 = median | (lowersum << 16);
        } else
        for (uint32_t p = (uint32_t)wave, np = n_picked; p < np; p += kCutWaves) {
            const uint32_t slot = s_pick[p];
            const CutBox box    = BoxOf(pool[slot]);
            uint32_t median, lowersum;
            if (in_lds)
                SplitBox<true>(box, tab, scratch, lane, &median, &lowersum);
            else
                SplitBox<false>(box, tab, scratch, lane, &median, &lowersum);
            if (lane == 0) pool[slot].w2 = median | (lowersum << 16);
        }
        __threadfence_block();
        __syncthreads();
#ifdef TIMG_CUT_TRACE
        {
            const long long now = wall_clock64();
            if (tid == 0 && n_rounds < 64) {
                s_tr_split[n_rounds] = now - t_mark;
                s_tr_np[n_rounds]    = s_npick;
            }
            t_mark = now;
        }
#endif
        // ---- libsixel's serial bookkeeping for every prepared box it would take next (wave 0)
        if (wave == 0) {
            const uint4 sv      = *reinterpret_cast<const uint4 *>(&s_S[lane * 4]);
            const bool item     = (uint32_t)lane < n_ready;
            const uint32_t slot = item ? s_ready[lane] : 0u;
            // the list's largest key that is not prepared
            uint32_t u0;
            {
                const uint32_t S[4] = {sv.x, sv.y, sv.z, sv.w};
                uint32_t u          = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) u = max(u, (S[q] & kReady) ? 0u : S[q]);
                u0 = WaveMaxU32(u) >> 3;
            }
            const uint32_t key = item ? s_S[s_index(slot)] >> 3 : 0u;
            const uint2 rec    = *reinterpret_cast<const uint2 *>(&pool[slot]);  // w0, w2
            s_sortkey[lane]    = key;
            TIMG_WAVE_SYNC();
            uint32_t rank = 0;  // prepared boxes with a larger key (keys are distinct; lanes without a box: all of them)
            for (uint32_t j = 0; j < n_ready; j += 8) {
                const uint4 k0 = *reinterpret_cast<const uint4 *>(&s_sortkey[j]);
                const uint4 k1 = *reinterpret_cast<const uint4 *>(&s_sortkey[j + 4]);
                rank += (k0.x > key ? 1u : 0u) + (k0.y > key ? 1u : 0u) + (k0.z > key ? 1u : 0u) + (k0.w > key ? 1u : 0u);
                rank += (k1.x > key ? 1u : 0u) + (k1.y > key ? 1u : 0u) + (k1.z > key ? 1u : 0u) + (k1.w > key ? 1u : 0u);
            }
            // lane i takes the box of rank i (ds_permute: lane -> lane; the lanes without a box all aim at lane n_ready)
            const int to         = (int)(rank << 2);
            const uint32_t k_s   = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)key);
            const uint32_t sl_s  = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)slot);
            const uint32_t w0    = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)rec.x);
            const uint32_t w2    = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)rec.y);
            const uint32_t nb_i  = nb + (uint32_t)lane;                 // boxes in the list when this split happens
            const uint32_t head  = (w0 ^ 0x8000u) & 0xffffu;            // ind | the other table half << 15
            const uint32_t med   = w2 & 0xffffu, lower = w2 >> 16;
            const uint32_t c_hi  = (w0 >> 16) - med, s_hi = (k_s >> 9) - lower;
            // keys of the halves: tie = 256 - nb_i (low), 256 + nb_i (high)
            const uint32_t k_lo  = med >= 2u ? (lower << 9) | (255u + nb_i) : 0u;
            const uint32_t k_hi  = c_hi >= 2u ? (s_hi << 9) | (255u - nb_i) : 0u;
            const uint32_t above = WaveExclusiveMaxU32(item ? max(k_lo, k_hi) : 0u);  // halves made in front of this lane
            const bool valid     = item && k_s > u0 && k_s > above && nb_i < (uint32_t)kMaxColors;
            const unsigned long long stop = ~__ballot(valid);
            const uint32_t n_split        = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
            if ((uint32_t)lane < n_split) {
                *reinterpret_cast<uint4 *>(&pool[sl_s]) =
                    make_uint4(head | (med << 16), 0u, lower | ((256u - nb_i) << 16), 0u);
                *reinterpret_cast<uint4 *>(&pool[nb_i]) =
                    make_uint4((head + med) | (c_hi << 16), 0u, s_hi | ((256u + nb_i) << 16), 0u);
                s_S[s_index(sl_s)] = k_lo ? (k_lo << 3) | (sl_s >> 6) : 0u;
                s_S[s_index(nb_i)] = k_hi ? (k_hi << 3) | (nb_i >> 6) : 0u;
            } else if (item) {
                s_ready[(uint32_t)lane - n_split] = sl_s;  // still prepared, still in key order
            }
            nb += n_split;
            n_ready -= n_split;
            TIMG_WAVE_SYNC();
            uint32_t done = nb >= (uint32_t)kMaxColors ? 1u : 0u;
            if (!done) {
                pick();
                // no box with two colours left; (a round splits at least the box picked first -- it was the largest
                // unprepared one: a round without a split cannot happen, and must not become a hang if it does)
                done = n_ready == 0 || n_split == 0 ? 1u : 0u;
            }
            if (lane == 0) {
                s_nboxes = nb;
                s_done   = done;
            }
        }
        __threadfence_block();
        __syncthreads();
#ifdef TIMG_CUT_TRACE
        if (tid == 0 && n_rounds < 64) {
            s_tr_book[n_rounds] = wall_clock64() - t_mark;
            s_tr_nb[n_rounds]   = s_nboxes;
        }
        ++n_rounds;
#endif
        if (s_done) break;
    }
#ifdef TIMG_CUT_TRACE
    const long long t_loop_end = wall_clock64();
#endif
    const uint32_t nboxes = s_nboxes;
    // list position of every box: the number of boxes with a larger key
    if (tid < kMaxColors) {
        s_key[tid]  = (uint32_t)tid < nboxes ? order_key(pool[tid].w1 & 0xffffu, pool[tid].w1 >> 16) : 0u;
        s_rank[tid] = 0;
    }
    __syncthreads();
    {
        // (keys read four at a time, the reads independent of each other: one key per dependent iteration cost 5 us)
        const uint32_t box = tid & (kMaxColors - 1), part = tid / kMaxColors;
        static_assert(kCutWaves * 64 / kMaxColors == 3 && kMaxColors == 256, "three parts: 88 + 88 + 80 keys");
        const uint32_t j0 = part * 88u, j1 = min((uint32_t)kMaxColors, j0 + 88u);
        if (box < nboxes) {
            const uint32_t mine = s_key[box];
            uint32_t r = 0;
#pragma unroll 22
            for (uint32_t j = j0; j < j1; j += 4) {
                const uint4 k = *reinterpret_cast<const uint4 *>(&s_key[j]);  // (0 behind the last box: never larger)
                r += (k.x > mine ? 1u : 0u) + (k.y > mine ? 1u : 0u) + (k.z > mine ? 1u : 0u) + (k.w > mine ? 1u : 0u);
            }
            atomicAdd(&s_rank[box], r);
        }
    }
    __syncthreads();
#ifdef TIMG_CUT_TRACE
    const long long t_rank_end = wall_clock64();
#endif
    // SIXEL_REP_AVERAGE_COLORS: unweighted mean of the box's colours
    for (uint32_t bi = tid; bi < nboxes; bi += blockDim.x) {
        const CutBox box  = BoxOf(pool[bi]);
        const uint32_t at = s_rank[bi];
        uint32_t sum[3];
        if (in_lds)
            BoxColourSums((const CutLdsWord *)tab[box.buf] + box.ind, box.colors, sum);
        else
            BoxColourSums(tab[box.buf] + box.ind, box.colors, sum);
        const uint32_t r = (sum[0] / box.colors) & 0xffu, gg = (sum[1] / box.colors) & 0xffu, bb = (sum[2] / box.colors) & 0xffu;
        s.palette[at * 3 + 0] = (uint8_t)r;
        s.palette[at * 3 + 1] = (uint8_t)gg;
        s.palette[at * 3 + 2] = (uint8_t)bb;
        atomicAdd(&s_pal_bytes, (uint32_t)PaletteEntryLen((int)at, r, gg, bb));
    }
    __syncthreads();
    if (tid == 0) {
        s.meta[0] = (int)nboxes;
        s.meta[1] = 1;  // more colours than palette entries: diffuse
        s.meta[2] = (int)s_pal_bytes;  // bytes of the palette in the output (K6)
    }
#ifdef TIMG_CUT_TRACE
    __syncthreads();
    if (tid == 0 && (f == 0 || f == 5)) {
        const long long t_end = wall_clock64();
        long long sp = 0, bk = 0;
        for (int r = 0; r < n_rounds && r < 64; ++r) {
            printf("cut: f%d round %d picks %u split %lld book %lld -> %u boxes\n", f, r + 1, s_tr_np[r], s_tr_split[r],
                   s_tr_book[r], s_tr_nb[r]);
            sp += s_tr_split[r];
            bk += s_tr_book[r];
        }
        printf("cut: f%d n=%u setup %lld rounds %d split %lld book %lld rank %lld palette %lld total %lld (100 MHz ticks)\n", f,
               n, t_setup, n_rounds, sp, bk, t_rank_end - t_loop_end, t_end - t_rank_end, t_end - t_kernel);
    }
#endif
}

// ---- K3: 15-bit cell -> nearest palette entry ----------------------------------------
// The diffusion holds a channel c as the signed number c - 128 (see "Number format" there), so the 5-bit field it
// cuts out of a channel is (c >> 3) ^ 16: it indexes its tables with the BIASED cell = cell ^ kCellBias.
constexpr uint32_t kCellBias = 0x4210u;
// A wave takes a COARSE cell: the 4 x 4 x 4 cells that share the upper three bits of every channel, one per lane.  Not
// every palette entry can be the nearest one of some cell in that box: with d_min(e) / d_max(e) the smallest / largest
// squared distance from entry e to the box of the 64 centres, the entries worth comparing are those with
// d_min(e) <= min over e' of d_max(e') -- the nearest entry of any cell x is within d(x, e') <= d_max(e') of it for
// every e', and so is every entry at the same distance (ties go to the smaller index: the key's low byte).  The wave
// ranks the 256 entries once (four per lane), keeps the candidates (a tenth of the palette for a photograph), and
// every lane searches only those: the same minimum as the exhaustive search (until round 4: 256 entries for each of
// the 32 768 cells, 55 us per 64 frames at three quarters of the VALU peak).
constexpr int kLutWaves = 16;  // coarse cells per workgroup
__global__ void __launch_bounds__(kLutWaves * 64) BuildLutKernel(SixelGeom g, SixelBatch b) {
    const int f               = blockIdx.y;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    // |cell - entry|^2 = |cell|^2 + |entry|^2 - 2 <cell, entry>.  |cell|^2 is the same for
    // every entry, the byte dot product is one instruction (v_dot4_u32_u8), and with the
    // entry's number in the low 8 bits of the key
    //     key = (|entry|^2 << 8 | i) - 512 * <cell, entry>
    // "smallest distance, first of equally near entries" (strict < in libsixel's loop) is a
    // plain signed minimum.  All integer: the same ordering as the direct form.
    __shared__ uint2 pal[kMaxColors];              // {r | g << 8 | b << 16, |entry|^2 << 8 | i}
    __shared__ uint2 cand[kLutWaves][kMaxColors + 4];  // per wave: the entries worth comparing (+ padding to four)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncolors = s.meta[0];
    for (int i = tid; i < ncolors; i += kLutWaves * 64) {
        const uint32_t pr = s.palette[i * 3], pg = s.palette[i * 3 + 1], pb = s.palette[i * 3 + 2];
        pal[i] = make_uint2(pr | (pg << 8) | (pb << 16), ((pr * pr + pg * pg + pb * pb) << 8) | (uint32_t)i);
    }
    __syncthreads();
    // the wave's coarse cell (R3 G3 B3) and the box of its centres: channel values 32 C + 4 ... 32 C + 28
    const uint32_t coarse = blockIdx.x * kLutWaves + (uint32_t)wave;
    const int lo[3] = {(int)((coarse >> 6) & 7u) * 32 + 4, (int)((coarse >> 3) & 7u) * 32 + 4, (int)(coarse & 7u) * 32 + 4};
    uint32_t d_min[4], d_max_all = 0xffffffffu;
    uint2 mine[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = q * 64 + lane;
        mine[q]     = pal[i < ncolors ? i : 0];
        uint32_t dn = 0, dx = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int p    = (int)((mine[q].x >> (8 * ch)) & 0xffu);
            const int near = max(0, max(lo[ch] - p, p - (lo[ch] + 24)));
            const int far  = max(abs(p - lo[ch]), abs(p - (lo[ch] + 24)));
            dn += (uint32_t)(near * near);
            dx += (uint32_t)(far * far);
        }
        d_min[q]  = i < ncolors ? dn : 0xffffffffu;
        d_max_all = min(d_max_all, i < ncolors ? dx : 0xffffffffu);
    }
    const uint32_t reach = ~WaveMaxU32(~d_max_all);  // the smallest d_max: some entry is this near to EVERY cell of the box
    uint32_t n_cand = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool keep              = d_min[q] <= reach;
        const unsigned long long set = __ballot(keep);
        if (keep) cand[wave][n_cand + (uint32_t)__popcll(set & ((1ull << lane) - 1ull))] = mine[q];
        n_cand += (uint32_t)__popcll(set);
    }
    TIMG_WAVE_SYNC();
    {  // (the list is walked four at a time: padded with its first entry, which changes no minimum)
        const uint2 c0 = cand[wave][0];
        if (lane < 3) cand[wave][n_cand + (uint32_t)lane] = c0;
        TIMG_WAVE_SYNC();
    }
    // this lane's cell: the coarse cell's bits above two own bits per channel
    const uint32_t cell = (((coarse >> 6) & 7u) << 12) | ((((uint32_t)lane >> 4) & 3u) << 10) | (((coarse >> 3) & 7u) << 7) |
                          ((((uint32_t)lane >> 2) & 3u) << 5) | ((coarse & 7u) << 2) | ((uint32_t)lane & 3u);
    const uint32_t me = (((cell >> 10) & 0x1f) << 3 | 4) | (((cell >> 5) & 0x1f) << 3 | 4) << 8 | ((cell & 0x1f) << 3 | 4) << 16;
    int k = 0x7fffffff;
    for (uint32_t i = 0; i < n_cand; i += 4) {
        uint2 c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = cand[wave][i + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) k = min(k, (int)c[q].y - 512 * (int)__builtin_amdgcn_udot4(me, c[q].x, 0u, false));
    }
    const int best = k & 255;
    s.lut[cell] = (uint32_t)best | (pal[best].x << 8);
    s.lut8[cell ^ kCellBias] = (uint8_t)best;
    // the hand-over counters of a diffusion spread over several CUs start at zero (the kernel boundary in
    // front of DitherKernel publishes these plain stores)
    if (blockIdx.x == 0 && threadIdx.x < kDitherMaxParts - 1)
        b.xwg[((size_t)f * (kDitherMaxParts - 1) + threadIdx.x) * XwgStride(g.w) + XwgData(g.w)] = 0u;
}

// ---- K4: lookup + Floyd-Steinberg -----------------------------------------------------
// libsixel diffuses in place on 8-bit data: every contribution is added with
// C integer division (err * num / 16, truncating toward zero) and clamped to
// 0..255 before the next one arrives, so order and rounding of each term matter.
// The pixel that PRODUCES an error computes its four terms once (7/16 right,
// 3/16 below-left, 5/16 below, 1/16 below-right); consumers only add and clamp.
//
// One workgroup per frame, one wave per 32 consecutive rows.  Inside a wave row y
// runs two columns behind row y-1 (the minimum Floyd-Steinberg allows: pixel (x,y)
// needs e(x+1,y-1)) and receives the terms of the row above through DPP moves.
// Between waves the last row of wave k publishes its terms in an LDS boundary row
// plus a progress counter, and the first row of wave k+1 follows it as closely as
// the data allows: the waves of a frame form a pipeline, so a frame costs about
// W + 2*(H-1) steps.  Frames with more rows than the waves cover go round again
// (wave 0 then follows the last wave of the previous round).
//
// A wave issues in order and the chain of steps is serial, so the kernel is bound by
// the INSTRUCTIONS OF ONE STEP (measured: ~7 cycles per instruction for a wave alone on
// its SIMD), not by bandwidth.  Hence:
//  * a row is handled by a PAIR of lanes, the even lane carrying (r, g), the odd lane
//    (b, -), as packed 16-bit values: every v_pk_*_i16 instruction serves two channels
//    and the pair does in one instruction stream what a single lane did in three;
//  * a number format in which "add, clamp to 0..255" is one saturating add (below);
//  * straight-line steps: every lane computes, in range or not (loads clamped, tables
//    indexed with clamped values), only stores are predicated -- the compiler can then
//    schedule across the two dependent LDS lookups;
//  * UNCONDITIONAL source-pixel prefetch 8 steps ahead: a load inside a branch makes the
//    compiler lose count of what is in flight and wait for vmcnt(0) -- the prefetch issued
//    a step earlier -- in every step, a memory round trip per step (that, not the
//    arithmetic, bounded the first version of this kernel);
//  * all hand-over traffic is LDS traffic of the form "data, then counter" from ONE wave,
//    which the LDS executes in order: no fence (a workgroup fence would drain the wave's
//    global prefetches), only relaxed atomics and compiler barriers.
constexpr int kDitherAhead    = 8;  // source pixels are requested this many steps early
// How far past the end of the row above a step reads (unclamped -- the lanes that receive those records are outside
// their own rows by then): the skew of the wave's last row pair (2 columns a row), the request lead, one unrolled body
// of 8 steps, the record look-ahead (t + 3) and the pair's second half -- 62 + 8 + 8 + 7 + 5 = 90 slots with today's
// constants.  Changing any of them must move kDitherOverrun (sixel_launch.h) with it: ADVICE r4.
static_assert(kDitherOverrun >= 2 * (kPairRows - 1) + kDitherAhead + 8 + 7 + 5, "record reads past a row's end must stay inside the slack");

// (0x138 below: wave_shr:1 -- every lane receives the value of the lane below it in index)

typedef short PairI16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ PairI16 AsPair(uint32_t u) { return __builtin_bit_cast(PairI16, u); }
__device__ __forceinline__ uint32_t AsBits(PairI16 v) { return __builtin_bit_cast(uint32_t, v); }

// the same half of the pair one row up; the two lanes of the wave's first row, which have no row above them in the wave,
// receive lane 0's `first` (lane 1) and `second` (lane 0)
__device__ __forceinline__ uint32_t DownOneRow(uint32_t v, uint32_t first, uint32_t second) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)second, (int)t, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t FromPairPartner(uint32_t v) {  // quad_perm [1,0,3,2]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);
}
typedef unsigned short PairU16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t AsBits(PairU16 v) { return __builtin_bit_cast(uint32_t, v); }
// add one contribution and clamp: see "Number format" in the kernel
__device__ __forceinline__ PairI16 ApplyPair(PairI16 v, uint32_t q) {
    return __builtin_elementwise_add_sat(v, AsPair(q));
}

constexpr int kDitherSpinLimit = 1 << 22;
// Timing experiments only (scratch/build_variant.sh ... -DTIMG_DITHER_ABL=<bits>; results are garbage, addresses stay
// in range, nothing blocks): 1 no waiting for the producer, 2 no palette-index load, 4 no pixel load, 8 no index
// store, 16 no table reads, 32 no hand-down (and no helper waves), 64 no record reads, 128 no wait for the pixel, 256 no helper waves (with 1).  profiles/r4/dither_ablation.txt
#ifndef TIMG_DITHER_ABL
#define TIMG_DITHER_ABL 0
#endif
constexpr int kDitherAbl = TIMG_DITHER_ABL;
// The helper waves (flusher, fetcher) share their SIMDs with diffusing waves: what they issue, the diffusion does not
// (every row waits for the one above it: the slowest wave sets the pace).  Polling every 128 clocks and moving a
// boundary row column by column they cost a 64-frame batch 4 % (profiles/r4/dither_ablation.txt); they nap
// g.helper_naps x 128 clocks between polls and move pieces of >= g.helper_batch columns -- the launch chooses: longer
// naps where a frame has few parts (a part boundary then lags by the piece), shorter ones where it has many.
__device__ __forceinline__ void HelperNap(int naps) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(2);
}

// kNarrow: frames of 1 or 2 columns, where the row-wrap term and the 7/16 term come from
// the same pixel and arrive in the other order
//
// kSplit: the frame's waves are spread over gridDim.x workgroups = CUs (part p takes the p-th share of the
// frame's row groups; every wave of the frame exists at once, no rounds).  With 13-15 waves on one CU the
// SIMDs are saturated by the instructions of the steps (0.38 us per step against 0.29-0.32 with one or two
// waves per SIMD: profiles/r2/sixel_rows_sweep.txt).  The boundary between two parts travels through memory,
// and NOT through the waves that diffuse: the last wave of part p publishes in LDS exactly as it does for a
// wave of its own workgroup, a helper wave (the "flusher") follows that counter, writes the finished slots
// through to memory (sc0 sc1 stores, drained, then the counter: MI355X_MICROARCH.md, inter-workgroup
// visibility) and in part p + 1 another helper (the "fetcher") polls that counter, copies the new slots into
// ITS workgroup's boundary row 0 and publishes them in LDS -- the diffusing waves run the same code in either
// mode and never wait for a memory round trip.  Block x = part is the fastest index: a part is dispatched
// after the part it follows, and every wait is bounded (kDitherSpinLimit -> the batch's error word).
//
// kOneTrip: the lookup form (sixel_launch.h).  The chain of a step runs pixel -> cell -> palette colour -> error ->
// the next pixel of the row; with the cell's COLOUR in LDS (as signed bytes p - 128) that is one LDS round trip, and the
// palette index, which only the output needs, is not looked up here at all: the index image takes the CELL (two bytes a
// pixel, SixelGeom::idx_shift) and the band kernel turns cells into indices when it reads them;
// the two-trip form reads cell -> index -> colour from 34 KB of tables and leaves the LDS to the boundary rows.
//
// kPix2: the pixels are requested two at a time (one global_load_dwordx2 every second step).  A wave's 32 rows make
// every pixel request 32 cache lines, 66 clocks of issue (scratch/ubench/wave_latency.hip: iss_gload_rows) in a step
// of ~340; a lane's column advances by one per step and starts even, so an even frame width (and 8-byte aligned
// rows) makes the pair (x, x + 1) one aligned load that never straddles the row's end.  One-trip form only.
template <bool kNarrow, bool kSplit, bool kOneTrip, bool kPix2 = false>
__global__ void __launch_bounds__(kDitherMaxWaves * 64) DitherKernel(SixelGeom g, SixelBatch b) {
    static_assert(!(kNarrow && kOneTrip), "the narrow kernel keeps the small tables");
    static_assert(!kPix2 || kOneTrip, "pixel pairs: one-trip form only");
    constexpr int kDitherTabWords = DitherTabWords(kOneTrip);
    // All of the kernel's LDS is dynamic and starts at address 0: [progress counters: kDitherLdsHead bytes][tables]
    // [boundary rows][slack].  (With a static array in front, every LDS address the steps form from a running offset
    // paid a v_add of the static part's size.)
    extern __shared__ uint32_t lds_all[];
    uint32_t *lds = lds_all + kDitherLdsHead / 4;
    const int W = g.w, H = g.h6;
    const int n_waves  = blockDim.x >> 6;
    // roles (kSplit): [fetcher (parts > 0)] [nd diffusing waves] [flusher (parts before the last)] [idle]
    const int parts    = kSplit ? (int)gridDim.x : 1;
    const int part     = kSplit ? (int)blockIdx.x : 0;
    const int n_all    = (H + kPairRows - 1) / kPairRows;  // row groups of the frame
    const int nd       = kSplit ? n_all / parts + (part < n_all % parts ? 1 : 0) : n_waves;
    const int group0   = kSplit ? part * (n_all / parts) + min(part, n_all % parts) : 0;  // first row group of this part
    const int lw0      = (kSplit && part > 0) ? 1 : 0;  // local index of the first diffusing wave
    const int n_local  = lw0 + nd;                      // boundary rows written in this workgroup
    const bool flushes = kSplit && part < parts - 1;
    const int n_pad    = H - g.h;
    // one trip: three byte tables indexed by the biased cell -- b, r, g, 32 KB each: a lane reads [its base + cell] and
    // 32 KB further on with ONE address (the even lane of a pair r then g, the odd lane b and a byte nobody uses).
    // Two trips: [biased cell] -> palette index, then pal[2][256]: 16 * colour as (r, g) / (b, 0)
    uint8_t *tab8      = reinterpret_cast<uint8_t *>(lds);
    uint32_t *pal      = lds + 8192;
    // Boundary rows, one per wave plus one that stays zero (what a wave with no row above it reads).  A row holds one
    // 12-byte RECORD per column of the producer's last row -- the three terms its error sends down -- at slot x + 1:
    //     bytes 0..3   3/16 as (0, r, 0, g)            } written by the even lane of the pair as ONE two-word store
    //     bytes 4..7   1/16 as (r, g), 5/16 as (r, g)  }
    //     bytes 8..11  (0, 3/16, 1/16, 5/16) of b      -- the odd lane's store, whose upper half spills into the next
    //                                                     slot's first word and is overwritten a step later
    // (A lane's first word is its 3/16 as a TERM WORD -- the one term on a step's serial chain is added as it arrives; the
    // odd lane's other two terms ride in the half of its pair that carries no channel, where what they add up to is never
    // looked at.)
    // Slot 0 (column -1) stays zero; column W -- outside the row, all terms zero -- is written like any other, so that
    // what spilled into slot W + 1 is replaced before the consumer's last column reads it: W + 3 slots, and W + 1
    // columns to publish.  The consumer reads one record per step (column c needs 1/16 of c - 1, 5/16 of c, 3/16 of
    // c + 1) and keeps what it unpacked until its column comes.  (Until round 4 the producer scattered its terms
    // over the three slots of their consumers, six bytes at a time in three stores: 28 clocks of LDS issue a step
    // against 9, scratch/ubench/wave_latency.hip.)
    const int brow     = (W + 3) * 3;
    const int n_pub    = W + 1;                     // columns a row publishes
    uint32_t *boundary = lds + kDitherTabWords;  // [n_local + 1][W + 3][3]
    int *progress = reinterpret_cast<int *>(lds_all);  // [kDitherMaxWaves]
    // The steps' running addresses are absolute LDS addresses (this array's own address added once, outside the
    // loops) used through address-space-3 pointers: formed as `array + offset` each step paid a v_add of the array's
    // link-time address.
    typedef __attribute__((address_space(3))) uint32_t LdsU32;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_all;
    const int f   = kSplit ? blockIdx.y : blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (in a scalar register: what depends on it branches for free)
    const int rl   = lane >> 1;        // row inside the wave
    const bool odd = (lane & 1) != 0;  // the (b, -) half
    // Number format.  A channel value c (0..255) is held as the signed 16-bit number
    // (c - 128) << 8 | junk, an error term q as q << 8: then "add, clamp to 0..255" -- what
    // libsixel does for every single contribution -- is ONE saturating 16-bit add (a sum
    // above 255 hits 0x7fff, below 0 hits 0x8000, anything else leaves the junk byte alone).
    // v_perm_b32 selector of this lane's half (0x0c = zero byte): bytes (r, g) / (b) of a pixel
    // or of a term word -> HIGH byte of each 16-bit half
    const uint32_t sel_hi   = odd ? 0x0c0c020cu : 0x010c000cu;
    const uint32_t px_bias  = odd ? 0x00008000u : 0x80008000u;  // c -> c - 128 (the unused half stays 0)
    // the lane's share of the cell as ONE v_dot2_u32_u16 of its two 5-bit fields: r5 << 10 | g5 << 5 / b5
    const uint32_t cell_mul = odd ? 0x00000001u : 0x00200400u;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const uint8_t *frame      = b.fb + (size_t)f * g.frame_stride;
    for (int i = tid; i < 8192; i += blockDim.x) {  // four cells a turn; the bias leaves the low 2 bits alone
        const uint4 v = reinterpret_cast<const uint4 *>(s.lut)[i ^ (kCellBias >> 2)];  // idx | r << 8 | g << 16 | b << 24
        if (kOneTrip) {
            // (as SIGNED bytes p - 128 = p ^ 0x80: the step holds a channel as c - 128 and subtracts like from like)
            lds[i]         = ((v.x >> 24) | ((v.y >> 24) << 8) | ((v.z >> 24) << 16) | ((v.w >> 24) << 24)) ^ 0x80808080u;                              // b
            lds[8192 + i]  = (((v.x >> 8) & 0xffu) | (((v.y >> 8) & 0xffu) << 8) | (((v.z >> 8) & 0xffu) << 16) | (((v.w >> 8) & 0xffu) << 24)) ^ 0x80808080u;    // r
            lds[16384 + i] = (((v.x >> 16) & 0xffu) | (((v.y >> 16) & 0xffu) << 8) | (((v.z >> 16) & 0xffu) << 16) | (((v.w >> 16) & 0xffu) << 24)) ^ 0x80808080u;  // g
        } else {
            lds[i] = (v.x & 0xffu) | ((v.y & 0xffu) << 8) | ((v.z & 0xffu) << 16) | ((v.w & 0xffu) << 24);
        }
    }
    if (!kOneTrip)
        for (int i = tid; i < 256; i += blockDim.x) {
            const bool in = i < s.meta[0];
            const uint32_t pr = in ? s.palette[i * 3] : 128, pg = in ? s.palette[i * 3 + 1] : 128,
                           pb = in ? s.palette[i * 3 + 2] : 128;
            pal[i]       = (pr * 16) | ((pg * 16) << 16);
            pal[256 + i] = pb * 16;
        }
    // the rows SixelCanvas::Send appends below the frame, as pixels in scratch memory: the
    // steps then fetch every row the same way (written and read by this workgroup only)
    uint32_t *pad_rows = b.pad_rows + (size_t)f * 5 * W;
    for (int i = tid; i < n_pad * W; i += blockDim.x) pad_rows[i] = PaddedPixel(frame, g, i % W, g.h + i / W);
    // (the row that stays zero: row n_local -- or, for a workgroup of one wave, that wave's own row: sixel_launch.h)
    const int zero_row = n_local > 1 ? n_local : 0;
    for (int i = tid; i < (zero_row + 1) * brow; i += blockDim.x) boundary[i] = 0u;
    if (tid < n_local) progress[tid] = 0;
    if (tid < kDitherMaxWaves)  // the waves' spin counts (wait_for below): in the slack behind the rows, past the lanes' dummy stores
        boundary[(zero_row + 1) * brow + 192 + tid] = 0u;
    const bool dither = s.meta[1] != 0;
    __syncthreads();

    if (kSplit) {
        if (wave >= n_local + (flushes ? 1 : 0)) return;  // (the block is sized for the largest part)
        if ((kDitherAbl & (32 | 256)) && ((flushes && wave == n_local) || (lw0 && wave == 0))) return;
        if (flushes && wave == n_local) {
            // ---- flusher: LDS boundary row of the part's last wave -> memory, as its counter advances
            const uint32_t *row = boundary + (size_t)(n_local - 1) * brow;
            uint32_t *xw        = b.xwg + ((size_t)f * (kDitherMaxParts - 1) + part) * XwgStride(W);
            int *xflag          = reinterpret_cast<int *>(xw + XwgData(W));
            int done = 0, spins = 0;
            for (;;) {
                const int p = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&progress[n_local - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                asm volatile("" ::: "memory");
                // (slot x + 1 is final once the producer has finished column x; the last one with the row)
                const int limit = p >= n_pub ? W + 3 : p + 1;
                if (limit >= done + g.helper_batch || (limit > done && p >= n_pub)) {
                    for (int i = done * 3 + lane; i < limit * 3; i += 64)
                        __hip_atomic_store(xw + i, row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slots have left before the counter does
                    if (lane == 0) __hip_atomic_store(xflag, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    done = limit;
                    if (p >= n_pub) break;
                } else {
                    HelperNap(g.helper_naps);
                    if (++spins > kDitherSpinLimit) {
                        if (lane == 0) atomicExch(b.error, 1);
                        break;
                    }
                }
            }
            return;
        }
        if (lw0 && wave == 0) {
            // ---- fetcher: the previous part's boundary row, memory -> this workgroup's boundary row 0
            uint32_t *row      = boundary;
            const uint32_t *xw = b.xwg + ((size_t)f * (kDitherMaxParts - 1) + part - 1) * XwgStride(W);
            const int *xflag   = reinterpret_cast<const int *>(xw + XwgData(W));
            int done = 0, spins = 0;
            for (;;) {
                const int p     = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(xflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                const int limit = p >= n_pub ? W + 3 : p + 1;
                if (limit > done) {
                    for (int i = done * 3 + lane; i < limit * 3; i += 64)
                        row[i] = __hip_atomic_load(xw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("" ::: "memory");  // (data, then the counter, from one wave: the LDS keeps the order)
                    if (lane == 0)
                        __hip_atomic_store(&progress[0], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = limit;
                    if (p >= n_pub) break;
                } else {
                    HelperNap(g.helper_naps);
                    if (++spins > kDitherSpinLimit) {
                        if (lane == 0) {
                            atomicExch(b.error, 1);
                            // (let the waves behind this one run to their end instead of into their own limit)
                            __hip_atomic_store(&progress[0], n_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        break;
                    }
                }
            }
            return;
        }
    }

#ifdef TIMG_DITHER_PRIO
    __builtin_amdgcn_s_setprio(3);  // (experiment: the diffusing waves above the helper waves they share a SIMD with)
#endif
    // this lane's half of the tables
    const uint32_t tab_base  = odd ? 0u : 32768u;                 // (one trip: b / r, and g 32 KB behind r)
    const uint32_t *pal_half = pal + (odd ? 256 : 0);             // (two trips)
    // Record <-> term words (a term word is this half's pair of q << 8: 0xGG00RR00 / 0x0000BB00).  Packing, two
    // v_perm: w1 = (1/16, 5/16) of this half [odd: b of both, still without the 3/16]; w2 = the 3/16 [odd: w1's two
    // bytes and the 3/16 -- the finished word]; the 8-byte store takes (w2, w1).  Unpacking, one v_perm per term
    // from the 8 bytes read (v_perm_b32(hi, lo, .): bytes 0-3 of the selector's index space are lo, 4-7 hi).
    const uint32_t sel_w1 = odd ? 0x0c0c0501u : 0x07050301u;
    const uint32_t sel_w2 = odd ? 0x0100050cu : 0x070c050cu;  // (the 3/16 as a term word; odd: 1/16 and 5/16 of b in the unused half)
    const uint32_t sel_u1 = odd ? 0x0c0c020cu : 0x050c040cu;
    const uint32_t sel_u5 = odd ? 0x0c0c030cu : 0x070c060cu;

    const int rows_per_round = (kSplit ? n_all : n_waves) * kPairRows;  // (kSplit: one round)
    const int first_row      = (group0 + wave - lw0) * kPairRows;
    // a row's W pixels, skewed by two columns per row; the palette index of a pixel arrives kDitherAhead steps
    // after the pixel's own step, and a row's last group of four indices is completed by up to three junk ones
    // (the index rows are padded to a multiple of four and nobody reads the pad)
    // (+ 8: a row's last group of eight indices is completed by up to seven junk ones -- nobody reads the row's padding)
    const int steps          = W + 2 * (kPairRows - 1) + kDitherAhead + 8;
    // the steady blocks (see step below): t = 64, 72, ... <= W - 16 -- the last row of the wave has begun (t >= 62), the
    // first has not reached its last column (t + 7 <= W - 2), the pixels requested kDitherAhead steps early exist
    // (t + 7 + kDitherAhead <= W - 1); rows too short for one: none
    static_assert(2 * (kPairRows - 1) <= 64 && kDitherAhead == 8, "the steady blocks' range is written for these constants");
    const bool has_steady    = !kNarrow && W - 16 >= 64;
    const int t_steady0      = has_steady ? 64 : steps;
    const int t_steady1      = has_steady ? ((W - 16) / 8) * 8 + 8 : steps;
    for (int round = 0; round * rows_per_round + first_row < H; ++round) {
        const int row      = round * rows_per_round + first_row + rl;
        const bool has_row = row < H;
        const uint8_t *src_row  = row < g.h ? frame + (size_t)row * g.stride
                                            : reinterpret_cast<const uint8_t *>(pad_rows + (size_t)(min(row, H - 1) - g.h) * W);
        // (the lane's index row as a 32-bit offset from the frame's index image, already moved back to where the group
        // of four that ends kDitherAhead columns behind x begins: base + offset addressing, no 64-bit arithmetic)
        // At the last step of a block of eight (t + 7) the eight indices of columns t - 8 - 2 rl ... t - 1 - 2 rl are
        // complete: bytes idx_q ... idx_q + 7 of the lane's index row, idx_q = t - 8 - 8 (rl >> 2) -- IndexRow()'s shift
        // of 2 (rl & 3) is what makes it a multiple of eight for every row (rl = row mod 32).
        // One trip: the image holds the pixels' CELLS, two bytes each (the palette index of a cell is a byte in memory: asked
        // for here it was a row-scattered load a step -- 36 clocks of issue in the one instruction stream that sets the
        // kernel's pace; the band kernel, which reads the image once, looks the indices up instead).
        constexpr uint32_t kIdxBytes = kOneTrip ? 2u : 1u;
        const int idx_px        = g.idx_stride / (int)kIdxBytes;  // a row's pixels incl. padding
        int idx_q               = -8 - 8 * (rl >> 2);
        uint32_t idx_addr       = (uint32_t)min(row, H - 1) * (uint32_t)g.idx_stride + (uint32_t)idx_q * kIdxBytes;
        const bool diffuses     = dither && row < H - 1;
        // a lane that never spreads an error (no row, the last row, an exact palette) multiplies by zero:
        // 16 * err = 16 * c - 16 * p as ONE v_pk_mad_u16 of the table bytes
        const bool lane_spreads = has_row && diffuses;
        // (one trip: times -16, and the byte the odd lane reads second meets a 0; two trips: the table holds 16 * p)
        const uint32_t k_err    = !lane_spreads ? 0u : !kOneTrip ? 0xffffffffu : odd ? 0x0000fff0u : 0xfff0fff0u;
        const uint32_t c16_mask = lane_spreads ? (odd ? 0x00000ff0u : 0x0ff00ff0u) : 0u;
        // one trip: err = (c - 128) - (p - 128) as a pair of signed 16-bit numbers, and 16 * n * err + sign bias for the four
        // terms with the 16 * n as LANE constants (0 in a lane that never spreads, 0 in the odd lane's half without a channel)
        const uint32_t k_lane   = !lane_spreads ? 0u : odd ? 0x0000ffffu : 0xffffffffu;
        const uint32_t k112 = 0x00700070u & k_lane, k80 = 0x00500050u & k_lane, k48 = 0x00300030u & k_lane, k16 = 0x00100010u & k_lane;
        const uint32_t sgn_mask = 0x00f000f0u & k_lane;
        // which lanes complete a group of four indices at the steps k & 3 == 3 / k & 3 == 1 of the unrolled body
        const bool stores_idx   = has_row && !odd;
        const bool hands_down   = has_row && rl == kPairRows - 1;  // the row above the next wave's first row
        // (kSplit: local wave 0 is either the frame's first wave or the fetcher, which writes row 0 like a wave)
        const int producer       = wave == 0 ? n_local - 1 : wave - 1;
        const int producer_round = wave == 0 ? round - 1 : round;
        const bool follows       = producer_round >= 0;
        // Addresses that advance by a constant per step are kept per BLOCK of eight steps (in_addr, out_addr, idx_addr
        // below, bumped at the end of the unrolled body): the steps then address with immediate offsets instead of a
        // multiply-add each.  in_addr: this half's part of slot t of the row above (absolute LDS address).
        uint32_t in_addr         = lds0 + (uint32_t)kDitherLdsHead +
                                   (uint32_t)(kDitherTabWords + (follows ? producer : zero_row) * brow) * 4u + (odd ? 8u : 0u);
        // out_addr: this half's part of the record of column x = t - 2 * rl in this wave's own row
        const uint32_t my_row    = lds0 + (uint32_t)kDitherLdsHead + (uint32_t)(kDitherTabWords + wave * brow) * 4u + (odd ? 8u : 0u);
        uint32_t out_addr        = my_row + 12u - 24u * (uint32_t)rl;
        // Handing down WITHOUT touching exec: every lane packs and stores a record and a progress value, every step -- the
        // lanes that have nothing to hand down (all but one pair of the wave) into a slot of their own in the slack behind
        // the rows, the pair that has with its address clamped to its row: column -1 (zeros into slot 0, which holds
        // zeros) while it has not started, slot W + 2 (nobody reads it) once it is through.  (The predicated form --
        // v_cmp, s_and, s_and_saveexec, branch, s_and exec, s_or exec around five instructions of work -- cost every
        // step of every wave ~75 clocks for one row in thirty-two: scratch/ubench/wave_latency.hip.)
        const uint32_t slack     = lds0 + (uint32_t)kDitherLdsHead + (uint32_t)(kDitherTabWords + (zero_row + 1) * brow) * 4u;
        const uint32_t rec_lo    = hands_down ? my_row : slack + 8u * (uint32_t)lane;                      // slot 0 = column -1
        const uint32_t rec_hi    = hands_down ? my_row + 12u * (uint32_t)(W + 2) : slack + 8u * (uint32_t)lane;
        const uint32_t prog_addr = (hands_down && !odd) ? lds0 + 4u * (uint32_t)wave : slack + 512u + 4u * (uint32_t)lane;
        // (steady blocks: the record's address without the clamp -- the row's own for the pair that hands down, the dummy
        // slot, which does not advance, for everybody else; a dummy record at + 12 k overlaps other lanes' dummies)
        uint32_t out_steady      = 0;
        const uint32_t out_pace  = hands_down ? 96u : 0u;
        int prog_run             = round * n_pub + 1 - 2 * rl;  // (+ t: columns this row has finished, before clamping)
        const int prog_lo = round * n_pub, prog_hi = round * n_pub + n_pub;
        // (a wave with no row above it finds every column "published": its own counter, against a base far below)
        const int in_base        = follows ? producer_round * n_pub : -(1 << 30);

        // terms of this row's own recent errors: own7 = 7/16 of e(x-1) as a pair of q << 8 -> this row's next pixel.
        // What goes to the row BELOW travels as the two words of the column's RECORD (w2, w1: the format of the boundary
        // rows above), moved down one row -- two lanes -- a step after they were packed and unpacked by the receiver, one
        // v_perm per term: 3/16 from the record that arrived this step (column x + 1 of the row above), 5/16 from the one
        // before (column x), 1/16 from the one before that (column x - 1).  The wave's first row takes the same words
        // from the boundary row: lane 0 hands its two reads (and its partner's first word) to the lane moves as the value
        // of the lanes that have no lane above them, so one instruction stream serves every row.  (Until round 6 the
        // three terms were moved as three registers, two v_mov_dpp each, the first row chose between them and its
        // unpacked boundary terms with three v_cndmask, and every term was masked for the add: 16 instructions of a
        // step for what takes 9.)
        //
        // A step's serial chain is: the row above's record of a step ago -> two lane moves -> 3/16 -> pixel value -> cell ->
        // table bytes (an LDS round trip) -> error -> terms -> record.  Everything else of the NEXT step's pixel value --
        // the pixel's bytes into the number format, the 1/16 and 5/16 from records that arrived earlier -- is computed a
        // step ahead (v_pre), in the shadow of the table reads: a wave alone on its SIMD issues in order, so whatever
        // stands in front of the table reads delays them, and whatever stands behind them is free while they travel.
        uint32_t own7 = 0;
        uint32_t w1p = 0, w2p = 0;      // the record this row packed a step ago (stored for the next wave a step late, see below)
        uint32_t q_pt = 0;              // lane 1's first word of the boundary record in lane 0 (prepared a step ahead)
        uint32_t s1a = 0, s1b = 0;      // the record that arrived a step ago (a: w1, b: w2)
        PairI16 v_pre = {0, 0};         // this step's pixel with the 1/16 and 5/16 from above added
        uint32_t first_q3 = 0;
        uint32_t pk_lo = 0, pk_hi = 0;  // the last eight indices, the newest in pk_hi's top byte (two trips)
        uint32_t pkc[4] = {0, 0, 0, 0};  // the block's eight cells, two a word (one trip)
        uint32_t n_lo = 0, n_hi = 0;    // the boundary record requested a step ago (column t + 1 at step t)
        // The producer's progress counter is read EVERY step, one step before it is looked at (two instructions, no
        // wait: the value has long arrived), so a wave follows its producer as closely as the data allows and the
        // common case -- the producer is far enough -- is a branch that is NOT taken.  (Measured with
        // scratch/ubench/wave_latency.hip: a wave alone on its SIMD issues one instruction of any kind per 4 clocks,
        // dependent or not, but a TAKEN branch costs ~80; the previous form polled in batches of four columns behind
        // a taken branch per step: ~80 clocks a step for the branch and ~50 a step for the polls, of ~600.)
        // The check itself is one scalar add, a compare and the branch: the value the counter must have reached is kept
        // running (lim_run, + 8 a block), and the path that waits keeps NO state in registers -- its spin count lives in
        // an LDS word of the wave's own, it reports a give-up where it happens.  (With a spin count and a flag carried
        // through the steps the compiler put their copies -- three scalar moves -- on the path that does not wait, and
        // computed the threshold from the column with a min every time: 8 scalar instructions a check, of a step's 55.)
        uint32_t prog_raw = 0;
        int lim_run       = in_base + 4;               // (+ t + k: published columns before record t + k + 3 is requested)
        const int lim_cap = in_base + n_pub;           // (a row publishes no more)
        LdsU32 *const spun = (LdsU32 *)(uintptr_t)(slack + 768u + 4u * (uint32_t)wave);
        auto peek = [&]() __attribute__((always_inline)) {
            prog_raw = (uint32_t)__hip_atomic_load(&progress[producer], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto wait_for = [&](int lim) __attribute__((always_inline)) {
            if (kDitherAbl & 1) return;
            if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)prog_raw) < lim, 0)) {
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&progress[producer], __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_WORKGROUP)) >= lim)
                        break;
                    const uint32_t n = *(volatile LdsU32 *)spun + 1u;
                    *(volatile LdsU32 *)spun = n;
                    if (n > (uint32_t)kDitherSpinLimit) {  // never on a healthy run: give up and say so, do not hang
                        if (lane == 0) __hip_atomic_store(b.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            asm volatile("" ::: "memory");
        };
        // record x is complete once the producer has finished column x (x + 1 columns published); a record is
        // requested one step before it is unpacked.  (The slot is not clamped to the row: past its end the slots of the
        // following row -- or the slack behind the last one, sixel_launch.h -- are read, for lanes that are outside
        // their rows by then.)
        auto request = [&](int lim, uint32_t slot_addr, bool check = true, bool look = true) __attribute__((always_inline)) {
            if (check) wait_for(lim);
            if (kDitherAbl & 64) return;
            if (look) peek();
            const LdsU32 *slot = (const LdsU32 *)(uintptr_t)slot_addr;  // = this half's part of slot x_rec + 1
            n_lo = slot[0];
            n_hi = slot[1];
        };
        peek();
        request(in_base + 1, in_addr + 12u);  // column 0 of the row above the wave's first row: "arrived a step ago" at step 0
        if (rl == 0) {
            s1a = n_hi;
            s1b = n_lo;
        }
        request(in_base + 2, in_addr + 24u);  // (step 0 hands it to the lane moves)
        q_pt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_lo, 0xb1, 0xf, 0xf, true);

        // Source pixels: unconditional, from clamped addresses, 8 steps ahead, as inline assembly
        // with hand-placed waits.  Left to the compiler, the ring of 8 loads in flight loses its
        // count at the loop header (and at any branch around a load) and the step waits for
        // vmcnt(0): a memory round trip every step or every 8 steps, depending on the version.
        // The loads return in order, so "at most 14 younger operations outstanding" (seven steps' pixel and
        // index requests; 7 in the two-trip form, which requests pixels only) means the oldest pixel -- and the index
        // requested before it -- has arrived (the index stores in between only make the wait more conservative).
        // (inside: the caller knows that the column is inside the row for every lane -- the steady blocks below)
        auto fetch = [&](int t, auto inside_tag) -> uint32_t {
            uint32_t x;  // the column, clamped into the row (one v_med3: the compiler cannot know 0 <= W - 1)
            if constexpr (decltype(inside_tag)::value)
                x = (uint32_t)(t - 2 * rl);
            else
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x) : "v"(t - 2 * rl), "s"(W - 1));
            const uint8_t *p = src_row + (size_t)x * 4;
            uint32_t v;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));
            return v;
        };
        typedef uint32_t PixPair __attribute__((ext_vector_type(2)));
        // Steady blocks: the pixels of the block's columns from a RUNNING pointer (column t - 2 rl of the lane's row, t the
        // block's first step: + 32 bytes a block) with the step's distance as the load's immediate offset -- no column,
        // no 64-bit address arithmetic in the step.
        const uint8_t *px_run = src_row - 8 * rl;
        auto fetch_run = [&](auto off_tag) -> uint32_t {
            uint32_t v;
            asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v) : "v"(px_run), "i"(decltype(off_tag)::value));
            return v;
        };
        auto fetch2_run = [&](auto off_tag) -> PixPair {
            PixPair v;
            asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v) : "v"(px_run), "i"(decltype(off_tag)::value));
            return v;
        };
        auto fetch2 = [&](int t, auto inside_tag) -> PixPair {  // columns t - 2 * rl (even) and the next one, the pair clamped into the row
            uint32_t x;
            if constexpr (decltype(inside_tag)::value)
                x = (uint32_t)(t - 2 * rl);
            else
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x) : "v"(t - 2 * rl), "s"(W - 2));
            const uint8_t *p = src_row + (size_t)x * 4;
            PixPair v;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p));
            return v;
        };
        // One step.  Everything is computed by every lane, in range or not (the loads are
        // clamped, the tables indexed with clamped values): straight-line code the compiler can
        // schedule across the LDS reads; only the stores are predicated, and a lane outside its row
        // produces zero terms.  `lidx` receives the step's request for the palette index of its pixel (consumed
        // kDitherAhead steps later, with the wait for that step's pixel).
        // (k: the step's position in the unrolled body -- t - k is a multiple of eight)
        // (steady: a step of a block in which EVERY lane's column is inside 0 .. W - 2 and so are the columns whose pixels
        // it requests -- blocks 64 <= t <= W - 16, five sixths of an 800-column row.  Nothing of what only the ends of a
        // row need is in such a step: the column itself, the error's range select, the wrap term and the capture of the
        // row's first 3/16, the clamps of the pixel, record and progress addresses -- 12 instructions of 59.)
        // px_next: the NEXT step's pixel (landed: the wait in front of a step is for it); px_slot: the ring register the
        // step's own request goes to (one pixel a request); pair: the same for pixel pairs (odd steps)
        // next_wait: the ring wait (and index alignbytes) in front of the NEXT step, issued in this step's shadow
        auto step = [&](int t, uint32_t px_next, uint32_t &px_slot, uint32_t &lidx, auto k_tag, auto steady_tag, PixPair *pair, auto next_wait) __attribute__((always_inline)) {
            constexpr int k       = decltype(k_tag)::value;
            constexpr bool steady = decltype(steady_tag)::value;
            const int x = t - 2 * rl;
            // What can branch goes first -- the index store and the poll inside request() -- so that everything
            // from the pixel to the error terms is ONE basic block: the table reads and their uses are then
            // scheduled together (split by a branch, the byte reads came back through a v_and each, and their
            // latency covered nothing).
            // the eight indices completed by this block (see idx_q): one store, in the block's last step
            // (one trip: the eight cells of the block BEFORE this one, in this block's first step -- the same columns)
            if constexpr (kOneTrip && k == 0) {
                if (!(kDitherAbl & (8 | 512)) && stores_idx && (unsigned)idx_q < (unsigned)idx_px)
                    *reinterpret_cast<uint4 *>(s.index + idx_addr) = make_uint4(pkc[0], pkc[1], pkc[2], pkc[3]);
            }
            if constexpr (!kOneTrip && k == 7) {
                if (!(kDitherAbl & (8 | 512)) && stores_idx && (unsigned)idx_q < (unsigned)idx_px)
                    *reinterpret_cast<uint2 *>(s.index + idx_addr) = make_uint2(pk_lo, pk_hi);
            }
            // the boundary record requested a step ago (column t + 1), and the request for column t + 2
            const uint32_t q_lo = n_lo, q_hi = n_hi;
            // (the counter is looked at every other step, for two records: half the scalar work of the check ...
            // ... and read only in the step before it is looked at: a read nobody uses still has to land before its
            // register takes the next record)
            // (steady blocks end sixteen columns before the row does: no cap)
            request(steady ? lim_run + k : min(lim_run + k, lim_cap), in_addr + (uint32_t)(k + 3) * 12u, (k & 1) == 0, (k & 1) != 0);
            // last step's records, one row down (wave_shr:1 twice; a lane with no lane above it keeps `old`: lane 0 in both
            // moves -- its own reads the second time, what lane 1 has to end up with the first time).  Only w2 is on the
            // step's chain: the 3/16 of either half of the pair are in it; w1 follows in the shadow of the table reads.
            const uint32_t sb   = DownOneRow(w2p, q_pt, q_lo);
            PairI16 v = ApplyPair(v_pre, sb);  // (3/16 of column x + 1 of the row above: the record's first word as it is)
            const uint32_t wrap = x == W - 1 ? first_q3 : 0u;
            // the last pixel of a row also receives 3/16 of the row's FIRST error (its "below-left"
            // neighbour in libsixel's linear addressing)
            if constexpr (steady) {
                v = ApplyPair(v, own7);
            } else if (!kNarrow) {
                v = ApplyPair(v, wrap);
                v = ApplyPair(v, own7);
            } else {
                v = ApplyPair(v, own7);
                v = ApplyPair(v, wrap);
            }
            const uint32_t c5   = AsBits(__builtin_bit_cast(PairU16, v) >> 11);  // 5 bits per channel, biased
            const uint32_t part = __builtin_amdgcn_udot2(__builtin_bit_cast(PairU16, c5), __builtin_bit_cast(PairU16, cell_mul), 0u, false);
            // the biased cell, in both lanes of the pair.  (One trip: a second dot product carries this lane's table base in
            // its accumulator -- the partner's half is OR-ed into THAT, and the base's add leaves the chain; bit 15 of what
            // the even lanes store as "the cell" is that base: the band kernel masks it.)
            const uint32_t cell = kOneTrip ? (__builtin_amdgcn_udot2(__builtin_bit_cast(PairU16, c5), __builtin_bit_cast(PairU16, cell_mul), tab_base, false) |
                                              FromPairPartner(part))
                                           : (part | FromPairPartner(part));
            // 16 * c of 16 * err = 16 * c - 16 * p: c is the high byte of v ^ 0x8000 (zero in lanes that never spread: c16_mask)
            const PairU16 c16 = __builtin_bit_cast(PairU16, AsBits(__builtin_bit_cast(PairU16, AsBits(v) ^ px_bias) >> 4) & c16_mask);  // (two trips)
            const PairI16 cs  = v >> 8;  // c - 128 per half (one trip)
            uint32_t p_cell;  // the cell's palette colour as this lane's pair, times 1 (one trip) or 16
            uint32_t t0 = 0, t1 = 0;
            if constexpr (kOneTrip) {
                const uint32_t at = cell;  // (= the cell + this lane's table base)
                // (`at` as the LDS address itself: the kernel's LDS is all dynamic and starts at 0 -- the launch checks that
                // the kernel has no static LDS -- so no add of the array's link-time address stands between the cell and its read)
                typedef __attribute__((address_space(3))) uint8_t LdsU8;
                const LdsU8 *tab_at = (const LdsU8 *)(uintptr_t)(at + (uint32_t)kDitherLdsHead);
                typedef __attribute__((address_space(3))) int8_t LdsI8;
                const LdsI8 *tab_s = (const LdsI8 *)tab_at;  // (sign-extending reads: p - 128)
                t0 = (kDitherAbl & 16) ? at & 0xffu : (uint32_t)(int)tab_s[0];          // r / b
                t1 = (kDitherAbl & 16) ? (at >> 7) & 0xffu : (uint32_t)(int)tab_s[32768];  // g / (r: a byte the odd lane does not use, it meets a 0 multiplier)
                // (512: what a step would issue if helper waves staged the pixels into LDS and took the cells from it -- one
                // conflict-free ds_read_b32 and one ds_write_b16 instead of three row-scattered memory instructions)
                if (kDitherAbl & 512) *(volatile __attribute__((address_space(3))) uint16_t *)(uintptr_t)(slack + 1024u + 2u * (uint32_t)lane) = (uint16_t)cell;
                // (the pixel for kDitherAhead steps on is requested HERE, in the shadow of the table reads, with the
                // unpacking below: ~48 clocks of LDS latency otherwise spent in s_waitcnt)
                if constexpr (kPix2) {  // (the pair of this step and the one before it is used up: its next request)
                    if constexpr ((k & 1) != 0) {
                        if constexpr (steady)
                            *pair = fetch2_run(std::integral_constant<int, (k - 1 + kDitherAhead) * 4>());
                        else
                            *pair = fetch2(t - 1 + kDitherAhead, steady_tag);
                    }
                } else if (kDitherAbl & 512) {
                    px_slot = *(volatile LdsU32 *)(uintptr_t)(slack + 768u + 4u * (uint32_t)lane);
                } else if (!(kDitherAbl & 4)) {
                    if constexpr (steady)
                        px_slot = fetch_run(std::integral_constant<int, (k + kDitherAhead) * 4>());
                    else
                        px_slot = fetch(t + kDitherAhead, steady_tag);
                }
            } else {
                lidx    = tab8[cell];
                if constexpr (steady)
                    px_slot = fetch_run(std::integral_constant<int, (k + kDitherAhead) * 4>());
                else
                    px_slot = fetch(t + kDitherAhead, steady_tag);
                p_cell  = pal_half[lidx];
            }
            // (Until the index request left the step a pair of sched_barriers held this block behind the table reads.  Without
            // that request the step is bound by what it ISSUES -- taking the table reads out altogether buys 4 %,
            // -DTIMG_DITHER_ABL=16 -- and the barriers cost it eight hazard s_nops per two steps that the scheduler fills
            // when it may: 263 -> 256 us.  -DTIMG_DITHER_BARRIER: the old form.)
#ifdef TIMG_DITHER_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
            {  // LAST step's record and the counter behind it (all lanes, see rec_lo above; column W: a record of zeros)
                uint32_t at;  // (records are 12 bytes apart: 4-byte aligned only -- two words in one ds_write2_b32)
                if constexpr (steady)  // (the pair that hands down is inside its row; the others' dummies may overlap each other)
                    at = out_steady + (uint32_t)(k - 1) * 12u;
                else
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(at) : "v"(out_addr + (uint32_t)(k - 1) * 12u), "v"(rec_lo), "v"(rec_hi));
                LdsU32 *rec = (LdsU32 *)(uintptr_t)at;
                if (!(kDitherAbl & 32)) {
                    rec[0] = w2p;
                    rec[1] = w1p;
                }
                asm volatile("" ::: "memory");  // (data, then the counter, from one wave: the LDS keeps the order)
                // (the counter every other step, for two records: the consumer looks at it every other step too)
                if constexpr ((k & 1) != 0) {
                    int pv;
                    if constexpr (steady)
                        pv = prog_run + (k - 1);
                    else
                        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(pv) : "v"(prog_run + (k - 1)), "s"(prog_lo), "v"(prog_hi));  // (one SGPR per VALU instruction)
                    if (!(kDitherAbl & 32)) *(volatile LdsU32 *)(uintptr_t)prog_addr = (uint32_t)pv;
                }
            }
            if constexpr (kOneTrip) pkc[k >> 1] = (k & 1) ? (cell << 16) | pkc[k >> 1] : cell;
            const uint32_t sa = DownOneRow(w1p, w1p, q_hi);  // (the odd lanes' terms are all in w2)
            q_pt              = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_lo, 0xb1, 0xf, 0xf, true);  // (bound_ctrl: every lane has a source)
            next_wait();
            {  // the next step's pixel value as far as it does not depend on this step's error (see v_pre)
                const uint32_t nx_l = __builtin_amdgcn_perm(s1a, s1b, sel_u1);  // 1/16 of column x of the row above (for x + 1)
                const uint32_t nx_c = __builtin_amdgcn_perm(sa, sb, sel_u5);    // 5/16 of column x + 1
                v_pre = AsPair(__builtin_amdgcn_perm(px_next, px_next, sel_hi) ^ px_bias);
                v_pre = ApplyPair(v_pre, nx_l);
                v_pre = ApplyPair(v_pre, nx_c);
                s1a   = sa;
                s1b   = sb;
                // (pinned here: as a value with its one use in the next step, the compiler sank the five instructions
                // into that step's first block -- in front of its table reads -- wherever a step begins with a branch)
                uint32_t pin = AsBits(v_pre);
                asm volatile("" : "+v"(pin));
                v_pre = AsPair(pin);
            }
#ifdef TIMG_DITHER_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
            const PairI16 zero = {0, 0};
            uint32_t m7, m5, m3, m1;
            // trunc(err * n / 16) << 8 == (16 * err * n + (err < 0 ? 240 : 0)) & 0xff00: two channels at a time
            // (only what is ADDED as it stands needs its low bytes cleared: the records take the high bytes by v_perm)
            if constexpr (kOneTrip) {
                // err = c - p from the signed halves (the two table bytes, sign-extended, as one pair: v_perm); the lanes
                // that never spread and the odd lane's empty half multiply by 0 (|112 * 255 + 240| fits 16 bits)
                const PairI16 ps  = AsPair(__builtin_amdgcn_perm(t1, t0, 0x05040100u));
                const PairI16 e1  = cs - ps;
                const PairI16 err = steady || (unsigned)x < (unsigned)(W - 1) ? e1 : zero;
                const PairI16 sgn = AsPair(AsBits(err >> 15) & sgn_mask);
                m7 = AsBits(err * AsPair(k112) + sgn) & 0xff00ff00u;
                m5 = AsBits(err * AsPair(k80) + sgn);
                m3 = AsBits(err * AsPair(k48) + sgn);
                m1 = AsBits(err * AsPair(k16) + sgn);
            } else {
                // 16 * err = 16 * c - 16 * p (|16 * err * 7 + 240| fits 16 bits): c is the high byte of v ^ 0x8000;
                // zero in lanes that never spread (k_err, c16_mask) and outside the columns 0 .. W - 2 of the row
                const PairU16 p8  = __builtin_bit_cast(PairU16, p_cell);
                const PairI16 e0  = __builtin_bit_cast(PairI16, p8 * __builtin_bit_cast(PairU16, k_err) + c16);
                const PairI16 err = steady || (unsigned)x < (unsigned)(W - 1) ? e0 : zero;
                const PairI16 k7 = {7, 7}, k5 = {5, 5}, k3 = {3, 3};
                const PairI16 sgn = AsPair(AsBits(err >> 15) & 0x00f000f0u);
                m7 = AsBits(err * k7 + sgn) & 0xff00ff00u;
                m5 = AsBits(err * k5 + sgn);
                m3 = AsBits(err * k3 + sgn);
                m1 = AsBits(err + sgn);
            }
            if constexpr ((k & 1) == 0 && !steady) first_q3 = x == 0 ? (m3 & 0xff00ff00u) : first_q3;  // (x == 0 at t == 2 * rl: even steps only)
            // the column's record: moved down a row by the next step, stored for the next wave in that step's shadow
            w1p  = __builtin_amdgcn_perm(m5, m1, sel_w1);
            w2p  = __builtin_amdgcn_perm(m3, w1p, sel_w2);
            own7 = m7;
            // (pinned like v_pre: used by the next step only, the terms were sunk behind that step's poll branch -- and the
            // table bytes, crossing a block boundary, came back through a v_and each)
            // (not the 7/16: its two instructions find room between the next step's lane moves)
            asm volatile("" : "+v"(w1p), "+v"(w2p));
        };

        if constexpr (kPix2) {
            asm volatile("" ::: "memory");
            PixPair q0 = fetch2(0, std::false_type()), q1 = fetch2(2, std::false_type()), q2 = fetch2(4, std::false_type()),
                    q3 = fetch2(6, std::false_type());
            uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, l6 = 0, l7 = 0;
            asm volatile("s_waitcnt vmcnt(0) ; ring all" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
            v_pre = ApplyPair(AsPair(__builtin_amdgcn_perm(q0.x, q0.x, sel_hi) ^ px_bias), __builtin_amdgcn_perm(s1a, s1b, sel_u5));
            // A step reads the NEXT step's pixel (v_pre).  Even step k: that is its own pair's second half, which landed with
            // the first.  Odd step k: the first half of the NEXT pair -- see the wait below.  (l0 .. l7: the index ring of
            // the two-trip form's macros; nothing is requested into it here.)
            uint32_t no_slot = 0;
            // The wait in front of a step -- and the alignbytes that take the index that landed with it -- are issued in the
            // shadow of the step BEFORE it, behind that step's own requests: the counts are the same.
            // (only the pixel pairs are in flight: the pair an odd step k reads was requested in step k - 6, behind it the
            // pairs of steps k - 4 and k - 2 -- 2; the cell stores in between only make the wait more conservative)
#define TIMG_DITHER_WAIT_EVEN(L) (void)0
#define TIMG_DITHER_WAIT_ODD(QN, L) asm volatile("s_waitcnt vmcnt(2) ; ring %0" : "+v"(QN) : : "memory")
#define TIMG_DITHER_STEP_EVEN(k, Q, QNN, L, LN, S) /* (the step behind it is odd: it waits for the pair after this one) */ \
    step(t + k, Q.y, no_slot, L, std::integral_constant<int, k>(), S(), &Q, [&]() __attribute__((always_inline)) { TIMG_DITHER_WAIT_ODD(QNN, LN); });
#define TIMG_DITHER_STEP_ODD(k, Q, QN, L, LN, S)                                                               \
    step(t + k, QN.x, no_slot, L, std::integral_constant<int, k>(), S(), &Q, [&]() __attribute__((always_inline)) { TIMG_DITHER_WAIT_EVEN(LN); });
#define TIMG_DITHER_BLOCK(S)                        \
    {                                               \
        TIMG_DITHER_STEP_EVEN(0, q0, q1, l0, l1, S) \
        TIMG_DITHER_STEP_ODD(1, q0, q1, l1, l2, S)  \
        TIMG_DITHER_STEP_EVEN(2, q1, q2, l2, l3, S) \
        TIMG_DITHER_STEP_ODD(3, q1, q2, l3, l4, S)  \
        TIMG_DITHER_STEP_EVEN(4, q2, q3, l4, l5, S) \
        TIMG_DITHER_STEP_ODD(5, q2, q3, l5, l6, S)  \
        TIMG_DITHER_STEP_EVEN(6, q3, q0, l6, l7, S) \
        TIMG_DITHER_STEP_ODD(7, q3, q0, l7, l0, S)  \
        in_addr += 96u;                         \
        out_addr += 96u;                        \
        out_steady += out_pace;                 \
        px_run += 32;                           \
        lim_run += 8;                           \
        prog_run += 8;                          \
        idx_addr += 8u * kIdxBytes;             \
        idx_q += 8;                             \
    }
            int t = 0;
            TIMG_DITHER_WAIT_EVEN(l0);
            for (; t < t_steady0; t += 8) TIMG_DITHER_BLOCK(std::false_type)
            out_steady = hands_down ? out_addr : rec_lo;
            for (; t < t_steady1; t += 8) TIMG_DITHER_BLOCK(std::true_type)
            for (; t < steps; t += 8) TIMG_DITHER_BLOCK(std::false_type)
#undef TIMG_DITHER_BLOCK
#undef TIMG_DITHER_WAIT_EVEN
#undef TIMG_DITHER_WAIT_ODD
            asm volatile("s_waitcnt vmcnt(0) ; ring all"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4), "+v"(l5),
                           "+v"(l6), "+v"(l7)
                         :
                         : "memory");
#undef TIMG_DITHER_STEP_EVEN
#undef TIMG_DITHER_STEP_ODD
        } else {
        asm volatile("" ::: "memory");
        const std::false_type clamped;
        uint32_t p0 = fetch(0, clamped), p1 = fetch(1, clamped), p2 = fetch(2, clamped), p3 = fetch(3, clamped), p4 = fetch(4, clamped),
                 p5 = fetch(5, clamped), p6 = fetch(6, clamped), p7 = fetch(7, clamped);
        uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, l6 = 0, l7 = 0;  // (nothing in flight yet: not consumed before step 8)
        // "At most 14 younger operations" names the oldest pixel only once the ring is full: with the eight requests of
        // the prologue alone in flight the first seven waits would let their steps through with nothing arrived.  The
        // first pixels are awaited as a whole (a wave's first step waits a memory round trip either way).
        asm volatile("s_waitcnt vmcnt(0) ; ring all"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                     :
                     : "memory");
        v_pre = ApplyPair(AsPair(__builtin_amdgcn_perm(p0, p0, sel_hi) ^ px_bias), __builtin_amdgcn_perm(s1a, s1b, sel_u5));
        // A step reads the NEXT step's pixel (v_pre), requested in step k - 7: behind it the pixel requests of steps
        // k - 6 ... k - 1 -- 6 in either lookup form (only pixels are in flight: the one-trip form stores cells, the
        // two-trip form's index is the value its lookup produced).
        // (no early exit inside the unrolled body: with one the compiler loses count of the
        // loads in flight; the up to 7 extra steps find every lane out of range)
        // Two trips: the index goes into pk_lo / pk_hi -- the newest (pixel x - kDitherAhead) at the top byte -- INSIDE
        // the wait's asm statement: as a value the compiler could see between its wait and the step's own request into
        // the same variable, it was given a second register and copied, in flight, at the back edge (check_ring_isa.py
        // refused the build).
        // (PN: the pixel of the step the wait stands in front of + 1; L: the index that step's alignbytes take)
#define TIMG_DITHER_WAIT(PN, L)                                                               \
    if constexpr (kOneTrip && (kDitherAbl & 512) != 0)                                        \
        ;                                                                                     \
    else if constexpr (kOneTrip && (kDitherAbl & 128) != 0)                                   \
        ;                                                                                     \
    else if constexpr (kOneTrip) /* (pixels only in flight, as in the two-trip form; the cells are packed by the step) */ \
        asm volatile("s_waitcnt vmcnt(6) ; ring %0" : "+v"(PN) : : "memory");                 \
    else /* (the index is the value the lookup produced: only the pixels are in flight) */    \
        asm volatile("s_waitcnt vmcnt(6) ; ring %0\n\tv_alignbyte_b32 %1, %2, %1, 1\n\tv_alignbyte_b32 %2, %3, %2, 1" \
                     : "+v"(PN), "+v"(pk_lo), "+v"(pk_hi) : "v"(L) : "memory");
#define TIMG_DITHER_STEP(k, P, PN, PNN, L, LN, S) \
    step(t + k, PN, P, L, std::integral_constant<int, k>(), S(), nullptr, [&]() __attribute__((always_inline)) { TIMG_DITHER_WAIT(PNN, LN) });
#define TIMG_DITHER_BLOCK(S)                          \
    {                                                 \
        TIMG_DITHER_STEP(0, p0, p1, p2, l0, l1, S)    \
        TIMG_DITHER_STEP(1, p1, p2, p3, l1, l2, S)    \
        TIMG_DITHER_STEP(2, p2, p3, p4, l2, l3, S)    \
        TIMG_DITHER_STEP(3, p3, p4, p5, l3, l4, S)    \
        TIMG_DITHER_STEP(4, p4, p5, p6, l4, l5, S)    \
        TIMG_DITHER_STEP(5, p5, p6, p7, l5, l6, S)    \
        TIMG_DITHER_STEP(6, p6, p7, p0, l6, l7, S)    \
        TIMG_DITHER_STEP(7, p7, p0, p1, l7, l0, S)    \
        in_addr += 96u;                   \
        out_addr += 96u;                  \
        out_steady += out_pace;           \
        px_run += 32;                     \
        lim_run += 8;                     \
        prog_run += 8;                    \
        idx_addr += 8u * kIdxBytes;       \
        idx_q += 8;                       \
    }
        int t = 0;
        TIMG_DITHER_WAIT(p1, l0)
        for (; t < t_steady0; t += 8) TIMG_DITHER_BLOCK(std::false_type)
        out_steady = hands_down ? out_addr : rec_lo;
        if constexpr (!kNarrow)
            for (; t < t_steady1; t += 8) TIMG_DITHER_BLOCK(std::true_type)
        for (; t < steps; t += 8) TIMG_DITHER_BLOCK(std::false_type)
#undef TIMG_DITHER_BLOCK
#undef TIMG_DITHER_WAIT
        // the 16 requests still in flight must land before their registers are used for anything else
        asm volatile("s_waitcnt vmcnt(0) ; ring all"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(l0), "+v"(l1),
                       "+v"(l2), "+v"(l3), "+v"(l4), "+v"(l5), "+v"(l6), "+v"(l7)
                     :
                     : "memory");
#undef TIMG_DITHER_STEP
        }
    }
}

// (libtimg_hip_debug.so only -- csrc/Makefile compiles this file a second time with TIMG_SIXEL_FIRST_HIT_BUILD: the
// product library has ONE lookup rule, the pipelined one above)
#ifdef TIMG_SIXEL_FIRST_HIT_BUILD
// ---- K4, first-hit variant: libsixel's lookup cache as libsixel fills it -----------------------------
// sixel_encode looks a pixel up through a 15-bit (5:5:5) cache whose entry is the palette colour nearest to
// the FIRST pixel value that lands in the cell -- first in raster order, with the diffused errors of all
// earlier pixels already added (what src/sixel-canvas.cc:144-145 executes; oracle/sixel.c lookup_mode 0).
// The cache makes every lookup depend on every earlier pixel, so there is exactly one order: this kernel
// walks a frame serially, ONE WAVE per frame -- the lanes agree on every value, lane 0 writes, a cache miss
// searches the palette with all 64 lanes.  Pixel rows y and y + 1 live in LDS as 8-bit RGB and take the error
// terms in place, clamped after every single contribution, as libsixel's in-place diffusion does.  Two orders
// of magnitude slower than the pipelined kernel above (TIMG_HIP_SIXEL_FIRST_HIT asks for it): a validation
// mode, and the proof that the difference between the two is the lookup rule and nothing else.
// data + error * num / 16 with C's truncating division, clamped to 0..255 (libsixel's diffuse step)
__device__ __forceinline__ void FirstHitAdd(uint8_t *rows, int at, int err, int num, int lane) {
    int c = (int)rows[at] + err * num / 16;
    c     = c < 0 ? 0 : (c > 255 ? 255 : c);
    if (lane == 0) rows[at] = (uint8_t)c;
}
__device__ __forceinline__ void FirstHitLoadRow(const uint8_t *frame, const SixelGeom &g, int y, uint8_t *rows, int at,
                                                int lane) {
    // (PaddedPixel's logic on scalars: with `g.pad[alt]` indexed dynamically here, instcombine of ROCm 7.2's
    // clang dies in visitAllocaInst)
    const uint32_t pad0 = g.pad[0], pad1 = g.pad[1];
    const int pw = g.pad_pw, ph = g.pad_ph, checker = g.pad_checker, h = g.h;
    const size_t stride = g.stride;
    for (int x = lane; x < g.w; x += 64) {
        uint32_t px;
        if (y < h) px = *reinterpret_cast<const uint32_t *>(frame + (size_t)y * stride + (size_t)x * 4);
        else px = (checker && (((x / pw) + (y / ph)) & 1)) ? pad1 : pad0;
        rows[at + 3 * x]     = (uint8_t)px;
        rows[at + 3 * x + 1] = (uint8_t)(px >> 8);
        rows[at + 3 * x + 2] = (uint8_t)(px >> 16);
    }
}
__global__ void __launch_bounds__(64) DitherFirstHitKernel(SixelGeom g, SixelBatch b) {
    extern __shared__ uint32_t lds[];
    const int W = g.w, H = g.h6;
    const int f = blockIdx.x, lane = threadIdx.x;
    uint16_t *cache = reinterpret_cast<uint16_t *>(lds);        // [32768]: palette index + 1, 0 = empty
    uint32_t *pal   = lds + 16384;                              // [256]: r | g << 8 | b << 16
    uint8_t *rows   = reinterpret_cast<uint8_t *>(pal + 256);   // [2][row_bytes]
    const int row_bytes = (3 * W + 3) & ~3;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const uint8_t *frame      = b.fb + (size_t)f * g.frame_stride;
    const int ncolors         = s.meta[0];
    const bool dither         = s.meta[1] != 0;
    for (int i = lane; i < 16384; i += 64) lds[i] = 0u;
    for (int i = lane; i < 256; i += 64)
        pal[i] = i < ncolors ? (uint32_t)s.palette[i * 3] | ((uint32_t)s.palette[i * 3 + 1] << 8) |
                                   ((uint32_t)s.palette[i * 3 + 2] << 16)
                             : 0u;
    FirstHitLoadRow(frame, g, 0, rows, 0, lane);
    if (H > 1) FirstHitLoadRow(frame, g, 1, rows, row_bytes, lane);
    __syncthreads();
    for (int y = 0; y < H; ++y) {
        // byte offsets into rows[] (plain integers: a select between two LDS pointers crashed the compiler)
        const int cur = (y & 1) * row_bytes, nxt = ((y + 1) & 1) * row_bytes;
        uint8_t *idx_row = s.index + IndexRow(g, y);
        for (int x = 0; x < W; ++x) {
            const int r = rows[cur + 3 * x], gg = rows[cur + 3 * x + 1], bb = rows[cur + 3 * x + 2];
            const uint32_t cell = ((uint32_t)(r >> 3) << 10) | ((uint32_t)(gg >> 3) << 5) | (uint32_t)(bb >> 3);
            int ci = (int)__builtin_amdgcn_readfirstlane((int)cache[cell]) - 1;
            if (ci < 0) {  // first pixel in this cell: the nearest palette entry (smallest index among equals)
                uint32_t best = 0xffffffffu;
                for (int i = lane; i < ncolors; i += 64) {
                    const uint32_t p = pal[i];
                    const int dr = r - (int)(p & 255u), dg = gg - (int)((p >> 8) & 255u), db = bb - (int)((p >> 16) & 255u);
                    best = min(best, ((uint32_t)(dr * dr + dg * dg + db * db) << 8) | (uint32_t)i);
                }
                ci = (int)(~WaveMaxU32(~best) & 255u);
                if (lane == 0) cache[cell] = (uint16_t)(ci + 1);
            }
            if (lane == 0) idx_row[x] = (uint8_t)ci;
            if (dither && x < W - 1 && y < H - 1) {
                const uint32_t p = pal[ci];
                // (the "below left" neighbour of a row's first pixel is, in libsixel's linear addressing, the
                // LAST pixel of the same row)
                const int right = cur + 3 * (x + 1), below_left = x == 0 ? cur + 3 * (W - 1) : nxt + 3 * (x - 1);
                const int below = nxt + 3 * x, below_right = nxt + 3 * (x + 1);
                for (int n = 0; n < 3; ++n) {
                    const int e = (n == 0 ? r : n == 1 ? gg : bb) - (int)((p >> (8 * n)) & 255u);
                    FirstHitAdd(rows, right + n, e, 7, lane);
                    FirstHitAdd(rows, below_left + n, e, 3, lane);
                    FirstHitAdd(rows, below + n, e, 5, lane);
                    FirstHitAdd(rows, below_right + n, e, 1, lane);
                }
            }
        }
        __syncthreads();
        if (y + 2 < H) FirstHitLoadRow(frame, g, y + 2, rows, cur, lane);  // (row y is done: its buffer takes row y + 2)
        __syncthreads();
    }
}
#endif  // TIMG_SIXEL_FIRST_HIT_BUILD

// ---- K5: band encode, three kernels --------------------------------------------------------
// libsixel encodes a 6-row band as "nodes" (a colour's run of columns, gaps of < 10
// empty columns merged), sorts them by (start asc, end desc, colour asc) and packs them
// greedily into left-to-right passes separated by '$'.  Per band:
//   K5a BandNodes  (256 lanes)  entries sorted by (colour, column) through a presence bitmap
//                               (radix sort for frames too wide for that) -> nodes -> bucket
//                               sort by start column; per entry: run length, byte prefix, node id
//   K5b BandPack   (ONE wave)   the greedy packing is first-fit over the passes' pen
//                               positions in sorted node order: serial by nature, so it
//                               runs one wave per band with no LDS and every band of the
//                               batch in flight at once
//   K5c BandEmit   (256 lanes)  pass-major output order; byte size per node in O(1) from the
//                               prefix, scan, then bytes: one lane per slot for what stands in
//                               front of a node's body, one lane per ENTRY for runs and gaps
__device__ __forceinline__ char *PutUInt(char *p, uint32_t v) {
    char t[10];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = t[--n];
    return p;
}

// libsixel's sixel_put_flash for a run of `count` identical characters.
template <bool kWrite>
__device__ __forceinline__ int FlushRun(char *&p, int ch, int count) {
    int n = 0;
    while (count > 255) {  // has_gri_arg_limit
        if (kWrite) {
            *p++ = '!'; *p++ = '2'; *p++ = '5'; *p++ = '5'; *p++ = (char)ch;
        }
        n += 5;
        count -= 255;
    }
    if (count > 3) {
        if (kWrite) {
            *p++ = '!';
            p    = PutUInt(p, (uint32_t)count);
            *p++ = (char)ch;
        }
        n += 2 + NumLen((uint32_t)count);
    } else {
        if (kWrite)
            for (int i = 0; i < count; ++i) *p++ = (char)ch;
        n += count;
    }
    return n;
}

// FlushRun for 1 <= count <= 255 without a branch: the run's bytes (at most five: "!255c") left-aligned in a word pair,
// their number in n.  (The emit kernel is bound by the instructions its waves issue -- 4 800 bands x 4 waves x a few
// thousand instructions: per-lane loops over digits and characters, every lane of a wave on a different path, were
// most of them.)
__device__ __forceinline__ uint64_t RunWord(int ch, int count, int &n) {
    const uint32_t c  = (uint32_t)count, chu = (uint32_t)ch;
    const uint32_t d2 = c >= 200u ? 2u : c >= 100u ? 1u : 0u, r = c - 100u * d2, d1 = (r * 205u) >> 11, d0 = r - 10u * d1;  // r < 100
    const uint32_t nd = c >= 100u ? 3u : c >= 10u ? 2u : 1u;
    // the digits, first one in the low byte
    const uint32_t dig = nd == 3u ? (0x303030u + (d2 | d1 << 8 | d0 << 16)) : nd == 2u ? (0x3030u + (d1 | d0 << 8)) : (0x30u + d0);
    const uint64_t flash = (uint64_t)('!' | dig << 8) | (uint64_t)chu << (8u * (1u + nd));
    const bool plain = c <= 3u;
    n = plain ? (int)c : (int)(2u + nd);
    return plain ? (uint64_t)(chu * 0x010101u) : flash;
}

// the first n (<= 8) bytes of v to p (any alignment): at most four stores
__device__ __forceinline__ void StoreBytes(char *p, uint64_t v, int n) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (n >= 4) __builtin_memcpy(p, &lo, 4);
    if (n == 8) __builtin_memcpy(p + 4, &hi, 4);
    const int base     = n & 4;  // (n == 8: nothing is left)
    const uint32_t w   = base ? hi : lo;
    const int rest     = n == 8 ? 0 : n & 3;
    if (rest & 2) {
        const uint16_t h = (uint16_t)w;
        __builtin_memcpy(p + base, &h, 2);
    }
    if (rest & 1) p[base + (rest & 2)] = (char)(w >> (8 * (rest & 2)));
}

// byte count of FlushRun for a run of `count` (0: nothing): "!255c" for every full 255 beyond
// the last piece, then "!nc" or up to three plain characters
__device__ __forceinline__ int RunBytes(int count) {
    const int full = count > 255 ? (count - 1) / 255 : 0;
    const int rest = count - 255 * full;  // 1..255 (0 only for count == 0)
    return 5 * full + (rest > 3 ? 2 + NumLen((uint32_t)rest) : rest);
}

template <int kT = 256>
__device__ __forceinline__ uint32_t BlockExclusiveScan(uint32_t v, uint32_t *s_tmp, uint32_t *total) {
    // kT threads; s_tmp: kT / 64 words of LDS.  Returns the exclusive prefix of v.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t incl = WaveInclusiveAdd(v);
    __syncthreads();  // s_tmp may still be read from a previous scan
    if (lane == 63) s_tmp[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < kT / 64; ++q) {
        const uint32_t t = s_tmp[q];
        before += q < wv ? t : 0u;
        all += t;
    }
    if (total) *total = all;
    return before + incl - v;
}

// kT lanes per band: 512 for bands whose sort buffers live in LDS (two workgroups per CU by their LDS: 16 waves per
// CU instead of 8 -- the kernel is a chain of short phases between barriers, latency-bound), 256 for wide frames
template <bool kWide, int kT>
__global__ void __launch_bounds__(kT) BandNodesKernel(SixelGeom g, SixelBatch b) {
    static_assert(kT == 256 || (!kWide && kT == 512), "the wide path's radix counters are laid out for 256 lanes");
    extern __shared__ uint32_t lds[];
#ifdef TIMG_BANDS_TRACE  // (-DTIMG_BANDS_TRACE: 100 MHz ticks between the phases of ONE band, printed by its thread 0)
    long long t_phase[16];
    int n_phase = 0;
#define TIMG_PHASE() do { __syncthreads(); if (n_phase < 16) t_phase[n_phase++] = wall_clock64(); } while (0)
#else
#define TIMG_PHASE() do { } while (0)
#endif
    const int NE = g.band_ne;
    // aux: 4096 words of radix histogram (wide frames), later ONE column-indexed bucket word per column: nodes
    // starting in the column (later: their base) in the low half, the fill counter in the high half.  The sort
    // buffers live in LDS, or -- for frames too wide for that -- in this band's global scratch slots, which later
    // kernels overwrite with their outputs (ent_a is the final home of the sorted entries anyway).
    // Narrow frames are laid out for THREE workgroups per CU (BandNodesLdsWords: 52.5 KB at 800 columns; at 64.4 KB
    // -- two bucket arrays of 2048 words and a prefix per bitmap word -- two fitted).
    uint32_t *aux      = lds;
    uint32_t *ent_a, *ent_b;
    uint16_t *nfirst_u;
    // narrow frames, first phase: presence bitmap [256 colours][nws words], a 16-bit prefix per PAIR of words,
    // per-colour bases -- in the space the node phase reuses (aux, ent_b, nfirst_u)
    const int nws       = BandBitmapWords(g.w);  // odd row stride: a lane per colour walks its row conflict-free
    const int nwp       = (nws + 1) >> 1;
    uint32_t *bitmap    = lds;
    uint16_t *wprefix   = reinterpret_cast<uint16_t *>(lds + 256 * nws);
    uint32_t *cbase     = lds + 256 * nws + 128 * nwp;
    {
        const size_t fslot = ((size_t)blockIdx.y * g.bands + blockIdx.x) * NE;
        if (kWide) {
            ent_a    = b.band_ent + fslot;
            ent_b    = b.band_pi + fslot;
            nfirst_u = reinterpret_cast<uint16_t *>(b.band_rec + fslot);
        } else {
            ent_b    = lds + BandBucketWords(g.w);                // the unsorted node keys
            nfirst_u = reinterpret_cast<uint16_t *>(ent_b + NE);  // first entry of node k
            ent_a    = lds + BandNodesSharedWords(g.w, NE);       // sorted entries
        }
    }
    __shared__ uint32_t s_tmp[kT / 64];

    const int band = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const int W               = g.w;
    const uint8_t *index      = s.index;
    const int row0            = band * 6;  // (pixel x of row r: index[IndexRow(g, r) + x], 2-byte aligned for even x)
    const size_t slot         = (size_t)band * NE;

    // ---- entries, column-major: count, scan, write.  (Wide frames:) a lane takes kGroups x 4 adjacent
    // columns; the six index rows of a group arrive as six 4-byte loads (the rows of the
    // index image are padded to 4) that are issued together and kept for both passes.
    constexpr int kGroups = 1024 / kT;  // kT lanes x kGroups groups x 4 columns >= kMaxSixelWidth
    static_assert(kT * kGroups * 4 >= kMaxSixelWidth, "a lane's column groups cover the widest frame");
    const int n_groups    = (W + 3) / 4;
    const int per_g       = (n_groups + kT - 1) / kT;
    const int g0 = min(n_groups, tid * per_g), g1 = min(n_groups, g0 + per_g);
    uint32_t cw[kGroups][6];
    if constexpr (kWide) {
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (g0 + gi >= g1) {
                    cw[gi][r] = 0u;
                } else if (g.idx_shift) {  // four cells (K4, one trip) -> their palette indices
                    const uint8_t *at = index + IndexRow(g, row0 + r) + 8 * (g0 + gi);
                    const uint32_t c01 = *reinterpret_cast<const uint32_t *>(at), c23 = *reinterpret_cast<const uint32_t *>(at + 4);
                    cw[gi][r] = (uint32_t)s.lut8[c01 & 0x7fffu] | ((uint32_t)s.lut8[(c01 >> 16) & 0x7fffu] << 8) |
                                ((uint32_t)s.lut8[c23 & 0x7fffu] << 16) | ((uint32_t)s.lut8[(c23 >> 16) & 0x7fffu] << 24);
                } else {
                    cw[gi][r] = (uint32_t)*reinterpret_cast<const uint16_t *>(index + IndexRow(g, row0 + r) + 4 * (g0 + gi)) |
                                ((uint32_t)*reinterpret_cast<const uint16_t *>(index + IndexRow(g, row0 + r) + 4 * (g0 + gi) + 2) << 16);
                }
    }
    // (wide frames) visit(first, ent): a column's six rows as entries in FIXED slots, bit r of `first` set where row r is
    // the first of its colour -- no compaction into an array indexed at run time (that array lived in scratch memory)
    auto for_columns = [&](auto &&visit) {
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi) {
            if (g0 + gi >= g1) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = 4 * (g0 + gi) + q;
                if (x >= W) break;
                uint32_t c[6], ent[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) c[r] = (cw[gi][r] >> (8 * q)) & 0xffu;
                uint32_t fr = 0x3fu;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    uint32_t mask = 1u << r;
#pragma unroll
                    for (int p = r + 1; p < 6; ++p) {
                        const bool same = c[p] == c[r];
                        mask |= same ? 1u << p : 0u;
                        fr &= same ? ~(1u << p) : ~0u;
                    }
                    ent[r] = (c[r] << 22) | ((uint32_t)x << 6) | mask;
                }
                visit(fr, ent);
            }
        }
    };
    TIMG_PHASE();  // 0: index rows loaded
    int n_ent;
    if constexpr (!kWide) {
        // ---- entries sorted by (colour, column) in one pass.  The sort is stable by construction:
        // the rank of (colour c, column x) among c's entries is the number of c's columns below
        // x -- a popcount over c's row of a presence bitmap plus a per-word prefix.  No
        // per-lane chains of dependent LDS updates (which is what a counting sort over lane
        // chunks is), just atomics, one row scan per colour and one placement per entry.
        //
        // A lane takes PAIRS of columns (kT lanes x kPairs pairs cover the 1365 columns whose bands sort in LDS; up to
        // 1024 columns one pair a lane: every lane works, where groups of four left 312 of 512 idle at 800 columns).
        // A column's entries are its six rows with a "first row of its colour" flag -- fixed register slots, no
        // compaction: round 3's e6[n++] with a run-time n lived in SCRATCH memory (112 scratch instructions in this
        // kernel), and was computed twice.  Phases 2 + 4 of a band: 14.9 -> see DESIGN.md 4.3.
        constexpr int kPairs = 2;
        static_assert(kT * kPairs * 2 >= 1366, "a lane's column pairs cover the widest frame that sorts in LDS");
        const int n_pairs = (W + 1) >> 1;
        const int per_p   = (n_pairs + kT - 1) / kT;
        const int h0 = min(n_pairs, tid * per_p), h1 = min(n_pairs, h0 + per_p);
        uint32_t c16[kPairs][6];
#pragma unroll
        for (int pi2 = 0; pi2 < kPairs; ++pi2)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                c16[pi2][r] = h0 + pi2 >= h1 ? 0u
                              : g.idx_shift  ? *reinterpret_cast<const uint32_t *>(index + IndexRow(g, row0 + r) + 4 * (h0 + pi2))
                                             : *reinterpret_cast<const uint16_t *>(index + IndexRow(g, row0 + r) + 2 * (h0 + pi2));
        for (int i = tid; i < 256 * nws; i += kT) bitmap[i] = 0;  // (while the index loads are in flight)
        if (g.idx_shift) {  // the image holds cells (K4, one trip): their palette indices, a pair's twelve lookups in flight together
#pragma unroll
            for (int pi2 = 0; pi2 < kPairs; ++pi2)
                if (h0 + pi2 < h1) {  // (at 800 columns no lane has a second pair)
                    uint32_t lo[6], hi[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        lo[r] = s.lut8[c16[pi2][r] & 0x7fffu];
                        hi[r] = s.lut8[(c16[pi2][r] >> 16) & 0x7fffu];
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r) c16[pi2][r] = lo[r] | (hi[r] << 8);
                }
        }
        // the entries of a lane's (up to four) columns: ent[k][r] valid where bit r of first[k] is set
        uint32_t ent[2 * kPairs][6];
        uint32_t first[2 * kPairs];
#pragma unroll
        for (int k = 0; k < 2 * kPairs; ++k) {
            const int x      = 2 * (h0 + (k >> 1)) + (k & 1);
            const bool there = h0 + (k >> 1) < h1 && x < W;
            uint32_t c[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) c[r] = (c16[k >> 1][r] >> (8 * (k & 1))) & 0xffu;
            uint32_t fr = there ? 0x3fu : 0u;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                uint32_t mask = 1u << r;
#pragma unroll
                for (int q = r + 1; q < 6; ++q) {
                    const bool same = c[q] == c[r];
                    mask |= same ? 1u << q : 0u;
                    fr &= same ? ~(1u << q) : ~0u;  // (a later row of the same colour is not a first row)
                }
                ent[k][r] = (c[r] << 22) | ((uint32_t)x << 6) | mask;
            }
            first[k] = fr;
        }
        __syncthreads();
        TIMG_PHASE();  // 1: bitmap cleared
#pragma unroll
        for (int k = 0; k < 2 * kPairs; ++k)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (first[k] & (1u << r)) {
                    const uint32_t c = ent[k][r] >> 22, x = (ent[k][r] >> 6) & 0xffffu;
                    atomicOr(&bitmap[c * nws + (x >> 5)], 1u << (x & 31u));
                }
        __syncthreads();
        TIMG_PHASE();  // 2: presence bitmap
        uint32_t total = 0;
        if (tid < 256)
            for (int w = 0; w < nws; ++w) {  // lane = colour
                if (!(w & 1)) wprefix[tid * nwp + (w >> 1)] = (uint16_t)total;
                total += (uint32_t)__popc(bitmap[tid * nws + w]);
            }
        uint32_t n_ent_u;
        const uint32_t cb = BlockExclusiveScan<kT>(total, s_tmp, &n_ent_u);
        if (tid < 256) cbase[tid] = cb;
        n_ent = (int)n_ent_u;
        __syncthreads();
        TIMG_PHASE();  // 3: row scans + colour bases
#pragma unroll
        for (int k = 0; k < 2 * kPairs; ++k)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (first[k] & (1u << r)) {
                    const uint32_t c = ent[k][r] >> 22, x = (ent[k][r] >> 6) & 0xffffu;
                    const uint32_t w = x >> 5, at = c * nws + w;
                    // (columns of c below x: the pair's prefix, the even word of the pair for an odd one, this word's bits)
                    const uint32_t below_pair = (w & 1u) ? (uint32_t)__popc(bitmap[at - 1]) : 0u;
                    ent_a[cbase[c] + wprefix[c * nwp + (w >> 1)] + below_pair +
                          (uint32_t)__popc(bitmap[at] & ((1u << (x & 31u)) - 1u))] = ent[k][r];
                }
        __syncthreads();
    } else {
        uint32_t mine = 0;
        for_columns([&](uint32_t fr, const uint32_t *) { mine += (uint32_t)__popc(fr); });
        uint32_t n_ent_u;
        uint32_t at = BlockExclusiveScan<kT>(mine, s_tmp, &n_ent_u);
        n_ent = (int)n_ent_u;
        for_columns([&](uint32_t fr, const uint32_t *ent) {  // (a column's entries in row order of their first rows)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (fr & (1u << r)) ent_a[at + (uint32_t)__popc(fr & ((1u << r) - 1u))] = ent[r];
            at += (uint32_t)__popc(fr);
        });
        if (kWide) __threadfence_block();
        __syncthreads();

        // ---- stable LSD radix sort by colour: two 4-bit passes, a contiguous chunk per lane
        const int per_e = (n_ent + 255) / 256;
        const int e0 = min(n_ent, tid * per_e), e1 = min(n_ent, e0 + per_e);
        for (int pass = 0; pass < 2; ++pass) {
            const int shift     = 22 + 4 * pass;
            const uint32_t *src = pass ? ent_b : ent_a;
            uint32_t *dst       = pass ? ent_a : ent_b;
            for (int d = 0; d < 16; ++d) aux[d * 256 + tid] = 0;
            for (int i = e0; i < e1; ++i) aux[((src[i] >> shift) & 15u) * 256 + tid] += 1;
            if (kWide) __threadfence_block();
            __syncthreads();
            // exclusive scan over the 4096 counters in (digit, lane) order
            uint32_t sum = 0;
            for (int j = 0; j < 16; ++j) sum += aux[tid * 16 + j];
            uint32_t run = BlockExclusiveScan<kT>(sum, s_tmp, nullptr);
            for (int j = 0; j < 16; ++j) {
                const uint32_t t  = aux[tid * 16 + j];
                aux[tid * 16 + j] = run;
                run += t;
            }
            if (kWide) __threadfence_block();
            __syncthreads();
            for (int i = e0; i < e1; ++i) {
                const uint32_t e = src[i];
                const uint32_t d = (e >> shift) & 15u;
                dst[aux[d * 256 + tid]++] = e;
            }
            if (kWide) __threadfence_block();
            __syncthreads();
        }

    }
    TIMG_PHASE();  // 4: entries placed (sorted)
    const int per_e = (n_ent + kT - 1) / kT;
    const int e0 = min(n_ent, tid * per_e), e1 = min(n_ent, e0 + per_e);

    // ---- nodes: a node starts at a new colour or after a gap of >= 10 empty columns
    // A node's bytes are an RLE of its columns; a RUN is a maximal stretch of entries of one node
    // in adjacent columns with the same sixel, and between runs lie gaps of '?'.  What every run
    // and gap costs -- and so where each goes inside its node's body -- is computed here, in
    // parallel over the entries and while they are in LDS: the emit kernel then never walks a
    // node (serial per node, divergent per wave: it used to cost 37 us per band).
    // (a lane's chunk of entries: flags gathered once, as bit masks when the chunk is short
    // enough -- it always is for frames whose bands fit LDS)
    auto node_break2 = [](uint32_t p, uint32_t e) {
        return (p >> 22) != (e >> 22) || (int)((e >> 6) & 0xffffu) - (int)((p >> 6) & 0xffffu) - 1 >= 10;
    };
    auto run_break2 = [](uint32_t p, uint32_t e) {  // does a new run start at e, p being the entry before it?
        return (p >> 22) != (e >> 22) || ((e >> 6) & 0xffffu) != ((p >> 6) & 0xffffu) + 1 || ((e ^ p) & 0x3fu) != 0;
    };
    constexpr bool kMasks = !kWide;  // per_e <= kLdsEntries / kT <= 32
    uint32_t node_bits = 0, run_bits = 0, starts = 0;  // starts: node starts << 16 | run starts
    {
        uint32_t prev = e0 > 0 ? ent_a[e0 - 1] : 0u;
        for (int i = e0; i < e1; ++i) {
            const uint32_t e = ent_a[i];
            const bool ns = i == 0 || node_break2(prev, e), rs = i == 0 || run_break2(prev, e);
            if (kMasks) {
                node_bits |= (ns ? 1u : 0u) << (i - e0);
                run_bits |= (rs ? 1u : 0u) << (i - e0);
            }
            starts += (ns ? 0x10000u : 0u) + (rs ? 1u : 0u);
            prev = e;
        }
    }
    auto node_starts_at = [&](int i) {
        return kMasks ? ((node_bits >> (i - e0)) & 1u) != 0 : (i == 0 || node_break2(ent_a[i - 1], ent_a[i]));
    };
    auto run_starts_at = [&](int i) {
        return kMasks ? ((run_bits >> (i - e0)) & 1u) != 0 : (i == 0 || run_break2(ent_a[i - 1], ent_a[i]));
    };
    TIMG_PHASE();  // 5: node / run flags
    uint32_t totals;
    const uint32_t before = BlockExclusiveScan<kT>(starts, s_tmp, &totals);
    const int n_nodes = (int)(totals >> 16);
    uint16_t *run_first = reinterpret_cast<uint16_t *>(ent_b);       // [run] first entry   } both until the
    uint16_t *ent_bytes = reinterpret_cast<uint16_t *>(ent_b) + NE;  // [entry] its bytes   } node keys move in
    uint16_t *enode_g   = s.band_enode + slot;
    {
        uint32_t k = before >> 16, r = before & 0xffffu;
        for (int i = e0; i < e1; ++i) {
            if (node_starts_at(i)) nfirst_u[k++] = (uint16_t)i;
            if (run_starts_at(i)) run_first[r++] = (uint16_t)i;
            enode_g[i] = (uint16_t)(k - 1);
        }
    }
    for (int x = tid; x <= W; x += kT) aux[x] = 0;  // nodes starting in column x | fill counter << 16
    if (kWide) __threadfence_block();
    __syncthreads();
    TIMG_PHASE();  // 6: node / run firsts written
    // bytes per entry: the gap in front of a run at the run's first entry, the run at its last
    {
        uint16_t *erl_g = s.band_erl + slot;
        uint32_t r = before & 0xffffu, mine = 0;
        uint32_t prev = e0 > 0 ? ent_a[e0 - 1] : 0u, e = e0 < e1 ? ent_a[e0] : 0u;
        for (int i = e0; i < e1; ++i) {
            const uint32_t next = i + 1 < n_ent ? ent_a[i + 1] : 0u;
            if (run_starts_at(i)) ++r;
            // (inside a node gaps are shorter than 10 columns: RunBytes(gap) without its loops)
            const int gap  = node_starts_at(i) ? 0 : (int)((e >> 6) & 0xffffu) - (int)((prev >> 6) & 0xffffu) - 1;
            const bool end = i == n_ent - 1 || (i + 1 < e1 ? run_starts_at(i + 1) : run_break2(e, next));
            const int len  = end ? i - (int)run_first[r - 1] + 1 : 0;
            const int by   = (gap > 3 ? 3 : gap) + RunBytes(len);
            erl_g[i]       = (uint16_t)len;
            ent_bytes[i]   = (uint16_t)by;
            mine += (uint32_t)by;
            prev = e;
            e    = next;
        }
        uint32_t body_total;
        uint32_t at = BlockExclusiveScan<kT>(mine, s_tmp, &body_total);
        uint32_t *ep_g = s.band_ep + slot;
        for (int i = e0; i < e1; ++i) {
            ep_g[i] = at;
            at += ent_bytes[i];
        }
        if (tid == 0) s.band_cnt[band * 4 + 2] = (int)body_total;
    }
    if (kWide) __threadfence_block();
    __syncthreads();
    TIMG_PHASE();  // 7: bytes per entry + prefix
    uint32_t *key_u = ent_b;
    uint16_t *nf_g = s.band_nf + slot;
    for (int n = tid; n < n_nodes; n += kT) {
        const int first   = nfirst_u[n];
        const int last    = (n + 1 < n_nodes ? (int)nfirst_u[n + 1] : n_ent) - 1;
        nf_g[n]           = (uint16_t)first;
        const uint32_t e  = ent_a[first];
        const uint32_t sx = (e >> 6) & 0xffffu, mx = ((ent_a[last] >> 6) & 0xffffu) + 1;
        key_u[n]          = (sx << 20) | ((4095u - mx) << 8) | (e >> 22);
        atomicAdd(&aux[sx], 1u);
    }
    if (kWide) __threadfence_block();
    __syncthreads();
    TIMG_PHASE();  // 8: node keys + bucket counts
    // bucket bases: exclusive scan over the columns
    {
        const int per_x = (W + kT) / kT;
        const int c0 = min(W + 1, tid * per_x), c1 = min(W + 1, c0 + per_x);
        uint32_t sum = 0;
        for (int x = c0; x < c1; ++x) sum += aux[x];
        uint32_t run = BlockExclusiveScan<kT>(sum, s_tmp, nullptr);
        for (int x = c0; x < c1; ++x) {
            const uint32_t t = aux[x];
            aux[x]           = run;
            run += t;
        }
    }
    if (kWide) __threadfence_block();
    __syncthreads();
    TIMG_PHASE();  // 9: bucket bases
    uint32_t *nkey   = s.band_nkey + slot;
    uint16_t *nfirst = s.band_nfirst + slot;
    if constexpr (!kWide) {
        // Nodes into the buckets of their start columns as node NUMBERS in LDS (nfirst_u is free again: its last
        // reader was the loop above), ordered there and written to memory once.  (Keys and numbers used to be scattered
        // to memory in arrival order, read back per column, ordered and stored again: two dependent round trips to
        // memory at the end of every band, 3.5 of its 18 us.)
        uint16_t *order = nfirst_u;
        for (int n = tid; n < n_nodes; n += kT) {
            const uint32_t sx  = key_u[n] >> 20;
            const uint32_t was = atomicAdd(&aux[sx], 0x10000u);  // (base in the low half: < 6 * kMaxSixelWidth < 2^16)
            order[(was & 0xffffu) + (was >> 16)] = (uint16_t)n;
        }
        __syncthreads();
        TIMG_PHASE();  // 10: nodes into buckets
        // nodes starting in the same column (at most 6: one per colour of the column): by end desc, colour asc =
        // ascending key -- a sorting network over six fixed slots (absent ones carry the largest key)
        for (int x = tid; x < W; x += kT) {
            const uint32_t ax = aux[x];
            const int c = (int)(ax >> 16);
            if (c == 0) continue;
            const uint32_t base = ax & 0xffffu;
            uint32_t kk[6], nn[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                nn[j] = j < c ? order[base + j] : 0u;
                kk[j] = j < c ? key_u[nn[j]] : 0xffffffffu;
            }
            auto cx = [&](int i, int j) {
                const bool sw     = kk[i] > kk[j];
                const uint32_t ki = kk[i], ni = nn[i];
                kk[i] = sw ? kk[j] : ki;
                nn[i] = sw ? nn[j] : ni;
                kk[j] = sw ? ki : kk[j];
                nn[j] = sw ? ni : nn[j];
            };
            if (c > 1) {
                cx(0, 5); cx(1, 3); cx(2, 4);
                cx(1, 2); cx(3, 4);
                cx(0, 3); cx(2, 5);
                cx(0, 1); cx(2, 3); cx(4, 5);
                cx(1, 2); cx(3, 4);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < c) {
                    nkey[base + j]   = kk[j];
                    nfirst[base + j] = (uint16_t)nn[j];
                }
        }
    } else {
    for (int n = tid; n < n_nodes; n += kT) {
        const uint32_t key = key_u[n];
        const uint32_t sx  = key >> 20;
        const uint32_t was = atomicAdd(&aux[sx], 0x10000u);  // (base in the low half: < 6 * kMaxSixelWidth < 2^16)
        const uint32_t pos = (was & 0xffffu) + (was >> 16);
        nkey[pos]          = key;
        nfirst[pos]        = (uint16_t)n;
    }
    if (kWide) __threadfence_block();
    __syncthreads();  // (also orders the global writes above inside the workgroup)
    TIMG_PHASE();  // 10: nodes into buckets
    // nodes starting in the same column (at most 6: one per colour of the column):
    // order them by end desc, colour asc = ascending key
    for (int x = tid; x < W; x += kT) {
        const uint32_t ax = aux[x];
        const int c = (int)(ax >> 16);
        if (c < 2) continue;
        const uint32_t base = ax & 0xffffu;
        uint32_t kk[6];
        uint16_t ff[6];
        for (int j = 0; j < c && j < 6; ++j) {
            kk[j] = nkey[base + j];
            ff[j] = nfirst[base + j];
        }
        for (int i = 1; i < c && i < 6; ++i) {  // insertion sort
            const uint32_t kv = kk[i];
            const uint16_t fv = ff[i];
            int j             = i - 1;
            while (j >= 0 && kk[j] > kv) {
                kk[j + 1] = kk[j];
                ff[j + 1] = ff[j];
                --j;
            }
            kk[j + 1] = kv;
            ff[j + 1] = fv;
        }
        for (int j = 0; j < c && j < 6; ++j) {
            nkey[base + j]   = kk[j];
            nfirst[base + j] = ff[j];
        }
    }
    }
    TIMG_PHASE();  // 11: buckets ordered
    if (!kWide) {
        uint32_t *ent_g = s.band_ent + slot;
        for (int i = tid; i < n_ent; i += kT) ent_g[i] = ent_a[i];
    }
    TIMG_PHASE();  // 12: sorted entries out
    if (tid == 0) {
        s.band_cnt[band * 4 + 0] = n_ent;
        s.band_cnt[band * 4 + 1] = n_nodes;
    }
#ifdef TIMG_BANDS_TRACE
    if (tid == 0 && band == 30 && f == 0) {
        printf("bands: n_ent %d n_nodes %d ticks(100MHz) between phases:", n_ent, n_nodes);
        for (int i = 1; i < n_phase; ++i) printf(" %d:%lld", i, t_phase[i] - t_phase[i - 1]);
        printf(" total %lld\n", t_phase[n_phase - 1] - t_phase[0]);
    }
#endif
#undef TIMG_PHASE
}

// K5b: one wave per band.  Pass p's pen position lives in lane p % 64 of register p / 64
// (a fresh pass has pen 0, so first-fit opens passes in order by itself); 256 passes are
// always enough: the passes needed = the largest number of nodes crossing one column <=
// the number of colours.
// v[lane idx] = val for wave-uniform val and idx
__device__ __forceinline__ void WriteLane(uint32_t &v, uint32_t val, int idx) {
    // (gfx9: one SGPR operand per VALU instruction, so the lane select travels in M0 -- which nothing else of these
    // kernels uses: clobbered, not saved and restored, two instructions instead of four in a loop priced per instruction)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(idx) : "m0");
}

__global__ void __launch_bounds__(256) BandPackKernel(SixelGeom g, SixelBatch b, int n_frames) {
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);  // band of the batch
    if (unit >= g.bands * n_frames) return;                // whole wave
    const int f = unit / g.bands, band = unit % g.bands;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const size_t slot         = (size_t)band * g.band_ne;
    const uint32_t *nkey      = s.band_nkey + slot;
    uint32_t *pi              = s.band_pi + slot;
    uint16_t *xs              = s.band_xs + slot;
    const int n_nodes         = __builtin_amdgcn_readfirstlane(s.band_cnt[band * 4 + 1]);  // (wave-uniform: in a scalar register)

    // Pass p lives in lane p % 64 of register p / 64 as ONE word, pen position << 16 | nodes put so far (a pass holds
    // non-overlapping nodes: fewer than 4096): "does the node fit" is one compare against start << 16 | 0xffff, and a
    // node costs one v_readlane and one v_writelane of state.  (A wave alone prices its instructions at ~4 clocks
    // each and a TAKEN branch at 20-80, scratch/ubench/wave_latency.hip: the common case -- the node fits one of the
    // first 64 passes -- is the fall-through path, and two nodes share a trip round the loop.)
    uint32_t pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0;
    uint32_t key_next = lane < n_nodes ? nkey[lane] : 0u;
    for (int i0 = 0; i0 < n_nodes; i0 += 64) {
        const uint32_t key = key_next;
        key_next = (i0 + 64 + lane < n_nodes) ? nkey[i0 + 64 + lane] : 0u;
        uint32_t res_pi = 0, res_xs = 0;
        const int m = min(64, n_nodes - i0);
        auto place = [&](int j) __attribute__((always_inline)) {
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
            const uint32_t sx = kj >> 20, mx = 4095u - ((kj >> 8) & 0xfffu);
            const uint32_t limit = (sx << 16) | 0xffffu;
            unsigned long long fit = __ballot(pc0 <= limit);
            uint32_t old, pass;
            if (__builtin_expect(fit != 0, 1)) {
                const int p = __ffsll((long long)fit) - 1;
                old         = (uint32_t)__builtin_amdgcn_readlane((int)pc0, p);
                WriteLane(pc0, (mx << 16) | ((old & 0xffffu) + 1u), p);
                pass        = (uint32_t)p;
            } else if ((fit = __ballot(pc1 <= limit)) != 0) {
                const int p = __ffsll((long long)fit) - 1;
                old         = (uint32_t)__builtin_amdgcn_readlane((int)pc1, p);
                WriteLane(pc1, (mx << 16) | ((old & 0xffffu) + 1u), p);
                pass        = 64u + (uint32_t)p;
            } else if ((fit = __ballot(pc2 <= limit)) != 0) {
                const int p = __ffsll((long long)fit) - 1;
                old         = (uint32_t)__builtin_amdgcn_readlane((int)pc2, p);
                WriteLane(pc2, (mx << 16) | ((old & 0xffffu) + 1u), p);
                pass        = 128u + (uint32_t)p;
            } else {
                fit         = __ballot(pc3 <= limit);
                const int p = fit ? __ffsll((long long)fit) - 1 : 63;
                old         = (uint32_t)__builtin_amdgcn_readlane((int)pc3, p);
                WriteLane(pc3, (mx << 16) | ((old & 0xffffu) + 1u), p);
                pass        = 192u + (uint32_t)p;
            }
            WriteLane(res_pi, (pass << 16) | (old & 0xffffu), j);
            WriteLane(res_xs, old >> 16, j);
        };
        int j = 0;
        for (; j + 1 < m; j += 2) {
            place(j);
            place(j + 1);
        }
        if (j < m) place(j);
        if (i0 + lane < n_nodes) {
            pi[i0 + lane] = res_pi;
            xs[i0 + lane] = (uint16_t)res_xs;
        }
    }
    const uint32_t cnt0 = pc0 & 0xffffu, cnt1 = pc1 & 0xffffu, cnt2 = pc2 & 0xffffu, cnt3 = pc3 & 0xffffu;
    // output slot of the first node of every pass: exclusive scan of the 256 counts
    __shared__ uint32_t s_pbase[4][256];
    uint32_t *pb  = s_pbase[threadIdx.x >> 6];
    uint32_t c[4] = {cnt0, cnt1, cnt2, cnt3};
    uint32_t carry = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t incl = WaveInclusiveAdd(c[q]);
        pb[q * 64 + lane]   = carry + incl - c[q];
        carry += ReadLane(incl, 63);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's pi/xs stores and pb writes
    // one record per OUTPUT SLOT (passes back to back, nodes of a pass in packing order): the
    // emit kernel then reads its nodes with coalesced loads instead of chasing indices
    const uint16_t *nfirst = s.band_nfirst + slot;
    uint2 *rec             = s.band_rec + slot;
    for (int i = lane; i < n_nodes; i += 64) {
        const uint32_t v    = pi[i];
        const uint32_t pass = v >> 16, idx = v & 0xffffu;
        const uint32_t cr   = (idx == 0 && pass != 0) ? 1u : 0u;  // first node of a later pass: '$'
        rec[pb[pass] + idx] = make_uint2(nkey[i], (uint32_t)nfirst[i] | ((uint32_t)xs[i] << 15) | (cr << 27));
    }
}

__global__ void __launch_bounds__(256) BandEmitKernel(SixelGeom g, SixelBatch b) {
    extern __shared__ uint32_t lds[];
    uint32_t *node_off = lds;  // per output slot: size, then offset
    __shared__ uint32_t s_scan[5];
    __shared__ int s_overflow;

    const int band = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    const size_t slot         = (size_t)band * g.band_ne;
    const uint32_t *ent       = s.band_ent + slot;
    const uint2 *rec          = s.band_rec + slot;
    const uint32_t *ep        = s.band_ep + slot;
    const uint16_t *erl       = s.band_erl + slot;
    const uint16_t *enode     = s.band_enode + slot;
    const uint16_t *nf        = s.band_nf + slot;
    uint32_t *node_base       = s.band_pi + slot;  // (free again after the packing kernel)
    const int n_ent           = s.band_cnt[band * 4 + 0];
    const int n_nodes         = s.band_cnt[band * 4 + 1];
    const uint32_t body_total = (uint32_t)s.band_cnt[band * 4 + 2];
    if (tid == 0) s_overflow = 0;
    __syncthreads();
#ifdef TIMG_BANDS_TRACE
    long long t_phase[8];
    int n_phase = 0;
#define TIMG_PHASE() do { __syncthreads(); if (n_phase < 8) t_phase[n_phase++] = wall_clock64(); } while (0)
#else
#define TIMG_PHASE() do { } while (0)
#endif
    TIMG_PHASE();

    // A slot's node: what stands in front of its body ('$', "#c", the gap from the pen position)
    // and where the body's bytes lie in the band-wide prefix the node kernel left behind.
    struct Slot {
        int color, sx, x_start, node;
        bool tag, cr;
        uint32_t body0, body;  // prefix at the node's first entry, bytes of its runs and inner gaps
    };
    auto describe = [&](int k) {
        if (n_nodes <= 0) return Slot{};  // (nothing to describe: the records are not there)
        const uint2 r      = rec[k];
        const uint32_t key = r.x;
        Slot d;
        d.node    = (int)(r.y & 0x7fffu);
        d.x_start = (int)((r.y >> 15) & 0xfffu);
        d.cr      = ((r.y >> 27) & 1u) != 0;
        d.color   = (int)(key & 0xffu);
        d.sx      = (int)(key >> 20);
        // "#c" only when the active colour changes (first node: always, fixed up later)
        d.tag = k == 0 || (int)(rec[k - 1].x & 0xffu) != d.color;
        const int first = nf[d.node], next = d.node + 1 < n_nodes ? (int)nf[d.node + 1] : n_ent;
        d.body0 = ep[first];
        d.body  = (next < n_ent ? ep[next] : body_total) - d.body0;
        return d;
    };
    // phase 1: byte size of every output slot -- no walk over the node, see BandNodes.  (A slot's description is three
    // dependent levels of memory loads: the first two slots of a lane -- all of them for bands of up to 512 nodes --
    // are described once, both at a time, and kept for phase 2a.)
    auto slot_bytes = [&](const Slot &d) {
        return (d.cr ? 1u : 0u) + (d.tag ? 1u + (uint32_t)NumLen((uint32_t)d.color) : 0u) +
               (uint32_t)RunBytes(d.sx - d.x_start) + d.body;
    };
    const Slot d0 = describe(min(tid, max(n_nodes - 1, 0))), d1 = describe(min(tid + 256, max(n_nodes - 1, 0)));
    if (tid < n_nodes) node_off[tid] = slot_bytes(d0);
    if (tid + 256 < n_nodes) node_off[tid + 256] = slot_bytes(d1);
    for (int k = tid + 512; k < n_nodes; k += 256) node_off[k] = slot_bytes(describe(k));
    __syncthreads();
    TIMG_PHASE();  // 1: slot sizes
    // exclusive scan (256 contiguous chunks)
    const int lead = band > 0 ? 1 : 0;  // '-' (DECGNL) in front of every band but the first
    {
        const int per   = (n_nodes + 255) / 256;
        const int begin = min(tid * per, n_nodes), end = min(begin + per, n_nodes);
        uint32_t sum    = 0;
        for (int k = begin; k < end; ++k) sum += node_off[k];
        const uint32_t incl = WaveInclusiveAdd(sum);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int q = 0; q < wv; ++q) before += s_scan[q];
        uint32_t run = before + incl - sum + (uint32_t)lead;
        for (int k = begin; k < end; ++k) {
            const uint32_t sz = node_off[k];
            node_off[k]       = run;
            run += sz;
        }
        if (tid == 255) s_scan[4] = before + incl + (uint32_t)lead;
        __syncthreads();
    }
    TIMG_PHASE();  // 2: scan
    const uint32_t band_len = s_scan[4];
    char *out_band = s.band_bytes + (size_t)band * g.band_cap;
    if (band_len > g.band_cap) {
        if (tid == 0) s_overflow = 1;
    } else {
        if (tid == 0 && lead) out_band[0] = '-';
        // phase 2a, per slot: what stands in front of the body; and where the body goes, as an
        // offset to add to an entry's prefix
        auto front = [&](const Slot &d, int k) {
            char *p = out_band + node_off[k];
            if (d.cr) *p++ = '$';
            if (d.tag) {
                *p++ = '#';
                p    = PutUInt(p, (uint32_t)d.color);
            }
            if (d.sx > d.x_start) FlushRun<true>(p, '?', d.sx - d.x_start);
            node_base[d.node] = (uint32_t)(p - out_band) - d.body0;
        };
        if (tid < n_nodes) front(d0, tid);
        if (tid + 256 < n_nodes) front(d1, tid + 256);
        for (int k = tid + 512; k < n_nodes; k += 256) front(describe(k), k);
        __threadfence_block();
        __syncthreads();
        TIMG_PHASE();  // 3: what stands in front of the bodies
        // phase 2b, per ENTRY: the gap in front of a run is written by the run's first entry, the
        // run by its last
        // (four entries a lane at a time: their loads -- two dependent levels -- in flight together.  One entry per
        // trip was a chain of memory round trips, ten to twenty a band: the stores of a trip may alias the next
        // trip's loads for all the compiler knows, so it kept them in order -- 22 of the kernel's 34 us a band.)
        constexpr int kBatch = 4;
        for (int i0 = tid; i0 < n_ent; i0 += 256 * kBatch) {
            uint32_t e_[kBatch], pe_[kBatch], ep_[kBatch], nb_[kBatch];
            int node_[kBatch], pnode_[kBatch], len_[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int i  = min(i0 + 256 * j, n_ent - 1);  // (past the end: the last entry again, not emitted)
                e_[j]        = ent[i];
                node_[j]     = enode[i];
                len_[j]      = erl[i];
                ep_[j]       = ep[i];
                pe_[j]       = ent[max(i - 1, 0)];
                pnode_[j]    = enode[max(i - 1, 0)];
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) nb_[j] = node_base[node_[j]];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int i = i0 + 256 * j;
                if (i >= n_ent) continue;
                const uint32_t e  = e_[j];
                const int len     = len_[j];
                const bool inside = i > 0 && pnode_[j] == node_[j];
                const int gap     = inside ? (int)((e >> 6) & 0xffffu) - (int)((pe_[j] >> 6) & 0xffffu) - 1 : 0;
                if (gap <= 0 && len == 0) continue;
                char *p = out_band + (uint32_t)(nb_[j] + ep_[j]);  // (the sum wraps in 32 bits by design)
                if (__builtin_expect(len > 255, 0)) {  // (a run longer than a repeat count holds: libsixel cuts it into "!255c" pieces)
                    if (gap > 0) FlushRun<true>(p, '?', gap);
                    FlushRun<true>(p, (int)(e & 0x3fu) + '?', len);
                    continue;
                }
                // the gap in front of the run (inside a node: fewer than ten columns) and the run: at most 3 + 5 bytes
                int n_gap = 0, n_run = 0;
                const uint64_t v_gap = RunWord('?', gap > 0 ? gap : 1, n_gap);
                const uint64_t v_run = RunWord((int)(e & 0x3fu) + '?', len > 0 ? len : 1, n_run);
                if (gap <= 0) n_gap = 0;
                if (len <= 0) n_run = 0;
                StoreBytes(p, (n_gap ? v_gap & ((1ull << (8 * n_gap)) - 1ull) : 0ull) | (n_run ? v_run << (8 * n_gap) : 0ull),
                           n_gap + n_run);
            }
        }
    }
    __syncthreads();
    TIMG_PHASE();  // 4: runs and gaps
#ifdef TIMG_BANDS_TRACE
    if (tid == 0 && band == 30 && f == 0) {
        printf("emit: n_ent %d n_nodes %d ticks(100MHz) between phases:", n_ent, n_nodes);
        for (int i = 1; i < n_phase; ++i) printf(" %d:%lld", i, t_phase[i] - t_phase[i - 1]);
        printf(" total %lld\n", t_phase[n_phase - 1] - t_phase[0]);
    }
#endif
#undef TIMG_PHASE
    if (tid == 0) {
        const int first_color = n_nodes ? (int)(rec[0].x & 0xffu) : -1;
        const int last_color  = n_nodes ? (int)(rec[n_nodes - 1].x & 0xffu) : -1;
        // an overflowing band reports a length no caller buffer can hold
        s.band_meta[band * 4 + 0] = s_overflow ? 0x3fffffff : (int)band_len;
        s.band_meta[band * 4 + 1] = first_color;
        s.band_meta[band * 4 + 2] = last_color;
        s.band_meta[band * 4 + 3] = first_color >= 0 ? 1 + NumLen((uint32_t)first_color) : 0;
    }
}

// ---- K6 -------------------------------------------------------------------------------
// digits of v (< 100 000) straight into out[at ...] (bounded by cap); returns their number.  No character array: a
// `char tmp[]` filled through a moving pointer lives in SCRATCH memory, and a thread that formats a header through
// it is a chain of scratch round trips (20 us for one thread of a workgroup, whatever the rest of the chip does).
__device__ __forceinline__ uint32_t StoreUInt(char *out, size_t cap, uint32_t at, uint32_t v) {
    const uint32_t n      = (uint32_t)NumLen(v);
    const uint32_t dig[5] = {v / 10000u % 10u, v / 1000u % 10u, v / 100u % 10u, v / 10u % 10u, v % 10u};
    uint32_t pos          = at;
#pragma unroll
    for (int j = 0; j < 5; ++j)
        if ((uint32_t)(5 - j) <= n) {
            if (pos < cap) out[pos] = (char)('0' + dig[j]);
            ++pos;
        }
    return n;
}
__device__ __forceinline__ uint32_t StoreChar(char *out, size_t cap, uint32_t at, char c) {
    if (at < cap) out[at] = c;
    return 1;
}

// One kernel, one workgroup per band (round 4: it was two -- a workgroup per frame that computed where everything
// goes, 21 us of dependent global round trips by 64 workgroups, then the copy).  Where a band goes is a prefix over the
// lengths in front of it: <= 256 palette entries and the frame's bands, a few hundred loads that EVERY band's
// workgroup issues for itself (L2-resident, in flight together) and scans in LDS; band 0's workgroup also writes the
// header, the palette and the tail.
__global__ void __launch_bounds__(256) AssembleBandsKernel(SixelGeom g, SixelBatch b) {
    const int band            = blockIdx.x;
    const int f               = blockIdx.y;
    const int tid             = threadIdx.x;
    const SixelFrameScratch s = FrameScratch(b, g, f);
    char *out                 = b.out + (size_t)f * b.out_cap;
    __shared__ uint32_t s_tmp[5], s_at, s_elide;
    const int ncolors = s.meta[0];
    // cursor mode string + DCS + raster attributes: every workgroup needs their length, band 0's writes them
    const char *cursor   = g.broken_cursor ? "\033[80l\033[?7730l\033[?8452h" : "\033[80h\033[?7730h\033[?8452l";
    constexpr uint32_t kCursorLen = 21;  // (both strings)
    const uint32_t n_hdr = kCursorLen + 8u + (uint32_t)NumLen((uint32_t)g.w) + 1u + (uint32_t)NumLen((uint32_t)g.h6);
    if (band == 0 && tid < (int)kCursorLen + 8 && (size_t)tid < b.out_cap) {  // (a lane per fixed character)
        const char dcs[8] = {'\033', 'P', 'q', '"', '1', ';', '1', ';'};
        out[tid]          = tid < (int)kCursorLen ? cursor[tid] : dcs[tid - (int)kCursorLen];
    }
    if (band == 0 && tid == 0) {
        uint32_t at = kCursorLen + 8u;
        at += StoreUInt(out, b.out_cap, at, (uint32_t)g.w);
        at += StoreChar(out, b.out_cap, at, ';');
        at += StoreUInt(out, b.out_cap, at, (uint32_t)g.h6);
    }
    // the palette's bytes
    // (their total comes with the palette, from the median cut: only band 0's workgroup needs every entry's place)
    const uint32_t bands0 = n_hdr + (uint32_t)s.meta[2];
    const uint8_t *rgb    = s.palette + tid * 3;
    uint32_t pal_at       = 0;
    if (band == 0) pal_at = BlockExclusiveScan(tid < ncolors ? (uint32_t)PaletteEntryLen(tid, rgb) : 0u, s_tmp, nullptr);
    if (band == 0 && tid < ncolors) {
        uint32_t at = n_hdr + pal_at;
        at += StoreChar(out, b.out_cap, at, '#');
        at += StoreUInt(out, b.out_cap, at, (uint32_t)tid);
        at += StoreChar(out, b.out_cap, at, ';');
        at += StoreChar(out, b.out_cap, at, '2');
        at += StoreChar(out, b.out_cap, at, ';');
        at += StoreUInt(out, b.out_cap, at, (rgb[0] * 100u + 127u) / 255u);
        at += StoreChar(out, b.out_cap, at, ';');
        at += StoreUInt(out, b.out_cap, at, (rgb[1] * 100u + 127u) / 255u);
        at += StoreChar(out, b.out_cap, at, ';');
        at += StoreUInt(out, b.out_cap, at, (rgb[2] * 100u + 127u) / 255u);
    }
    // the bands in front of this one ('#c' at the start of a band is elided when the previous band ended in the same colour)
    {
        auto elided = [&](int k) -> uint32_t {
            const int *m = s.band_meta + k * 4;
            return k > 0 && m[1] >= 0 && m[1] == s.band_meta[(k - 1) * 4 + 2] ? (uint32_t)m[3] : 0u;
        };
        const int per = (g.bands + 255) / 256;
        const int b0 = min(g.bands, tid * per), b1 = min(g.bands, b0 + per);
        uint32_t mine = 0;
        for (int k = b0; k < b1; ++k) mine += (uint32_t)s.band_meta[k * 4 + 0] - elided(k);
        uint32_t bands_total;
        uint32_t at = bands0 + BlockExclusiveScan(mine, s_tmp, &bands_total);
        if (band >= b0 && band < b1) {  // the thread whose range holds this band
            for (int k = b0; k < band; ++k) at += (uint32_t)s.band_meta[k * 4 + 0] - elided(k);
            s_at    = at;
            s_elide = elided(band);
        }
        if (band == 0 && tid == 0) {
            // ST + cursor suffix
            const uint32_t end = bands0 + bands_total;
            StoreChar(out, b.out_cap, end, '\033');
            StoreChar(out, b.out_cap, end + 1, '\\');
            StoreChar(out, b.out_cap, end + 2, g.broken_cursor ? '\n' : '\r');
            b.out_len[f] = (unsigned long long)end + 3ull;
            if (b.len_host) {  // (pinned host words: visible to the host once the stream has passed this kernel)
                b.len_host[f] = (unsigned long long)end + 3ull;
                if (f == 0) *b.err_host = (unsigned long long)(unsigned)*b.error;
            }
        }
    }
    __syncthreads();
    const char *src      = s.band_bytes + (size_t)band * g.band_cap;
    const uint32_t len   = (uint32_t)s.band_meta[band * 4 + 0];
    const uint32_t at    = s_at;
    const uint32_t elide = s_elide;
    const uint32_t lead  = band > 0 ? 1u : 0u;  // the elided tag sits right after '-'
    if (lead && tid == 0 && (size_t)at < b.out_cap) out[at] = src[0];
    // the rest is one shifted copy: 16 bytes per lane (both ends unaligned), the last bytes one by one
    const uint32_t from = lead + elide, n = len > from ? len - from : 0u;
    const char *sp      = src + from;
    char *dp            = out + at + lead;
    const size_t room   = (size_t)at + lead < b.out_cap ? b.out_cap - ((size_t)at + lead) : 0;
    const uint32_t m    = (uint32_t)(n < room ? n : room);
    for (uint32_t i = tid * 16u; i + 16u <= m; i += 256u * 16u) {
        uint32_t v[4];
        __builtin_memcpy(v, sp + i, 16);
        __builtin_memcpy(dp + i, v, 16);
    }
    for (uint32_t i = (m & ~15u) + tid; i < m; i += 256u) dp[i] = sp[i];
}

}  // namespace
}  // namespace timg_amd

using namespace timg_amd;

static size_t Round6(int h) { return (size_t)((h + 5) - (h + 5) % 6); }

#ifndef TIMG_SIXEL_FIRST_HIT_BUILD
extern "C" size_t timg_hip_sixel_max_bytes(int w, int h) {
    return 1024 + (size_t)w * Round6(h) * 5;  // src/sixel-canvas.cc:123
}
#endif

// The encoder behind timg_hip_sixel_encode and timg_hip_scale_sixel_encode.  pieces_req > 0: the batch is cut into
// that many pieces whose kernel chains run on the context's side streams (forked from / joined to `stream` with
// events); before_piece, when given, is called with (piece, first frame, frames, stream) before a piece's first
// kernel is enqueued -- the fused entry point launches the piece's SCALE there, so that it runs beside the serial
// stages (median cut, diffusion: one workgroup per frame) of the pieces in front of it.  hook_ms (optional): device
// time of the hooks' work, summed over the pieces (HIP events on the pieces' own streams).
// (the second compilation of this file, for libtimg_hip_debug.so, is the same encoder with DitherFirstHitKernel in
// the place of BuildLut + Dither, under its own names: nothing of it is reachable through libtimg_hip.so)
#ifdef TIMG_SIXEL_FIRST_HIT_BUILD
#define TIMG_SIXEL_IMPL SixelEncodeFirstHitImpl
#else
#define TIMG_SIXEL_IMPL SixelEncodeImpl
#endif
namespace timg_amd {
int TIMG_SIXEL_IMPL(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride, size_t frame_stride,
                    int fb_on_device, int n_frames, int flags, const timg_hip_blend *pad_blend, char *out,
                    size_t out_cap, int out_on_device, size_t *out_len, void *stream, int pieces_req,
                    const std::function<hipError_t(int, int, int, hipStream_t)> *before_piece, float *hook_ms,
                    timg_hip_sixel_job *job) {
    if (!ctx || !fb || !out || (!out_len && !job) || w <= 0 || h <= 0 || n_frames <= 0)
        return TIMG_HIP_ERR_ARG;
    if (flags & ~TIMG_HIP_SIXEL_BROKEN_CURSOR) return ctx->Fail(TIMG_HIP_ERR_ARG, "unknown sixel flags 0x%x", flags);
    if (w > kMaxSixelWidth)
        return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "sixel width %d > %d", w, kMaxSixelWidth);
    if (stride == 0) stride = w * 4;
    if (stride < w * 4 || (stride & 3) || ((uintptr_t)fb & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    if (frame_stride == 0) frame_stride = (size_t)stride * h;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::lock_guard<std::mutex> lock(ctx->mu);

    SixelGeom g;
    memset(&g, 0, sizeof(g));
    g.w            = w;
    g.h            = h;
    g.h6           = (int)Round6(h);
    g.bands        = g.h6 / 6;
    g.stride       = (size_t)stride;
    g.frame_stride = frame_stride;
    g.broken_cursor = (flags & TIMG_HIP_SIXEL_BROKEN_CURSOR) != 0;
    // Pad rows: a transparent-black pixel blended over the background is the
    // background colour itself (sqrt(c*c) == c), opaque; without a usable
    // background it stays 0,0,0,0 (src/sixel-canvas.cc:111-118).
    g.pad[0] = g.pad[1] = 0u;
    g.pad_pw = g.pad_ph = 1;
    if (pad_blend && pad_blend->enabled && (pad_blend->bg >> 24) != 0) {
        g.pad[0] = (pad_blend->bg & 0x00ffffffu) | 0xff000000u;
        g.pad[1] = g.pad[0];
        const uint32_t pat = pad_blend->pattern;
        if (!((pat >> 24) == 0 || pat == pad_blend->bg || pad_blend->pattern_w <= 0 ||
              pad_blend->pattern_h <= 0)) {
            g.pad_checker = 1;
            g.pad[1]      = (pat & 0x00ffffffu) | 0xff000000u;
            g.pad_pw      = pad_blend->pattern_w;
            g.pad_ph      = pad_blend->pattern_h;
        }
    }
    // libsixel's sparse sampling (quality "low": 18383 samples budget)
    const uint32_t npix = (uint32_t)w * (uint32_t)g.h6;
    uint32_t sp         = npix / 18383u;
    if (npix < 18383u) sp = 6;
    if (sp == 0) sp = 1;
    g.sample_stride_px = sp;
    g.n_samples        = (npix + sp - 1) / sp;
    if (g.n_samples > (uint32_t)(kHistPerThread * kHistThreads))
        return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "sixel: %u samples", g.n_samples);
    g.band_cap         = (size_t)w * 6 * 16 + 1024;  // >= 16 bytes per (column, colour) entry

    const size_t fb_bytes = frame_stride * (size_t)(n_frames - 1) + (size_t)stride * h;
    const uint8_t *dfb    = fb;
    if (!fb_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(fb_bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, fb, fb_bytes, hipMemcpyHostToDevice, st));
        dfb = (const uint8_t *)ctx->dev[0].ptr;
    }
    char *dout = out;
    if (!out_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[1].Reserve(out_cap * (size_t)n_frames));
        dout = (char *)ctx->dev[1].ptr;
    }
    // one scratch allocation, carved up
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nf = (size_t)n_frames;
    size_t off      = 0;
    auto carve      = [&](size_t bytes) {
        const size_t at = off;
        off             = align(off + bytes);
        return at;
    };
    const size_t o_ent   = carve(nf * g.n_samples * 4);
    const size_t o_ta    = carve(nf * 32768 * 4);
    const size_t o_tb    = carve(nf * 32768 * 4);
    const size_t o_lut   = carve(nf * 32768 * 4);
    const size_t o_lut8  = carve(nf * 32768);
    const size_t o_pal   = carve(nf * 768);
    const size_t o_meta  = carve(nf * 4 * sizeof(int));
    // (the launch plan first: the diffusion's lookup form decides what the index image holds)
    const char *waves_env = getenv("TIMG_HIP_DITHER_WAVES"), *parts_env = getenv("TIMG_HIP_DITHER_PARTS"),
               *trips_env = getenv("TIMG_HIP_DITHER_TRIPS");
    const SixelLaunch plan = PlanSixelLaunch(w, g.h6, n_frames, ctx->cu_count, waves_env ? std::max(1, atoi(waves_env)) : 0,
                                             parts_env ? std::max(0, atoi(parts_env)) : -1, trips_env ? atoi(trips_env) : 0);
#ifdef TIMG_SIXEL_FIRST_HIT_BUILD
    g.idx_shift          = 0;  // (the serial checker writes indices)
#else
    g.idx_shift          = plan.one_trip ? 1 : 0;
#endif
    g.idx_stride         = ((w + 6 + 7) & ~7) << g.idx_shift;
    const size_t o_idx   = carve(nf * (size_t)g.h6 * g.idx_stride);
    const size_t o_bb    = carve(nf * g.bands * g.band_cap);
    const size_t o_bm    = carve(nf * g.bands * 4 * sizeof(int));
    const size_t o_bo    = carve(nf * g.bands * 2 * sizeof(uint32_t));
    g.band_ne            = BandEntries(w);
    const size_t n_band  = nf * g.bands;
    const size_t o_bent  = carve(n_band * g.band_ne * 4);
    const size_t o_bkey  = carve(n_band * g.band_ne * 4);
    const size_t o_bfst  = carve(n_band * g.band_ne * 2);
    const size_t o_bpi   = carve(n_band * g.band_ne * 4);
    const size_t o_bxs   = carve(n_band * g.band_ne * 2);
    const size_t o_brec  = carve(n_band * g.band_ne * sizeof(uint2));
    const size_t o_bep   = carve(n_band * g.band_ne * 4);
    const size_t o_berl  = carve(n_band * g.band_ne * 2);
    const size_t o_ben   = carve(n_band * g.band_ne * 2);
    const size_t o_bnf   = carve(n_band * g.band_ne * 2);
    const size_t o_bcnt  = carve(n_band * 4 * sizeof(int));
    const size_t o_prow  = carve(nf * (size_t)w * 5 * sizeof(uint32_t));
    const size_t o_xwg   = carve(nf * (kDitherMaxParts - 1) * (size_t)XwgStride(w) * sizeof(uint32_t));
    const size_t o_len   = carve((nf + 1) * sizeof(unsigned long long));  // + 1: device error word
    // An earlier ASYNCHRONOUS call may still be running on this scratch (context.h: sixel_done).  On the same stream the
    // kernels below queue behind it; on another stream they are made to (on the device: nobody blocks); and before the
    // scratch GROWS -- Reserve frees the old block -- the host waits for that call, whatever stream it is on.
    // (The event is the JOB's: the call in flight recorded it behind its last kernel, and a job that is destroyed waits
    // for its call and takes itself out of the context first -- one event record a call instead of two: 5 us of a step.)
    if (ctx->sixel_in_flight && ctx->sixel_last_job) {
        if (off > ctx->dev[5].bytes) {
            TIMG_HIP_TRY(ctx, hipEventSynchronize(ctx->sixel_last_job->done));
            ctx->sixel_in_flight = false;
        } else if (ctx->sixel_stream != st) {
            TIMG_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->sixel_last_job->done, 0));
        }
    }
    TIMG_HIP_TRY(ctx, ctx->dev[5].Reserve(off));
    char *base = (char *)ctx->dev[5].ptr;
    SixelBatch b;
    b.fb         = dfb;
    b.entries    = (uint32_t *)(base + o_ent);
    b.tab_a      = (uint32_t *)(base + o_ta);
    b.tab_b      = (uint32_t *)(base + o_tb);
    b.lut        = (uint32_t *)(base + o_lut);
    b.lut8       = (uint8_t *)(base + o_lut8);
    b.palette    = (uint8_t *)(base + o_pal);
    b.meta       = (int *)(base + o_meta);
    b.index      = (uint8_t *)(base + o_idx);
    b.band_bytes = base + o_bb;
    b.band_meta  = (int *)(base + o_bm);
    b.band_off   = (uint32_t *)(base + o_bo);
    b.band_ent    = (uint32_t *)(base + o_bent);
    b.band_nkey   = (uint32_t *)(base + o_bkey);
    b.band_nfirst = (uint16_t *)(base + o_bfst);
    b.band_pi     = (uint32_t *)(base + o_bpi);
    b.band_xs     = (uint16_t *)(base + o_bxs);
    b.band_rec    = (uint2 *)(base + o_brec);
    b.band_ep     = (uint32_t *)(base + o_bep);
    b.band_erl    = (uint16_t *)(base + o_berl);
    b.band_enode  = (uint16_t *)(base + o_ben);
    b.band_nf     = (uint16_t *)(base + o_bnf);
    b.band_cnt    = (int *)(base + o_bcnt);
    b.pad_rows    = (uint32_t *)(base + o_prow);
    b.xwg         = (uint32_t *)(base + o_xwg);
    b.error       = (int *)(base + o_len + nf * sizeof(unsigned long long));
    b.out        = dout;
    b.out_cap    = out_cap;
    b.out_len    = (unsigned long long *)(base + o_len);

    // which kernels, how many waves / workgroups per frame, how much LDS: sixel_launch.h.  (One workgroup per frame
    // for the diffusion: small frames, and the fallback of the multi-CU placement.  Round 1's two-CU version, whose
    // DIFFUSING waves read the bridge in global memory themselves, was not faster; DitherKernel<., true> keeps the
    // memory round trips in helper waves.)
    const int dither_waves = plan.dither_waves, dither_parts = plan.dither_parts, split_share = plan.split_share;
    const size_t dither_lds = plan.dither_lds, split_lds = plan.split_lds;
    const bool wide_bands   = plan.wide_bands;  // sort buffers in global scratch
    const size_t nodes_lds = plan.nodes_lds, emit_lds = plan.emit_lds;
    // the diffusion kernel that serves this geometry, its grid, block and dynamic LDS
    // (pixel pairs: an even width and rows that start on 8 bytes -- the pad rows in the scratch do: its carve is aligned)
    {  // (TIMG_HIP_DITHER_HELPERS=<columns>,<naps> overrides)
        const char *he = getenv("TIMG_HIP_DITHER_HELPERS");
        int hb = 0, hn = 0;
        if (he && sscanf(he, "%d,%d", &hb, &hn) == 2 && hb >= 1 && hn >= 1) {
            g.helper_batch = hb;
            g.helper_naps  = hn;
        } else {
            // (profiles/r4/dither_helpers.txt: 64 frames x 4 parts 347 us at 4,4 -- 350 at 8,8, 356 at 16,16, 366 at 32,32;
            // one frame x 16 parts: the step 0.63 ms at 2,2 against 0.64 at 1,1.  Round 6, with a step a quarter shorter
            // (profiles/r6/dither_helpers.txt): 283 us at 4,4 -- 278 at 8,4 / 8,8 / 16,8, 280-282 at 1,1 / 2,2)
            g.helper_batch = dither_parts <= 6 ? 8 : 2;
            g.helper_naps  = dither_parts <= 6 ? 4 : 2;
        }
    }
    const char *pix_env = getenv("TIMG_HIP_DITHER_PIX");
    const bool pix2     = (w & 1) == 0 && w >= 4 && (g.stride & 7) == 0 && (g.frame_stride & 7) == 0 &&
                      ((uintptr_t)b.fb & 7) == 0 && ((uintptr_t)b.pad_rows & 7) == 0 && !(pix_env && atoi(pix_env) == 1);
    const void *dither_fn = nullptr;
    dim3 dither_block;
    size_t dither_dyn = 0;
    if (dither_parts > 1) {
        dither_fn    = plan.one_trip ? (pix2 ? (const void *)DitherKernel<false, true, true, true> : (const void *)DitherKernel<false, true, true>)
                                     : (const void *)DitherKernel<false, true, false>;
        dither_block = dim3((split_share + 2) * 64);
        dither_dyn   = split_lds;
    } else {
        dither_fn = w <= 2 ? (const void *)DitherKernel<true, false, false>
                  : plan.one_trip ? (pix2 ? (const void *)DitherKernel<false, false, true, true> : (const void *)DitherKernel<false, false, true>)
                                  : (const void *)DitherKernel<false, false, false>;
        dither_block = dim3(dither_waves * 64);
        dither_dyn   = dither_lds;
    }
    // the kernels below need more than the default 64 KiB of dynamic LDS
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute(dither_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dither_dyn));
    {  // (the one-trip diffusion addresses its tables from LDS address 0: no static LDS may stand in front of the dynamic block)
        hipFuncAttributes fa;
        TIMG_HIP_TRY(ctx, hipFuncGetAttributes(&fa, dither_fn));
        if (fa.sharedSizeBytes != 0)
            return ctx->Fail(TIMG_HIP_ERR_DEVICE, "the sixel diffusion kernel has %zu bytes of static LDS", (size_t)fa.sharedSizeBytes);
    }
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute(wide_bands ? (const void *)BandNodesKernel<true, 256>
                                                     : (const void *)BandNodesKernel<false, kBandLanes>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)nodes_lds));
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)BandEmitKernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds));
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)HistKernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHistLdsBytes));
#ifdef TIMG_SIXEL_FIRST_HIT_BUILD
    const size_t first_hit_lds = (16384 + 256) * sizeof(uint32_t) + 2 * (size_t)((3 * w + 3) & ~3);
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)DitherFirstHitKernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)first_hit_lds));
#endif
    TIMG_HIP_TRY(ctx, hipFuncSetAttribute((const void *)MedianCutKernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kCutLdsBytes));
    // The palette (median cut) and diffusion kernels are serial per frame: one workgroup
    // per frame, microseconds of dependent steps, a few dozen CUs busy.  The batch can be
    // cut into groups whose kernel chains run on side streams (one group's median cut next
    // to another group's diffusion).  MEASURED on MI355X / ROCm 7.2: every extra group costs
    // ~1.8 ms of cross-queue event hand-over per 64-frame batch (1 group 2.8 ms, 2 groups
    // 4.7 ms, 4 groups 6.3 ms), far more than the overlap wins -- so the default is ONE
    // group; TIMG_HIP_SIXEL_GROUPS keeps the experiment reproducible.
    int n_groups = pieces_req > 0 ? pieces_req : 1;
    if (pieces_req <= 0)
        if (const char *e = getenv("TIMG_HIP_SIXEL_GROUPS"))  // tuning
            n_groups = atoi(e);
    n_groups = std::max(1, std::min(n_groups, std::min(n_frames, (int)timg_hip_ctx::kSideStreams)));
    // the words the byte counts and the error word end up in: the job's, or the context's (both pinned)
    unsigned long long *len_h = nullptr;
    if (job) {
        len_h = job->len_h;
    } else {
        TIMG_HIP_TRY(ctx, ctx->pin[0].Reserve(sizeof(unsigned long long) * (nf + 1)));
        len_h = (unsigned long long *)ctx->pin[0].ptr;
    }
    const bool direct = n_groups == 1 && !getenv("TIMG_HIP_SIXEL_COPY_LENGTHS");  // (the switch: the old form, for comparison)
    b.clears_error    = direct ? 1 : 0;
    b.len_host        = direct ? len_h : nullptr;
    b.err_host        = direct ? len_h + nf : nullptr;
    if (direct)
        len_h[nf] = 0;  // (nothing of an earlier call is in flight towards these words: a job is not pending, a blocking call has returned)
    else
        TIMG_HIP_TRY(ctx, hipMemsetAsync(b.error, 0, sizeof(unsigned long long), st));
    if (n_groups > 1 || hook_ms) TIMG_HIP_TRY(ctx, ctx->EnsureSideStreams());
    if (n_groups > 1) TIMG_HIP_TRY(ctx, hipEventRecord(ctx->fork_event, st));
    for (int grp = 0; grp < n_groups; ++grp) {
        const int f0 = (int)((long long)n_frames * grp / n_groups);
        const int f1 = (int)((long long)n_frames * (grp + 1) / n_groups);
        const int nfr = f1 - f0;
        if (nfr <= 0) continue;
        hipStream_t gs = n_groups > 1 ? ctx->side[grp] : st;
        if (n_groups > 1) TIMG_HIP_TRY(ctx, hipStreamWaitEvent(gs, ctx->fork_event, 0));
        // the group's view of the batch: every per-frame array starts at frame f0
        SixelBatch gb = b;
        const size_t o = (size_t)f0;
        gb.fb          = b.fb + o * g.frame_stride;
        gb.entries     = b.entries + o * g.n_samples;
        gb.tab_a       = b.tab_a + o * 32768;
        gb.tab_b       = b.tab_b + o * 32768;
        gb.lut         = b.lut + o * 32768;
        gb.lut8        = b.lut8 + o * 32768;
        gb.palette     = b.palette + o * 768;
        gb.meta        = b.meta + o * 4;
        gb.index       = b.index + o * g.h6 * g.idx_stride;
        gb.band_bytes  = b.band_bytes + o * g.bands * g.band_cap;
        gb.band_meta   = b.band_meta + o * g.bands * 4;
        gb.band_off    = b.band_off + o * g.bands * 2;
        gb.band_ent    = b.band_ent + o * g.bands * g.band_ne;
        gb.band_nkey   = b.band_nkey + o * g.bands * g.band_ne;
        gb.band_nfirst = b.band_nfirst + o * g.bands * g.band_ne;
        gb.band_pi     = b.band_pi + o * g.bands * g.band_ne;
        gb.band_xs     = b.band_xs + o * g.bands * g.band_ne;
        gb.band_rec    = b.band_rec + o * g.bands * g.band_ne;
        gb.band_ep     = b.band_ep + o * g.bands * g.band_ne;
        gb.band_erl    = b.band_erl + o * g.bands * g.band_ne;
        gb.band_enode  = b.band_enode + o * g.bands * g.band_ne;
        gb.band_nf     = b.band_nf + o * g.bands * g.band_ne;
        gb.band_cnt    = b.band_cnt + o * g.bands * 4;
        gb.pad_rows    = b.pad_rows + o * w * 5;
        gb.xwg         = b.xwg + o * (kDitherMaxParts - 1) * XwgStride(w);
        gb.out         = b.out + o * b.out_cap;
        gb.out_len     = b.out_len + o;
        gb.len_host    = b.len_host ? b.len_host + o : nullptr;

        if (before_piece) {
            if (hook_ms) TIMG_HIP_TRY(ctx, hipEventRecord(ctx->hook_event[2 * grp], gs));
            TIMG_HIP_TRY(ctx, (*before_piece)(grp, f0, nfr, gs));
            if (hook_ms) TIMG_HIP_TRY(ctx, hipEventRecord(ctx->hook_event[2 * grp + 1], gs));
        }
        hipLaunchKernelGGL(HistKernel, dim3(nfr), dim3(kHistThreads), kHistLdsBytes, gs, g, gb);
        hipLaunchKernelGGL(MedianCutKernel, dim3(nfr), dim3(kCutWaves * 64), kCutLdsBytes, gs, g, gb);
#ifdef TIMG_SIXEL_FIRST_HIT_BUILD
        hipLaunchKernelGGL(DitherFirstHitKernel, dim3(nfr), dim3(64), first_hit_lds, gs, g, gb);
        (void)dither_fn; (void)dither_block; (void)dither_dyn; (void)dither_parts;
#else
        {
            hipLaunchKernelGGL(BuildLutKernel, dim3(512 / kLutWaves, nfr), dim3(kLutWaves * 64), 0, gs, g, gb);
            {
                SixelGeom kg       = g;
                SixelBatch kb      = gb;
                void *kargs[]      = {&kg, &kb};
                const dim3 grid    = dither_parts > 1 ? dim3(dither_parts, nfr) : dim3(nfr);
                TIMG_HIP_TRY(ctx, hipLaunchKernel(dither_fn, grid, dither_block, kargs, dither_dyn, gs));
            }
        }
#endif
        if (wide_bands)
            hipLaunchKernelGGL((BandNodesKernel<true, 256>), dim3(g.bands, nfr), dim3(256), nodes_lds, gs, g, gb);
        else
            hipLaunchKernelGGL((BandNodesKernel<false, kBandLanes>), dim3(g.bands, nfr), dim3(kBandLanes), nodes_lds, gs, g, gb);
        hipLaunchKernelGGL(BandPackKernel, dim3((g.bands * nfr + 3) / 4), dim3(256), 0, gs, g, gb, nfr);
        hipLaunchKernelGGL(BandEmitKernel, dim3(g.bands, nfr), dim3(256), emit_lds, gs, g, gb);
        hipLaunchKernelGGL(AssembleBandsKernel, dim3(g.bands, nfr), dim3(256), 0, gs, g, gb);
        if (n_groups > 1) {
            TIMG_HIP_TRY(ctx, hipEventRecord(ctx->join_event[grp], gs));
            TIMG_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->join_event[grp], 0));
        }
    }
    TIMG_HIP_TRY(ctx, hipGetLastError());

    if (job) {
        // the asynchronous form: the frames' byte counts (and the device's error word) go to the JOB's pinned words
        // behind an event on the caller's stream; nothing is waited for -- the caller may enqueue its next batch (the
        // scratch is reused in stream order; the counts above were copied out before the next call's kernels run)
        if (!direct)
            TIMG_HIP_TRY(ctx, hipMemcpyAsync(job->len_h, b.out_len, sizeof(unsigned long long) * (nf + 1), hipMemcpyDeviceToHost, st));
        TIMG_HIP_TRY(ctx, hipEventRecord(job->done, st));
        ctx->sixel_last_job  = job;
        ctx->sixel_stream    = st;
        ctx->sixel_in_flight = true;
        job->n       = n_frames;
        job->out_cap = out_cap;
        job->pending = true;
        return TIMG_HIP_OK;
    }

    if (!direct)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(len_h, b.out_len, sizeof(unsigned long long) * (nf + 1), hipMemcpyDeviceToHost, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    ctx->sixel_in_flight = false;  // (this call waited behind whatever was in flight, and has finished itself)
    ctx->sixel_last_job  = nullptr;
    if ((int)len_h[nf] != 0)
        return ctx->Fail(TIMG_HIP_ERR_DEVICE, "sixel diffusion: a workgroup gave up waiting for its neighbour");
    if (hook_ms) {
        *hook_ms = 0.0f;
        for (int grp = 0; grp < n_groups && before_piece; ++grp) {
            float ms = 0.0f;
            if ((int)((long long)n_frames * (grp + 1) / n_groups) - (int)((long long)n_frames * grp / n_groups) <= 0) continue;
            TIMG_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->hook_event[2 * grp], ctx->hook_event[2 * grp + 1]));
            *hook_ms += ms;
        }
    }
    size_t worst = 0;
    for (int i = 0; i < n_frames; ++i) {
        out_len[i] = (size_t)len_h[i];
        if (out_len[i] > worst) worst = out_len[i];
    }
    if (worst > out_cap)
        return ctx->Fail(TIMG_HIP_ERR_SMALL, "frame needs %zu bytes, out_cap is %zu", worst, out_cap);
    if (!out_on_device) return CopyFramesToHost(ctx, out, out_cap, dout, out_len, n_frames, st);
    return TIMG_HIP_OK;
}
}  // namespace timg_amd

#ifndef TIMG_SIXEL_FIRST_HIT_BUILD
extern "C" int timg_hip_sixel_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h,
                                     int stride, size_t frame_stride, int fb_on_device,
                                     int n_frames, int flags, const timg_hip_blend *pad_blend,
                                     char *out, size_t out_cap, int out_on_device,
                                     size_t *out_len, void *stream) {
    return timg_amd::SixelEncodeImpl(ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, pad_blend, out,
                                     out_cap, out_on_device, out_len, stream, 0, nullptr, nullptr, nullptr);
}

// ---- the asynchronous form (round 5): a step of a pipeline must not end with a blocking read-back ----
extern "C" int timg_hip_sixel_job_create(timg_hip_ctx *ctx, int max_frames, timg_hip_sixel_job **out) {
    if (!ctx || !out || max_frames <= 0) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timg_hip_sixel_job *j = new timg_hip_sixel_job();
    j->ctx        = ctx;
    j->max_frames = max_frames;
    hipError_t e  = hipHostMalloc((void **)&j->len_h, sizeof(unsigned long long) * ((size_t)max_frames + 1), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (j->len_h) (void)hipHostFree(j->len_h);
        delete j;
        return ctx->FailHip(e, "timg_hip_sixel_job_create");
    }
    *out = j;
    return TIMG_HIP_OK;
}
extern "C" void timg_hip_sixel_job_destroy(timg_hip_sixel_job *j) {
    if (!j) return;
    if (j->pending) (void)hipEventSynchronize(j->done);  // (the kernels' writes into len_h must not land in freed memory)
    if (j->ctx) {  // (the context orders later calls behind this job's event: not behind a destroyed one)
        std::lock_guard<std::mutex> lock(j->ctx->mu);
        if (j->ctx->sixel_last_job == j) {
            if (!j->pending && j->done) (void)hipEventSynchronize(j->done);  // (waited for already: returns at once)
            j->ctx->sixel_last_job  = nullptr;
            j->ctx->sixel_in_flight = false;
        }
    }
    if (j->done) (void)hipEventDestroy(j->done);
    if (j->len_h) (void)hipHostFree(j->len_h);
    delete j;
}
extern "C" int timg_hip_sixel_encode_async(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                                           size_t frame_stride, int n_frames, int flags,
                                           const timg_hip_blend *pad_blend, char *out, size_t out_cap, void *stream,
                                           timg_hip_sixel_job *job) {
    if (!ctx || !job || job->ctx != ctx) return TIMG_HIP_ERR_ARG;
    if (job->pending) return ctx->Fail(TIMG_HIP_ERR_ARG, "timg_hip_sixel_encode_async: the job still holds a call that was not waited for");
    if (n_frames > job->max_frames) return ctx->Fail(TIMG_HIP_ERR_ARG, "timg_hip_sixel_encode_async: %d frames, the job was created for %d", n_frames, job->max_frames);
    return timg_amd::SixelEncodeImpl(ctx, fb, w, h, stride, frame_stride, 1, n_frames, flags, pad_blend, out, out_cap, 1,
                                     nullptr, stream, 0, nullptr, nullptr, job);
}
extern "C" int timg_hip_sixel_encode_wait(timg_hip_sixel_job *job, size_t *out_len) {
    if (!job || !out_len) return TIMG_HIP_ERR_ARG;
    timg_hip_ctx *ctx = job->ctx;
    if (!job->pending) return ctx->Fail(TIMG_HIP_ERR_ARG, "timg_hip_sixel_encode_wait: nothing to wait for");
    TIMG_HIP_TRY(ctx, hipEventSynchronize(job->done));
    job->pending = false;
    if ((int)job->len_h[job->n] != 0)
        return ctx->Fail(TIMG_HIP_ERR_DEVICE, "sixel diffusion: a workgroup gave up waiting for its neighbour");
    size_t worst = 0;
    for (int i = 0; i < job->n; ++i) {
        out_len[i] = (size_t)job->len_h[i];
        if (out_len[i] > worst) worst = out_len[i];
    }
    if (worst > job->out_cap)
        return ctx->Fail(TIMG_HIP_ERR_SMALL, "frame needs %zu bytes, out_cap is %zu", worst, job->out_cap);
    return TIMG_HIP_OK;
}
#else
// TEST-ONLY (libtimg_hip_debug.so): timg_hip_sixel_encode with libsixel's own lookup rule -- a 15-bit cell answers with
// the palette entry nearest to the FIRST pixel value that lands in it, in raster order, diffused errors included (what
// sixel_encode executes; oracle/sixel.c lookup_mode 0).  That order is serial: one wave walks a frame, ~0.3 s per
// 800x450 frame.  It exists so that the product's rule can be compared with libsixel's semantics pixel by pixel on
// the device (the per-pixel bound of DESIGN.md 2); ctx is a context of libtimg_hip.so.
extern "C" int timg_hip_debug_sixel_encode_first_hit(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                                                     size_t frame_stride, int fb_on_device, int n_frames, int flags,
                                                     const timg_hip_blend *pad_blend, char *out, size_t out_cap,
                                                     int out_on_device, size_t *out_len, void *stream) {
    return timg_amd::SixelEncodeFirstHitImpl(ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, pad_blend,
                                             out, out_cap, out_on_device, out_len, stream, 0, nullptr, nullptr, nullptr);
}
#endif
