// timg_amd/csrc/sixel_launch.h -- launch geometry of the sixel kernels as a function of the frame size: which
// kernel variants run, with how many waves / workgroups per frame and how much dynamic LDS.  Shared by
// sixel_canvas.hip (which launches by it) and the test-only libtimg_hip_debug.so (tests/test_sixel_launch.py sweeps
// every width and a ladder of heights on the CPU: a geometry whose tables fill the LDS to the byte failed to launch
// once -- 766 columns, round 2 -- and was found by a random stress run on the GPU, not by a test).
#ifndef TIMG_AMD_SIXEL_LAUNCH_H_
#define TIMG_AMD_SIXEL_LAUNCH_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace timg_amd {

constexpr int kSixelLdsBudget = 160 * 1024;  // LDS of one CU = the most one workgroup can have (static + dynamic)

// A band has at most 6 * width (colour, column) entries.  Up to kLdsEntries of them the band
// kernels sort in LDS; wider frames (up to kMaxSixelWidth: columns travel in 12-bit fields)
// use the same code on global-memory scratch.
constexpr int kLdsEntries    = 8192;
constexpr int kMaxSixelWidth = 4095;
constexpr int kBandLanes     = 512;  // lanes per band of BandNodesKernel (frames whose bands sort in LDS)

// diffusion (DitherKernel)
constexpr int kDitherMaxWaves = 16;
constexpr int kPairRows       = 32;  // rows per wave
// The diffusion of a frame spread over several workgroups (CUs): hand-over buffer of one boundary between two of
// them -- the boundary row's (W + 3) x 3 words, then (in a cache line of its own) the producer's progress counter
constexpr int kDitherMaxParts = 16;
__host__ __device__ inline int XwgData(int w) { return ((w + 3) * 3 + 31) & ~31; }
__host__ __device__ inline int XwgStride(int w) { return XwgData(w) + 32; }
// Two forms of the diffusion's table lookup (DitherKernel<., ., kOneTrip>):
//  * one trip: the palette COLOUR of every 15-bit cell in LDS -- three byte tables b, r, g of 32 KB each; one LDS
//    round trip on the serial chain, the palette index comes from memory off the chain.  96 KB of tables leave room
//    for six boundary rows of 800 columns;
//  * two trips: cell -> palette index (32 KB), index -> colour (2 KB), two dependent LDS round trips; for geometries
//    whose boundary rows do not fit beside the colour tables in the placement that is wanted.
__host__ __device__ constexpr int DitherTabWords(bool one_trip) { return one_trip ? (65536 + 32768) / 4 : 8192 + 512; }
// dynamic LDS of a diffusion workgroup that writes `rows` boundary rows (+ the one that stays zero): the tables, then
// the boundary rows.  (A workgroup of ONE wave follows itself from round to round, 62 columns ahead of its own
// writes: the row that "stays zero" is its own row, freshly cleared.)
// The steps read the row above up to kDitherOverrun slots past its end (unclamped: the lanes that receive them are
// outside their rows by then): slack behind the last row.
constexpr int kDitherOverrun = 96;
constexpr int kDitherLdsHead = 64;  // bytes in front of the tables: the waves' progress counters
inline size_t DitherLdsBytes(int w, int rows, bool one_trip) {
    return kDitherLdsHead +
           (DitherTabWords(one_trip) + (size_t)(rows > 1 ? rows + 1 : 1) * 3 * (size_t)(w + 3) + 3 * kDitherOverrun) * sizeof(uint32_t);
}
// The slack behind the last boundary row (3 * kDitherOverrun words) has two tenants (DitherKernel, "handing down WITHOUT
// touching exec"): the unclamped record reads past a row's end, and every lane's DUMMY stores -- an 8-byte record at
// 8 * lane and a progress word at 512 + 4 * lane.  Both must fit, and the waves' real progress counters must fit the head:
static_assert(3 * kDitherOverrun * sizeof(uint32_t) >= 512 + 4 * 64, "the lanes' dummy record and progress stores must fit the slack");
static_assert(8 * 64 <= 512, "the dummy records (8 bytes a lane) must end where the dummy progress words begin");
static_assert(kDitherLdsHead >= 4 * kDitherMaxWaves, "one progress counter per wave in front of the tables");
// (Kept free of a workgroup's LDS budget although the kernel has had no static LDS since round 4: the placement
// decisions -- waves per workgroup, parts per frame -- at geometries that fill the LDS to the byte are the ones the
// parity tests pin (widths 766, 825, 1200...); giving the 512 bytes back would move them for nothing.)
constexpr size_t kDitherStaticLds = 512;

// LDS layout of BandNodesKernel for frames whose bands sort in LDS (words)
__host__ __device__ inline int BandBitmapWords(int w) { return ((w + 31) >> 5) | 1; }
__host__ __device__ inline int BandBucketWords(int w) { return (w + 8) & ~1; }
__host__ __device__ inline int BandNodesSharedWords(int w, int ne) {  // bitmap phase and node phase, one after the other
    const int nws = BandBitmapWords(w), nwp = (nws + 1) >> 1;
    const int a = 256 * nws + 128 * nwp + 256, b = BandBucketWords(w) + ne + ne / 2;
    return a > b ? a : b;
}
inline int BandEntries(int w) { return ((6 * w + 63) / 64) * 64; }

struct SixelLaunch {
    int band_ne;          // entries a band can have (rounded up to 64)
    bool one_trip;        // the diffusion's lookup form (above)
    int dither_waves;     // one workgroup per frame: its waves (frames with more row groups go round again) ...
    size_t dither_lds;    // ... and its dynamic LDS
    int dither_parts;     // > 1: DitherKernel<., true, .> with this many workgroups per frame ...
    int split_share;      // ... the largest part's row groups (its block has split_share + 2 waves) ...
    size_t split_lds;     // ... and its dynamic LDS
    bool wide_bands;      // the band kernels sort in global scratch
    size_t nodes_lds, emit_lds;
};

// the diffusion's placement for one lookup form
inline void PlanDither(SixelLaunch &L, int w, int h6, int n_frames, int cu_count, int waves_cap, int parts_env, bool one_trip) {
    L.one_trip = one_trip;
    // one wave per 32 rows, as many as the boundary rows leave room for next to the tables
    const int groups = (h6 + kPairRows - 1) / kPairRows;
    int waves        = groups < 1 ? 1 : (groups > kDitherMaxWaves ? kDitherMaxWaves : groups);
    if (waves_cap > 0 && waves > waves_cap) waves = waves_cap;
    while (waves > 1 && DitherLdsBytes(w, waves, one_trip) > kSixelLdsBudget - kDitherStaticLds) --waves;
    L.dither_waves = waves;
    L.dither_lds   = DitherLdsBytes(w, waves, one_trip);
    // Frames of eight row groups and more are spread over several workgroups = CUs (DitherKernel<., true, .>): about
    // four row groups a part (one wave per SIMD: 800x450 measured 644 / 587 / 557 / 522 us per 64 frames with
    // 1 / 2 / 3 / 4 parts), no more parts than the batch leaves CUs for (parts of frames that wait for a CU while
    // others spin cost more than they win), or what TIMG_HIP_DITHER_PARTS asks for; then the fewest parts from
    // there whose largest share of the row groups (+ fetcher + flusher) fits a workgroup and its LDS; 1: one
    // workgroup per frame.
    L.dither_parts = 1;
    if (w > 2 && groups >= 8 && waves_cap <= 0) {
        int by_cus = n_frames > 0 ? cu_count / n_frames : cu_count;
        if (by_cus < 1) by_cus = 1;
        int want = (groups + 3) / 4;
        if (want > kDitherMaxParts) want = kDitherMaxParts;
        if (want > by_cus) want = by_cus;
        if (parts_env >= 0) want = parts_env;
        if (want > groups) want = groups;  // (no part without a row group)
        for (int p = want < 1 ? 1 : want; p > 1 && p <= kDitherMaxParts && p <= groups; ++p) {
            const int share = (groups + p - 1) / p;
            if (share + 2 <= kDitherMaxWaves && DitherLdsBytes(w, share + 1, one_trip) <= kSixelLdsBudget - kDitherStaticLds) {
                L.dither_parts = p;
                break;
            }
        }
    }
    L.split_share = L.dither_parts > 1 ? (groups + L.dither_parts - 1) / L.dither_parts : 0;
    L.split_lds   = L.dither_parts > 1 ? DitherLdsBytes(w, L.split_share + 1, one_trip) : 0;
}

// waves_cap: TIMG_HIP_DITHER_WAVES (0: not set); parts_env: TIMG_HIP_DITHER_PARTS (< 0: not set); trips_env:
// TIMG_HIP_DITHER_TRIPS (1 / 2: that lookup form; anything else: chosen here)
inline SixelLaunch PlanSixelLaunch(int w, int h6, int n_frames, int cu_count, int waves_cap, int parts_env, int trips_env = 0) {
    SixelLaunch L{};
    L.band_ne = BandEntries(w);
    // The one-trip lookup where its tables do not cost the placement: as many parts as the two-trip form would get
    // (or, one workgroup per frame, as many waves -- a CU that diffuses a frame alone is bound by what its waves
    // issue together, and twelve waves issue more than five).  Never for the narrow kernel (w <= 2).
    SixelLaunch one = L, two = L;
    PlanDither(two, w, h6, n_frames, cu_count, waves_cap, parts_env, false);
    PlanDither(one, w, h6, n_frames, cu_count, waves_cap, parts_env, true);
    bool use_one;
    if (one.dither_parts > 1 || two.dither_parts > 1)  // (more parts than the other form needs only if all are resident at once)
        use_one = one.dither_parts > 1 && (one.dither_parts <= two.dither_parts ||
                                           (long)one.dither_parts * (n_frames > 0 ? n_frames : 1) <= cu_count);
    else
        use_one = one.dither_waves >= two.dither_waves;
    if (w <= 2) use_one = false;
    if (trips_env == 1 && w > 2) use_one = true;
    if (trips_env == 2) use_one = false;
    L = use_one ? one : two;
    L.wide_bands  = L.band_ne > kLdsEntries;
    L.nodes_lds   = L.wide_bands ? (size_t)(2 * 4096 + 16) * sizeof(uint32_t)
                                 : ((size_t)BandNodesSharedWords(w, L.band_ne) + L.band_ne + 16) * sizeof(uint32_t);
    L.emit_lds    = (size_t)L.band_ne * sizeof(uint32_t);
    return L;
}

}  // namespace timg_amd

#endif  // TIMG_AMD_SIXEL_LAUNCH_H_
