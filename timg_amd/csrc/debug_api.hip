// timg_amd/csrc/debug_api.hip -- TEST-ONLY library (libtimg_hip_debug.so): host-only
// introspection used by the CPU test suite (no GPU needed).  Not part of the public ABI
// (not declared in include/timg_hip.h) and not linked into libtimg_hip.so.
#include <cstring>

#include <algorithm>
#include <vector>

#include "cu_mask.h"
#include "gfx_layout.h"
#include "h2_strips.h"
#include "resample_plan.h"
#include "sixel_launch.h"

extern "C" int timg_hip_debug_plan_dump(int sw, int sh, int in_fmt, int dw, int dh, int filter,
                                        int *header, int *h_taps, float *h_coeff,
                                        int h_coeff_cap, int *v_cnt, int *v_rows,
                                        float *v_coeff, int v_cap) {
    timg_amd::ResamplePlan p;
    if (!timg_amd::BuildResamplePlan(sw, sh, in_fmt, dw, dh, filter, &p)) return -1;
    header[0] = p.vertical_first;
    header[1] = p.identity;
    header[2] = p.h_sequential;
    header[3] = p.h_width;
    header[4] = p.max_active_rows;
    header[5] = p.max_v_count;
    header[6] = p.max_h_count;
    header[7] = p.v_is_gather;
    if ((long)dw * p.h_width > h_coeff_cap) return -1;
    if ((long)p.v_rows.size() > v_cap) return -1;
    for (int x = 0; x < dw; ++x) {
        h_taps[2 * x]     = p.h_taps[x].n0;
        h_taps[2 * x + 1] = p.h_taps[x].count;
    }
    memcpy(h_coeff, p.h_coeff.data(), p.h_coeff.size() * sizeof(float));
    for (int y = 0; y < dh; ++y) v_cnt[y] = p.v_runs[y].count;
    memcpy(v_rows, p.v_rows.data(), p.v_rows.size() * sizeof(int));
    memcpy(v_coeff, p.v_coeff.data(), p.v_coeff.size() * sizeof(float));
    return (int)p.v_rows.size();
}

// The graphics-protocol path (gfx_canvas.hip) with plain loops in place of the kernels' lanes:
// the same per-index functions (gfx_layout.h), the same order of passes.  Host only, for
// tests/test_gfx_layout.py.  kind: 0 png, 1 kitty, 2 iTerm2; flags bit 0: RGB (alpha dropped), bit 1: the
// group-wise forms of the body and base64 passes (what the kernels run) instead of the element-wise ones.
extern "C" long timg_hip_debug_gfx_emulate(int kind, const uint8_t *fb, int w, int h, int flags, uint32_t image_id,
                                           uint8_t *out, long cap) {
    using namespace timg_amd;
    const PngGeom g = MakePngGeom(w, h, !(flags & 1));
    std::vector<uint8_t> png_store(g.png_n);
    uint8_t *png = kind == kGfxPng ? out : png_store.data();
    if (kind == kGfxPng && cap < (long)g.png_n) return -1;
    // pass 1: head, filtered bytes, block headers, the two sums
    for (uint32_t i = 0; i < kPngIdatData + 2; ++i) png[i] = g.head[i];
    unsigned long long sum_a = 0, sum_b = 0;
    const bool groupwise = (flags & 2) != 0;  // the kernels' four-pixel / four-group forms
    if (groupwise) {
        const uint32_t gpr = ((uint32_t)w + 3u) >> 2;
        for (uint32_t i = 0; i < PngBodyGroups(g); ++i) {
            const PngGroupSums r = PngBodyGroup(fb, (size_t)w * 4, g, i / gpr, i % gpr, png);
            sum_a += r.a;
            sum_b += (unsigned long long)(g.raw_n - r.j0) * r.a - (long long)r.t;
        }
    } else {
        for (uint32_t j = 0; j < g.raw_n; ++j) {
            const uint8_t v      = PngRawByte(fb, (size_t)w * 4, g, j);
            png[PngRawOffset(j)] = v;
            sum_a += v;
            sum_b += (unsigned long long)(g.raw_n - j) * v;
        }
    }
    for (uint32_t b = 0; b < g.n_blocks; ++b) PngBlockHeader(g, b, png + PngBlockHeaderOffset(b));
    // pass 2: Adler-32 closes the zlib stream
    PutBE32(png + PngAdlerOffset(g), AdlerFromSums(g, sum_a, sum_b));
    // passes 3 and 4: CRC of "IDAT" + stream from chunk CRCs
    std::vector<uint32_t> chunk(g.n_chunks), segment(g.n_segments);
    for (uint32_t c = 0; c < g.n_chunks; ++c) chunk[c] = PngChunkCrc(png, g, c);
    for (uint32_t s = 0; s < g.n_segments; ++s) {  // levels 0..9 inside a segment of 1024 chunks ...
        const uint32_t base = s * kCrcSegment, count = std::min(kCrcSegment, g.n_chunks - base);
        PngTreeReduce(chunk.data() + base, count, base, 0, kCrcSegmentLog, g);
        segment[s] = chunk[base];
    }
    PngTreeReduce(segment.data(), g.n_segments, 0, kCrcSegmentLog, kCrcLevels, g);  // ... the rest across segments
    PngTail(png, g, segment[0]);
    if (kind == kGfxPng) return (long)g.png_n;
    // pass 5: base64 and framing
    char header[kGfxHeaderCap];
    const GfxFraming f = MakeFraming(kind, g, FormatGfxHeader(kind, g, image_id, header));
    if (cap < (long)f.total) return -1;
    for (uint32_t i = 0; i < f.header_len; ++i) out[i] = (uint8_t)header[i];
    if (groupwise)
        for (uint32_t quad = 0; quad * 4u < f.n_groups; ++quad) GfxFrameQuad(png, g, f, quad, out);
    for (uint32_t grp = 0; !groupwise && grp < f.n_groups; ++grp) {
        const uint32_t q = Base64Quad(png, g.png_n, grp);
        uint8_t *o       = out + GfxGroupOffset(f, grp);
        o[0] = (uint8_t)q;
        o[1] = (uint8_t)(q >> 8);
        o[2] = (uint8_t)(q >> 16);
        o[3] = (uint8_t)(q >> 24);
    }
    if (kind == kGfxKitty)
        for (uint32_t c = 1; c < f.n_kitty_chunks; ++c) KittySeparator(f, c, out + KittySeparatorOffset(f, c));
    GfxTrailer(f, out);
    return (long)f.total;
}

// The sixel kernels' launch geometry for a frame of w x h6 (padded height) in a batch of n_frames on a device of
// cu_count CUs (sixel_launch.h; waves_cap / parts_env as TIMG_HIP_DITHER_WAVES / TIMG_HIP_DITHER_PARTS, 0 / -1: not
// set; trips: TIMG_HIP_DITHER_TRIPS, 0: chosen by the plan).  out[13]: band_ne, dither_waves, dither_lds, dither_parts,
// split_share, split_lds, wide_bands, nodes_lds, emit_lds, lds budget of a workgroup, the diffusion's static LDS
// allowance, kDitherMaxWaves, one_trip.
extern "C" void timg_hip_debug_sixel_launch(int w, int h6, int n_frames, int cu_count, int waves_cap, int parts_env,
                                            int trips, long out[13]) {
    const timg_amd::SixelLaunch L = timg_amd::PlanSixelLaunch(w, h6, n_frames, cu_count, waves_cap, parts_env, trips);
    out[0]  = L.band_ne;
    out[1]  = L.dither_waves;
    out[2]  = (long)L.dither_lds;
    out[3]  = L.dither_parts;
    out[4]  = L.split_share;
    out[5]  = (long)L.split_lds;
    out[6]  = L.wide_bands ? 1 : 0;
    out[7]  = (long)L.nodes_lds;
    out[8]  = (long)L.emit_lds;
    out[9]  = timg_amd::kSixelLdsBudget;
    out[10] = (long)timg_amd::kDitherStaticLds;
    out[11] = timg_amd::kDitherMaxWaves;
    out[12] = L.one_trip ? 1 : 0;
}

// The two-column horizontal-first kernel's tiling of a plan (h2_strips.h; what scale_stream.hip uploads): header[0..5] =
// ok, taps per lane, first step of the second column, widest window, strips, widest half window; strips[3 * n] (ox0, ox1,
// cx0), pairs[2 * 32 * n] (a, b), halves[3 * 2 * n].  Returns the number of strips, -1 when the arrays are too small
// (cap_strips) or the plan is degenerate.  The set of instantiated (taps, step) pairs is scale_stream.hip's.
extern "C" int timg_hip_debug_h2_tiling(int sw, int sh, int in_fmt, int dw, int dh, int filter, int cap_strips, int *header,
                                        int *strips, int *pairs, int *halves) {
    timg_amd::ResamplePlan p;
    if (!timg_amd::BuildResamplePlan(sw, sh, in_fmt, dw, dh, filter, &p)) return -1;
    const timg_amd::H2Tiling t = timg_amd::BuildH2Tiling(p, 64, 32, 1024, [](int taps_lane, int js) { return taps_lane == 20 && js >= 2 && js <= 5; });
    header[0] = t.ok;
    header[1] = t.taps_lane;
    header[2] = t.js;
    header[3] = t.win;
    header[4] = (int)t.strips.size();
    header[5] = t.half_win;
    if (!t.ok) return 0;
    if ((int)t.strips.size() > cap_strips) return -1;
    for (size_t i = 0; i < t.strips.size(); ++i) {
        strips[3 * i] = t.strips[i].ox0, strips[3 * i + 1] = t.strips[i].ox1, strips[3 * i + 2] = t.strips[i].cx0;
    }
    for (size_t i = 0; i < t.pairs.size(); ++i) pairs[2 * i] = t.pairs[i].a, pairs[2 * i + 1] = t.pairs[i].b;
    for (size_t i = 0; i < t.halves.size(); ++i) {
        halves[3 * i] = t.halves[i].ox0, halves[3 * i + 1] = t.halves[i].ox1, halves[3 * i + 2] = t.halves[i].cx0;
    }
    return (int)t.strips.size();
}

// the CU mask of timg_hip_stream_create (cu_mask.h): words[32]; returns the CUs of every XCD kept free, or -1
extern "C" int timg_hip_debug_cu_mask(int cu_count, int reserved_cus_per_xcd, uint32_t *words) {
    return timg_amd::BuildCuMask(cu_count, reserved_cus_per_xcd, words);
}
