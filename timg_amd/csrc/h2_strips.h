// timg_amd/csrc/h2_strips.h -- host-side tiling for ScaleStreamH2Kernel (scale_stream.hip): which horizontal-first plans
// the kernel with TWO output columns per lane pair serves, its strips, its pair table, and the half strips the
// seven-channel fallback runs on.  Pure host code over a ResamplePlan: shared by scale_stream.hip (which uploads the
// tables) and the test-only libtimg_hip_debug.so (tests/test_h2_strips.py checks the invariants on the CPU).
//
// A lane of pair i walks one chain parity p of source pixels n0(A) + p + 2j for column A AND for column B = A + 1, whose
// window starts d = n0(B) - n0(A) pixels on: B's taps under this lane form ONE of B's chains, q = (p - d) & 1, and begin
// at step (d + q - p) / 2.  The kernel gives B the TAPS + 1 steps from JS on: a pair is REGULAR when both lanes' starts
// are JS or JS + 1.  Columns that do not pair like that (at the clamped edges all windows start at the first pixel)
// take a lane pair alone.
#ifndef TIMG_AMD_H2_STRIPS_H_
#define TIMG_AMD_H2_STRIPS_H_

#include <algorithm>
#include <map>
#include <vector>

#include "resample_plan.h"

namespace timg_amd {

struct H2Strip {
    int ox0, ox1;  // output columns [ox0, ox1)
    int cx0;       // first source column of the window (multiple of 4)
};
struct H2Pair {
    int a, b;  // columns of the lane pair; b == -1: a alone; a == -1: an idle pair
};
struct H2Tiling {
    bool ok = false;
    int taps_lane = 0;           // taps per lane and column (the kernel's TAPS)
    int js        = 0;           // first step of the second column (the kernel's JS)
    int win       = 0;           // widest window of a strip, multiple of 4
    std::vector<H2Strip> strips; // at most cols_pair / 2 pairs each
    std::vector<H2Pair> pairs;   // [strip][cols_pair / 2]
    std::vector<H2Strip> halves; // two per strip (the second may be empty), at most cols_half columns each
    int half_win  = 0;           // widest window of a half
};

// cols_pair: columns a wave of the two-column kernel carries at most (64); cols_half: of the one-column kernel (32);
// win_max: source pixels of a row buffer (1024).  instantiated(taps_lane, js): does the kernel exist for it.
template <class Instantiated>
inline H2Tiling BuildH2Tiling(const ResamplePlan &p, int cols_pair, int cols_half, int win_max, Instantiated instantiated) {
    H2Tiling t;
    const int taps_lane = p.h_width <= 16 ? 8 : p.h_width <= 40 ? 20 : 40;
    t.taps_lane         = taps_lane;
    if (p.vertical_first || p.h_sequential || p.out_w < 2) return t;
    auto pair_steps = [&](int a, int *lo, int *hi) {  // first steps of column a + 1 in the two lanes of (a, a + 1)
        const int d = p.h_taps[a + 1].n0 - p.h_taps[a].n0;
        *lo = 1 << 30;
        *hi = -(1 << 30);
        for (int par = 0; par < 2; ++par) {
            const int q = ((par - d) % 2 + 2) % 2, js = (d + q - par) / 2;
            *lo = std::min(*lo, js);
            *hi = std::max(*hi, js);
        }
    };
    int js = 0;
    {  // the most frequent first step
        std::map<int, int> votes;
        for (int a = 0; a + 1 < p.out_w; ++a) {
            int lo, hi;
            pair_steps(a, &lo, &hi);
            if (hi - lo <= 1) ++votes[lo];
        }
        int best = 0;
        for (const auto &kv : votes)
            if (kv.second > best) best = kv.second, js = kv.first;
        // (d alternates between two neighbouring values: some pairs start at js, some at js or js + 1 -- the smaller of
        // the two most frequent when they are neighbours)
        if (votes.count(js - 1) && votes[js - 1] * 4 >= best) js = js - 1;
        if (!(best > 0 && js >= 1 && js < taps_lane && instantiated(taps_lane, js))) return t;
    }
    t.js = js;
    auto regular = [&](int a) {
        if (a + 1 >= p.out_w) return false;
        int lo, hi;
        pair_steps(a, &lo, &hi);
        return lo >= js && hi <= js + 1 && p.h_taps[a].count <= 2 * taps_lane && p.h_taps[a + 1].count <= 2 * taps_lane;
    };
    int n_single = 0;
    // strips of at most cols_pair / 2 lane pairs whose windows fit the row buffer; about equally many columns in each
    const int n_even = (p.out_w + cols_pair - 1) / cols_pair;
    const int cols   = ((p.out_w + n_even - 1) / n_even + 1) & ~1;
    for (int ox = 0; ox < p.out_w;) {
        H2Strip si;
        si.ox0  = ox;
        si.cx0  = p.h_taps[ox].n0 & ~3;
        int end = ox, reach = si.cx0, used = 0;
        std::vector<H2Pair> mine;
        auto fits = [&](int c) { return p.h_taps[c].n0 >= si.cx0 && p.h_taps[c].n0 + p.h_taps[c].count <= si.cx0 + win_max; };
        while (end < p.out_w && used < cols_pair / 2 && end - ox < cols) {
            if (p.h_taps[end].count > 2 * taps_lane) return t;
            if (!fits(end)) break;
            H2Pair e = {end, -1};
            if (regular(end) && fits(end + 1) && end + 1 - ox < cols) e.b = end + 1;
            else ++n_single;
            for (int c = end; c <= (e.b >= 0 ? e.b : end); ++c) reach = std::max(reach, p.h_taps[c].n0 + p.h_taps[c].count);
            mine.push_back(e);
            end = (e.b >= 0 ? e.b : end) + 1;
            ++used;
        }
        if (end == ox) return t;
        si.ox1 = end;
        t.win  = std::max(t.win, (reach - si.cx0 + 3) & ~3);
        mine.resize((size_t)cols_pair / 2, H2Pair{-1, -1});
        t.pairs.insert(t.pairs.end(), mine.begin(), mine.end());
        t.strips.push_back(si);
        ox = end;
    }
    // (worth it only where most columns share a lane pair)
    if (n_single * 8 > p.out_w) return t;
    // the one-column kernel's strips: the two halves of every strip (the second may be empty)
    for (const H2Strip &w : t.strips) {
        const int half = std::min(cols_half, (w.ox1 - w.ox0 + 1) / 2);
        for (int k = 0; k < 2; ++k) {
            H2Strip si;
            si.ox0 = k == 0 ? w.ox0 : std::min(w.ox1, w.ox0 + half);
            si.ox1 = k == 0 ? std::min(w.ox1, w.ox0 + half) : w.ox1;
            si.cx0 = p.h_taps[std::min(si.ox0, p.out_w - 1)].n0 & ~3;
            if (si.ox1 - si.ox0 > cols_half) return t;  // (cannot happen: a strip has at most cols_pair columns)
            int reach = si.cx0;
            for (int c = si.ox0; c < si.ox1; ++c) reach = std::max(reach, p.h_taps[c].n0 + p.h_taps[c].count);
            if (reach - si.cx0 > win_max) return t;
            t.half_win = std::max(t.half_win, (reach - si.cx0 + 3) & ~3);
            t.halves.push_back(si);
        }
    }
    t.ok = true;
    return t;
}

}  // namespace timg_amd
#endif  // TIMG_AMD_H2_STRIPS_H_
