"""The host side of ScaleStreamH2Kernel (timg_amd/csrc/h2_strips.h), on the CPU.

The horizontal-first scale kernel with two output columns per lane pair (round 6) rests on a tiling the
host builds from the resample plan: strips of up to 32 lane pairs, a pair table (B = A + 1 where the two windows sit as the
kernel's steps assume, a column alone at the clamped edges), the first step JS of the second column, and the half
strips the seven-channel fallback runs on.  The kernel's bit-exactness against the reference is a GPU test
(tests/test_gpu_parity.py::test_two_column_horizontal_first_kernel_and_its_fallback, scratch/r6_h2stress.py); what can
be checked without a device is that the tiling is a tiling and that the kernel's weight slots -- restated here from
its source, lane by lane -- hand every tap of every column to exactly one lane, once, in stb's order
(stb_image_resize2.h:5801-6009: taps k, k + 2, ... of a chain accumulate in ascending order)."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS_PAIR, COLS_HALF, WIN_MAX = 64, 32, 1024


def tiling(sw, sh, dw, dh, in_fmt=0):
    L = ctypes.CDLL(os.path.join(ROOT, "timg_amd", "libtimg_hip_debug.so"))
    cap = dw // 2 + 8
    hdr = (ctypes.c_int * 6)()
    st, pr, hv = (ctypes.c_int * (3 * cap))(), (ctypes.c_int * (COLS_PAIR * cap))(), (ctypes.c_int * (6 * cap))()
    n = L.timg_hip_debug_h2_tiling(sw, sh, in_fmt, dw, dh, 0, cap, hdr, st, pr, hv)
    assert n >= 0
    ok, taps_lane, js, win, n_strips, half_win = list(hdr)
    if not ok:
        return None
    return dict(taps_lane=taps_lane, js=js, win=win, half_win=half_win,
                strips=np.array(st[:3 * n]).reshape(n, 3), pairs=np.array(pr[:COLS_PAIR * n]).reshape(n, COLS_PAIR // 2, 2),
                halves=np.array(hv[:6 * n]).reshape(2 * n, 3))


GEOMS = [(4433, 73, 767, 33), (3793, 205, 532, 53), (4634, 196, 707, 61), (3438, 138, 674, 80), (867, 436, 109, 44), (4000, 400, 800, 40),
         (7680, 4320, 800, 450), (7680, 1000, 800, 100), (7000, 400, 800, 40), (6601, 400, 777, 40), (8000, 400, 800, 40),
         (9999, 500, 1001, 50), (5003, 600, 601, 60), (8191, 300, 850, 33), (4100, 400, 455, 41), (12000, 300, 1200, 30)]


@pytest.mark.parametrize("sw,sh,dw,dh", GEOMS)
def test_tiling_covers_every_column_once_and_every_tap_once(sw, sh, dw, dh):
    info = oracle_lib.Oracle().plan_info(sw, sh, dw, dh)
    t = tiling(sw, sh, dw, dh)
    if info["vertical_first"] or not (17 <= info["h_widest"] <= 40):
        # outside the instantiated range the plan keeps the one-column kernel: the tiling must say so
        assert t is None, (info, t and t["js"])
        return
    assert t is not None, info
    plan = oracle_lib.product_plan_dump(sw, sh, dw, dh)
    n0, cnt = plan["h_taps"][0::2], plan["h_taps"][1::2]
    hw = plan["header"][3]
    coeff = plan["h_coeff"].reshape(dw, hw)  # bit patterns
    taps, js = t["taps_lane"], t["js"]
    assert taps == 20 and js in (2, 3, 4, 5)
    # -- a tiling: strips cover [0, dw) in order, pairs cover each strip's columns in order, once
    assert t["strips"][0, 0] == 0 and t["strips"][-1, 1] == dw
    assert (t["strips"][1:, 0] == t["strips"][:-1, 1]).all()
    seen = np.zeros(dw, int)
    singles = 0
    for (ox0, ox1, cx0), pairs in zip(t["strips"], t["pairs"]):
        assert cx0 % 4 == 0 and cx0 <= n0[ox0]
        cols = []
        for a, b in pairs:
            if a < 0:
                assert b < 0
                continue
            cols.append(a)
            if b >= 0:
                assert b == a + 1
                cols.append(b)
            else:
                singles += 1
        assert cols == list(range(ox0, ox1)), (ox0, ox1, cols[:6])
        seen[ox0:ox1] += 1
        # the window holds every tap of the strip, and every pixel a lane's steps reach stays inside the row buffer
        reach = max(n0[c] + cnt[c] for c in cols)
        assert reach - cx0 <= t["win"] <= WIN_MAX
    assert (seen == 1).all()
    assert singles * 8 <= dw
    # -- the halves: two per strip, each at most 32 columns, together the strip
    for k, (ox0, ox1, _) in enumerate(t["strips"]):
        h0, h1 = t["halves"][2 * k], t["halves"][2 * k + 1]
        assert h0[0] == ox0 and h0[1] == h1[0] and h1[1] == ox1
        assert 0 < h0[1] - h0[0] <= COLS_HALF and 0 <= h1[1] - h1[0] <= COLS_HALF
    # -- the kernel's weight slots, restated from ScaleStreamH2Kernel: lane (pair, par) walks pixels n0(A) + par + 2j;
    # hw_a[j] = A's tap 2j + par (j < TAPS); hw_b[t] = B's tap par + 2 (JS + t) - d (t <= TAPS), d = n0(B) - n0(A)
    for pairs in t["pairs"]:
        for a, b in pairs:
            if a < 0:
                continue
            got_a = {}
            for par in (0, 1):
                ks = [2 * j + par for j in range(taps)]
                for k in ks:
                    if k < cnt[a]:
                        assert k not in got_a
                        got_a[k] = par
            assert sorted(got_a) == list(range(cnt[a])), (a, cnt[a])  # every tap of A, once, in one of its two chains
            if b < 0:
                continue
            d = n0[b] - n0[a]
            got_b = {}
            for par in (0, 1):
                ks = [par + 2 * (js + s) - d for s in range(taps + 1)]
                valid = [k for k in ks if 0 <= k < cnt[b]]
                assert valid == sorted(valid)                      # ascending: stb's accumulation order
                assert len({k & 1 for k in valid}) <= 1            # ONE chain of B per lane
                # the slots in front of the first and behind the last valid tap are padding (weight 0), never a hole
                if valid:
                    first, last = ks.index(valid[0]), ks.index(valid[-1])
                    assert ks[first:last + 1] == valid
                for k in valid:
                    assert k not in got_b
                    got_b[k] = par
                # the pixel under B's tap k is the pixel this lane reads at that step
                for s, k in enumerate(ks):
                    if 0 <= k < cnt[b]:
                        assert n0[a] + par + 2 * (js + s) == n0[b] + k
            assert sorted(got_b) == list(range(cnt[b])), (a, b, d, js, cnt[b], sorted(got_b)[:5])
            # (weights themselves are read from the plan's own table by column and tap: nothing to restate)
            assert coeff.shape[1] >= max(cnt[a], cnt[b])


def test_plans_outside_the_range_keep_the_one_column_kernel():
    # vertical-first (4K -> 800x450), few taps (640x480 -> 67x50: 9.6:1 but ... horizontal-first with 39 taps pairs up), upscales
    assert tiling(3840, 2160, 800, 450) is None
    assert tiling(800, 600, 1600, 1200) is None
    assert tiling(1280, 960, 120, 90) is None   # 43 taps: the 40-tap instantiation has no two-column form
    assert tiling(2000, 400, 800, 40) is None   # 10 taps: the 8-tap-per-lane instantiation has no two-column form
