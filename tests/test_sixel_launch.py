"""Launch geometry of the sixel kernels (timg_amd/csrc/sixel_launch.h) for EVERY frame width and a ladder of heights,
on the CPU: waves per workgroup, workgroups per frame and dynamic LDS stay inside what a CU has.  A geometry whose
boundary rows filled the LDS to the byte once failed to launch (766 columns, round 2) and was found by a random
stress run on the GPU, not by a test."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan_fn():
    L = ctypes.CDLL(os.path.join(ROOT, "timg_amd", "libtimg_hip_debug.so"))  # test-only library
    f = L.timg_hip_debug_sixel_launch
    f.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_long)]
    f.restype = None
    out = (ctypes.c_long * 13)()
    keys = ("band_ne", "dither_waves", "dither_lds", "dither_parts", "split_share", "split_lds", "wide_bands",
            "nodes_lds", "emit_lds", "budget", "dither_static", "max_waves", "one_trip")

    def plan(w, h6, n_frames=64, cus=256, waves_cap=0, parts_env=-1, trips=0):
        f(w, h6, n_frames, cus, waves_cap, parts_env, trips, out)
        return dict(zip(keys, out))
    return plan


HEIGHTS = (6, 12, 96, 192, 222, 228, 252, 258, 288, 450, 510, 516, 960, 1104, 2160, 4320)


def _check(p, w, h6):
    groups = (h6 + 31) // 32
    room = p["budget"] - p["dither_static"]
    assert 1 <= p["dither_waves"] <= p["max_waves"] and p["dither_lds"] <= room, (w, h6, p)
    assert p["dither_waves"] >= min(groups, 1)
    if p["dither_parts"] > 1:
        parts, share = p["dither_parts"], p["split_share"]
        assert parts <= 16 and w > 2 and groups >= 8, (w, h6, p)
        assert share + 2 <= p["max_waves"] and p["split_lds"] <= room, (w, h6, p)            # fetcher + waves + flusher
        assert share * parts >= groups and groups // parts >= 1, (w, h6, p)                  # no part without rows
    else:
        assert p["split_share"] == 0 and p["split_lds"] == 0
    assert not (p["one_trip"] and w <= 2), (w, h6, p)                                        # the narrow kernel: small tables
    assert p["band_ne"] >= 6 * w and p["band_ne"] % 64 == 0
    assert bool(p["wide_bands"]) == (p["band_ne"] > 8192)
    assert p["nodes_lds"] + 64 <= p["budget"] and p["emit_lds"] <= p["budget"], (w, h6, p)


def test_every_width_fits_a_cu():
    plan = _plan_fn()
    for w in range(1, 4096):
        for h6 in HEIGHTS:
            for n in (1, 64, 300):
                _check(plan(w, h6, n), w, h6)
                if w % 5 == 0:
                    for trips in (1, 2):  # TIMG_HIP_DITHER_TRIPS: either lookup form on request
                        p = plan(w, h6, n, trips=trips)
                        _check(p, w, h6)
                        assert p["one_trip"] == (1 if trips == 1 and w > 2 else 0), (w, h6, n, trips, p)


def test_requested_parts_and_wave_caps_fit_too():
    plan = _plan_fn()
    for w in list(range(1, 1400, 7)) + [766, 800, 825, 1365, 1366, 2048, 4095]:
        for h6 in HEIGHTS:
            for parts in range(0, 20):
                _check(plan(w, h6, 64, 256, 0, parts), w, h6)
            for cap in (1, 2, 7, 13, 16, 99):
                p = plan(w, h6, 64, 256, cap, -1)
                _check(p, w, h6)
                assert p["dither_parts"] == 1 and p["dither_waves"] <= cap  # a wave cap asks for the one-workgroup kernel


def test_known_geometries():
    plan = _plan_fn()
    # the bench frame: four CUs per frame in a batch of 64 (the one-trip lookup: its 96 KB of colour tables leave room
    # for the six boundary rows of a part of four row groups), one workgroup of twelve waves with the small tables
    # when the batch alone fills the chip; three BandNodes workgroups per CU
    p = plan(800, 450, 64)
    assert (p["one_trip"], p["dither_parts"], p["split_share"]) == (1, 4, 4) and 3 * (p["nodes_lds"] + 64) <= p["budget"]
    assert plan(800, 450, 1)["dither_parts"] == 4 and plan(800, 450, 1)["one_trip"] == 1
    p = plan(800, 450, 128)                                    # two parts of eight: only beside the small tables
    assert (p["one_trip"], p["dither_parts"], p["split_share"]) == (0, 2, 8)
    p = plan(800, 450, 300)
    assert (p["one_trip"], p["dither_parts"], p["dither_waves"]) == (0, 1, 12)
    # the last width at which 13 waves' boundary rows + the zero row + the overrun slack fit beside the small tables
    assert plan(755, 450, 300)["dither_waves"] == 12 and plan(754, 450, 300)["dither_waves"] == 13
    assert plan(800, 222, 64)["dither_parts"] == 1             # seven row groups: one workgroup
    assert plan(800, 228, 64)["dither_parts"] == 2             # eight: two parts of four
    assert plan(64, 1104, 64)["dither_parts"] == 4 and plan(64, 1104, 1)["dither_parts"] == 9
    p = plan(1200, 600, 64)                                    # wide boundary rows: five row groups a part, small tables
    assert (p["one_trip"], p["dither_parts"], p["split_share"]) == (0, 4, 5)
    p = plan(1920, 1080, 1)                                    # a full-HD frame alone: twelve parts of three row groups
    assert (p["dither_parts"], p["split_share"]) == (12, 3)
    p = plan(4095, 450, 64)                                    # the widest frame: one wave that follows itself, ONE boundary row
    assert (p["dither_parts"], p["dither_waves"]) == (1, 1) and p["dither_lds"] <= p["budget"] - p["dither_static"]
    p = plan(200, 150, 64)                                     # a small frame: one workgroup, one-trip
    assert (p["one_trip"], p["dither_parts"], p["dither_waves"]) == (1, 1, 5)
    assert plan(2, 4320, 1)["dither_parts"] == 1               # the narrow kernel has no split form
    assert plan(1365, 30)["wide_bands"] == 0 and plan(1366, 30)["wide_bands"] == 1
