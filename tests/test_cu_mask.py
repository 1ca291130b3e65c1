"""The CU mask of timg_hip_stream_create (timg_amd/csrc/cu_mask.h) on the CPU, through the test-only debug library.

The layout it is built for -- bit n of the mask is a CU of XCD n % 8, an XCD's bits go round its four shader engines --
is the driver's, and a reserve pays only when every engine of every XCD gives up the same number of CUs
(profiles/r6/partitioned_streams.txt: 4 / 8 / 12 / 16 CUs an XCD against 3 / 5 / 9 / 13).  So: whatever is asked for, the mask
takes the same number of CUs from every engine of every XCD, never all of them, and nothing when nothing is asked for."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dbg():
    L = ctypes.CDLL(os.path.join(ROOT, "timg_amd", "libtimg_hip_debug.so"))
    L.timg_hip_debug_cu_mask.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    return L


def mask_bits(dbg, cus, reserve):
    words = (ctypes.c_uint32 * 32)()
    r = dbg.timg_hip_debug_cu_mask(cus, reserve, words)
    return r, [(words[n >> 5] >> (n & 31)) & 1 for n in range(32 * 32)]


@pytest.mark.parametrize("cus", [256, 304, 128, 64])
def test_every_engine_of_every_xcd_gives_up_the_same_number_of_cus(dbg, cus):
    per_xcd = cus // 8
    for reserve in range(0, per_xcd + 3):
        r, bits = mask_bits(dbg, cus, reserve)
        want = reserve // 4 * 4
        if want >= per_xcd and want > 0:
            assert r == -1, (cus, reserve)
            continue
        if want > 0 and per_xcd % 4:
            assert r == -1, (cus, reserve)
            continue
        assert r == want, (cus, reserve)
        assert sum(bits[cus:]) == 0, "no bit beyond the device's CUs"
        assert sum(bits) == cus - 8 * want
        if want == 0:
            assert bits[:cus] == [1] * cus
            continue
        for xcd in range(8):
            for engine in range(4):
                kept = sum(bits[n] for n in range(cus) if n % 8 == xcd and (n // 8) % 4 == engine)
                assert kept == (per_xcd - want) // 4, (cus, reserve, xcd, engine)


def test_the_mi355x_mask_at_twelve(dbg):
    """256 CUs, 12 an XCD reserved: the first 160 bits set (20 CUs of every XCD = 5 of every engine), the last 96 clear"""
    r, bits = mask_bits(dbg, 256, 12)
    assert r == 12 and bits[:160] == [1] * 160 and sum(bits[160:]) == 0
    r, bits = mask_bits(dbg, 256, 15)  # rounded down
    assert r == 12 and sum(bits) == 160
    r, bits = mask_bits(dbg, 256, 3)   # rounded down to nothing: every CU
    assert r == 0 and sum(bits) == 256


def test_shapes_the_layout_does_not_describe_are_refused(dbg):
    assert mask_bits(dbg, 250, 4)[0] == -1    # not eight equal XCDs
    assert mask_bits(dbg, 80, 4)[0] == -1     # 10 CUs an XCD: not four equal engines
    assert mask_bits(dbg, 256, 32)[0] == -1   # nothing left
    assert mask_bits(dbg, 0, 0)[0] == -1 and mask_bits(dbg, 256, -1)[0] == -1
    assert mask_bits(dbg, 250, 0)[0] == 0     # no reserve: any shape
