"""CPU-only: the index arithmetic of the device graphics-protocol path (timg_amd/csrc/
gfx_layout.h -- offsets, stored-block headers, Adler-32 from plain sums, CRC-32 from chunk CRCs,
base64 groups, kitty chunk framing), run with host loops through a debug entry point of the
product library, must reproduce the oracle (itself pinned against the real reference) byte for
byte.  The kernels call the same functions per lane."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulate():
    L = ctypes.CDLL(os.path.join(ROOT, "timg_amd", "libtimg_hip_debug.so"))  # test-only library
    f = L.timg_hip_debug_gfx_emulate
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                  ctypes.c_void_p, ctypes.c_long]

    def run(kind, fb, flags=0, image_id=0):
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = 4096 + (w * h * 4 + h + 1024) * 2
        out = np.zeros(cap, np.uint8)
        n = f(kind, fb.ctypes.data, w, h, flags, image_id, out.ctypes.data, cap)
        assert n > 0
        return out[:n].tobytes()
    return run


def _fb(rng, w, h):
    return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)


SIZES = [(1, 1), (2, 1), (5, 3), (67, 50), (30, 26), (129, 127), (200, 90), (400, 300), (4095, 5), (127, 129),
         (819, 20), (16383, 1), (341, 3)]


# form 0: one byte / one base64 group per index; form 2: four pixels / four groups per index, the way the
# kernels walk (dword loads, byte-wise Sub filter, sums by byte sums and dot products, arithmetic alphabet)
@pytest.mark.parametrize("form", [0, 2])
@pytest.mark.parametrize("w,h", SIZES)
def test_png_layout(oracle, emulate, w, h, form):
    rng = np.random.default_rng(w * 7 + h)
    fb = _fb(rng, w, h)
    assert emulate(0, fb, form) == oracle.png_encode(fb, True)
    assert emulate(0, fb, form | 1) == oracle.png_encode(fb, False)


@pytest.mark.parametrize("form", [0, 2])
@pytest.mark.parametrize("w,h", SIZES)
def test_kitty_and_iterm2_layout(oracle, emulate, w, h, form):
    rng = np.random.default_rng(w * 11 + h)
    fb = _fb(rng, w, h)
    for image_id in (7, 4_000_000_123):
        assert emulate(1, fb, form, image_id) == oracle.kitty_encode(fb, image_id, True)
    assert emulate(1, fb, form | 1, 99) == oracle.kitty_encode(fb, 99, False)
    assert emulate(2, fb, form) == oracle.iterm2_encode(fb, True)
    assert emulate(2, fb, form | 1) == oracle.iterm2_encode(fb, False)


def test_sizes_around_the_block_chunk_and_segment_boundaries(oracle, emulate):
    """Filtered sizes of exactly 65535 / 65536 bytes, checksummed regions of exactly n x 512 and
    n x 32768 bytes, PNG sizes around multiples of 3072."""
    rng = np.random.default_rng(1)
    hit = 0
    for w in list(range(1, 40)) + [255, 256, 257, 1023, 1024, 1025, 4095, 5461, 5460, 16383, 21844, 21845]:
        for h in (1, 2, 3, 4, 5, 7):
            if w * h > 70_000:
                continue
            fb = _fb(rng, w, h)
            assert emulate(0, fb, 0) == oracle.png_encode(fb, True), (w, h)
            assert emulate(1, fb, 1, 5) == oracle.kitty_encode(fb, 5, False), (w, h)
            assert emulate(0, fb, 2 | 1) == oracle.png_encode(fb, False), (w, h)       # (group-wise forms)
            assert emulate(2, fb, 2) == oracle.iterm2_encode(fb, True), (w, h)
            hit += 1
    assert hit > 200


def test_crc_tree_across_segments(oracle, emulate):
    """More than 1024 chunk CRCs: the tree continues across the workgroups' segments (a segment = levels 0..9);
    checksummed regions of exactly 1024 x 512 bytes and one byte more, two segments and a bit, three segments."""
    rng = np.random.default_rng(5)
    for w, h, rgb in ((5957, 22, 0), (5698, 23, 0), (8191, 16, 0), (14562, 24, 1), (14569, 24, 1), (17483, 15, 0),
                      (800, 450, 0), (400, 340, 1)):
        fb = _fb(rng, w, h)
        assert emulate(0, fb, 2 | rgb) == oracle.png_encode(fb, not rgb), (w, h, rgb)
    fb = _fb(rng, 800, 450)
    assert emulate(1, fb, 2, 77) == oracle.kitty_encode(fb, 77, True)
