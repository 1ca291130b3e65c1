"""CPU-only: the restatement of the graphics-protocol path at --compress=0 (oracle/png.c)
against the REAL reference -- png::Encode with this image's libdeflate, KittyGraphicsCanvas and
ITerm2GraphicsCanvas through their own thread pool and write sequencer (oracle/_ref) -- and the
golden vectors that carry the pin to the GPU box (no device code yet)."""
import os
import re
import zlib

import numpy as np
import pytest

import oracle_lib

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "png.npz")


def _frame(rng, w, h, kind):
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    fb = np.stack([(x * 3 + y) & 255, (x + y * 5) & 255, (x * y) & 255, 255 - ((x + y) & 127)], -1)
    return fb.astype(np.uint8)


@pytest.fixture(scope="module")
def png_ref(ref):
    if ref is None or not ref.has_png():
        pytest.skip("reference library without png/kitty/iterm2 (libdeflate or /root/reference absent)")
    return ref


CASES = [(1, 1), (5, 3), (67, 50), (200, 56), (128, 128), (129, 127), (800, 21), (181, 91)]


@pytest.mark.parametrize("w,h", CASES)
@pytest.mark.parametrize("with_alpha", [True, False])
def test_png_level0_matches_reference_and_libdeflate(oracle, png_ref, w, h, with_alpha):
    rng = np.random.default_rng(w * 1000 + h)
    for kind in ("noise", "ramp"):
        fb = _frame(rng, w, h, kind)
        want = png_ref.png_encode(fb, level=0, with_alpha=with_alpha)
        got = oracle.png_encode(fb, with_alpha)
        assert got == want, (w, h, with_alpha, kind, len(got), len(want))


def test_png_with_several_stored_blocks(oracle, png_ref):
    """More than 65535 filtered bytes: libdeflate splits into stored blocks of 65535."""
    rng = np.random.default_rng(5)
    for w, h in ((400, 300), (4095, 5), (16384 // 4 + 3, 4)):
        fb = _frame(rng, w, h, "noise")
        assert oracle.png_encode(fb, True) == png_ref.png_encode(fb, 0, True), (w, h)
        assert oracle.png_encode(fb, False) == png_ref.png_encode(fb, 0, False), (w, h)


def test_png_decodes_back_to_the_pixels(oracle):
    """Independent of the reference: the stream is a valid PNG whose sub-filtered rows give the frame."""
    rng = np.random.default_rng(9)
    fb = _frame(rng, 37, 11, "noise")
    data = oracle.png_encode(fb, True)
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[-12:-8] == b"\0\0\0\0" and data[-8:-4] == b"IEND"
    idat = data[8 + 25 + 8:-12 - 4]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(11, 1 + 37 * 4)
    assert (raw[:, 0] == 1).all()
    rows = raw[:, 1:].reshape(11, 37, 4).astype(np.uint16)
    assert np.array_equal(np.cumsum(rows, axis=1).astype(np.uint8), fb)


def test_checksum_building_blocks(oracle):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 100_000, dtype=np.uint8).tobytes()
    assert oracle.crc32(data) == zlib.crc32(data)
    assert oracle.adler32(data) == zlib.adler32(data)
    # the combine rule a parallel CRC is built from, at uneven cut points
    for cut in (0, 1, 255, 256, 8192, 65535, 99_999, 100_000):
        a, b = data[:cut], data[cut:]
        assert oracle.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(data), cut


def test_base64(oracle):
    import base64
    rng = np.random.default_rng(4)
    for n in (0, 1, 2, 3, 4, 5, 3071, 3072, 3073, 10_000):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.base64(d) == base64.b64encode(d)


@pytest.mark.parametrize("w,h", [(5, 3), (67, 50), (200, 56), (30, 26), (400, 300)])
def test_kitty_and_iterm2_bytes_match_the_real_canvases(oracle, png_ref, w, h):
    rng = np.random.default_rng(w + h)
    fb = _frame(rng, w, h, "noise")
    for local_alpha in (False, True):
        real = png_ref.graphics_send(0, fb, level=0, local_alpha=local_alpha)
        m = re.search(rb"i=(\d+),", real)
        assert m, real[:80]
        image_id = int(m.group(1))  # (time based in the reference: src/kitty-canvas.cc:47-52)
        mine = oracle.kitty_encode(fb, image_id, with_alpha=not local_alpha)
        assert real.endswith(mine), (w, h, local_alpha, len(real), len(mine))
        assert len(real) - len(mine) <= 8  # only a cursor prefix may stand in front
        real = png_ref.graphics_send(1, fb, level=0, local_alpha=local_alpha)
        mine = oracle.iterm2_encode(fb, with_alpha=not local_alpha)
        assert real.endswith(mine) and len(real) - len(mine) <= 8, (w, h, local_alpha)


def test_golden_vectors(oracle):
    """Vectors generated from the real reference (tests/golden/make_golden.py): valid on the GPU
    box, where neither /root/reference nor oracle/_ref's sources exist."""
    if not os.path.exists(GOLDEN):
        pytest.skip("tests/golden/png.npz not generated")
    g = np.load(GOLDEN)
    n = int(g["count"])
    assert n >= 6
    for i in range(n):
        fb, with_alpha = g[f"fb{i}"], bool(g[f"alpha{i}"])
        assert oracle.png_encode(fb, with_alpha) == g[f"png{i}"].tobytes(), i
        assert oracle.kitty_encode(fb, int(g[f"id{i}"]), with_alpha) == g[f"kitty{i}"].tobytes(), i
        assert oracle.iterm2_encode(fb, with_alpha) == g[f"iterm{i}"].tobytes(), i
