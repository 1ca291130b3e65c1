"""Pins the oracle (this repo's CPU restatement) to the REAL reference compiled
from /root/reference into oracle/_ref (skipped where that tree is absent; the
committed golden vectors in tests/golden/ then carry the pin)."""
import numpy as np
import pytest

from timg_amd import synth

SIZES = [(64, 48, 20, 15), (640, 480, 67, 50), (100, 100, 10, 10), (33, 17, 7, 5),
         (50, 40, 120, 90), (64, 64, 64, 64), (64, 48, 64, 20), (64, 48, 20, 48),
         (200, 150, 31, 150), (500, 400, 13, 11), (37, 29, 111, 87), (300, 300, 7, 3),
         (960, 540, 200, 56), (17, 1000, 5, 20), (1, 1, 3, 3), (3, 3, 1, 1)]


@pytest.mark.parametrize("sw,sh,dw,dh", SIZES)
def test_scale_matches_reference(oracle, ref, sw, sh, dw, dh):
    for kind in ("noise", "alpha", "photo"):
        for fmt in (0, 1):
            src = synth.make(kind, sw, sh, seed=dw)
            assert np.array_equal(oracle.scale(src, dw, dh, fmt), ref.scale(src, dw, dh, fmt)), (kind, fmt)


def test_scale_random_geometries(oracle, ref):
    rng = np.random.default_rng(1234)
    for _ in range(400):
        sw, sh, dw, dh = (int(v) for v in rng.integers(1, 220, 4))
        src = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
        mode = rng.integers(0, 3)
        if mode == 0:
            src[..., 3] = 255
        elif mode == 2:
            src[..., 3] = np.where(rng.random((sh, sw)) < 0.5, 0, src[..., 3])
        assert np.array_equal(oracle.scale(src, dw, dh), ref.scale(src, dw, dh)), (sw, sh, dw, dh)


@pytest.mark.parametrize("sw,sh,dw,dh", [(3840, 2160, 800, 450), (3840, 2160, 200, 56)])
def test_scale_baseline_sizes(oracle, ref, sw, sh, dw, dh):
    src = synth.alpha(sw, sh, seed=1)
    assert np.array_equal(oracle.scale(src, dw, dh), ref.scale(src, dw, dh))


def test_alpha_compose_matches_reference(oracle, ref):
    rng = np.random.default_rng(7)
    for _ in range(600):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        m = rng.integers(0, 4)
        if m == 0:
            fb[..., 3] = 255
        elif m == 1:
            fb[..., 3] = rng.choice([0, 1, 0x5F, 0x60, 254, 255], size=(h, w))
        elif m == 2:
            fb[:h // 2, :, 3] = 255
        bg = (*rng.integers(0, 256, 3), int(rng.choice([0, 255])))
        pat = (*rng.integers(0, 256, 3), int(rng.choice([0, 255])))
        if rng.random() < 0.1:
            pat = bg
        pw, ph = int(rng.integers(-1, 12)), int(rng.integers(-1, 12))
        sr, hg = int(rng.integers(0, h + 1)), bool(rng.random() < 0.9)
        a, ca = oracle.alpha_compose(fb, bg, pat, pw, ph, sr, hg)
        b, cb = ref.alpha_compose(fb, bg, pat, pw, ph, sr, hg)
        assert ca == cb and np.array_equal(a, b)


def test_block_encode_matches_reference(oracle, ref):
    rng = np.random.default_rng(9)
    for _ in range(800):
        w, h = int(rng.integers(1, 50)), int(rng.integers(1, 40))
        fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        m = rng.integers(0, 5)
        if m == 0:
            fb[..., 3] = 255
        elif m == 1:
            fb[..., 3] = rng.choice([0, 1, 0x5F, 0x60, 254, 255], size=(h, w))
        elif m == 2:
            fb[..., 3] = 255
            fb[..., :3] = (fb[..., :3] // 64) * 64
        elif m == 3:
            fb[..., 3] = 255
            fb[:, :, :3] = fb[:1, :1, :3]
        q, up, c256 = (bool(rng.integers(0, 2)) for _ in range(3))
        x = int(rng.integers(0, 9))
        assert oracle.block_encode(fb, q, up, c256, x) == ref.block_encode(fb, q, up, c256, x)


def test_block_frame_diff_sequence_matches_reference(oracle, ref):
    rng = np.random.default_rng(10)
    for _ in range(60):
        q, up, c256 = (bool(rng.integers(0, 2)) for _ in range(3))
        w, h, x = int(rng.integers(2, 30)), int(rng.integers(2, 24)), int(rng.integers(0, 5))
        rc, oc = ref.block_canvas(q, up, c256), oracle.block_canvas(q, up, c256)
        fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        fb[..., 3] = 255
        acc = b""
        for f in range(5):
            dy = 0 if f == 0 else -h
            if f == 3:
                dy = -h - 2  # breaks the emit_difference condition
            k = rng.integers(0, 4)
            fb = fb.copy()
            if k == 1:
                fb[rng.integers(0, h), rng.integers(0, w)] = [1, 2, 3, 255]
            elif k == 2:
                fb[rng.integers(0, h):] = rng.integers(0, 256, 3).tolist() + [255]
            elif k == 3:
                fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
                fb[..., 3] = 255
            rc.send(x, dy, fb)
            acc += oc.send(x, dy, fb)
        assert rc.read_all() == acc
        rc.close()
        oc.close()


def test_256_colour_table(oracle, ref):
    rng = np.random.default_rng(3)
    cols = [(v, v, v, 255) for v in range(256)] + [tuple(rng.integers(0, 256, 4)) for _ in range(3000)]
    for c in cols:
        assert oracle.as_256(c) == ref.as_256(c)


# ---- SixelCanvas::Send: the REAL class, compiled from src/sixel-canvas.cc against
# oracle/stub/sixel.h, whose libsixel calls land in oracle/sixel.c.  What is pinned here is everything the
# class does itself: padding to 6-row bands, the background of the pad rows only, cursor strings, the
# prefix for x / dy, one buffer per Send.  The encoder between is the same code on both sides (unpinned).
def _sixel_frames(rng):
    """Frames whose sixel stream fits the reference's fixed buffer guess (1024 + w*h*5 bytes,
    src/sixel-canvas.cc:122-123 "TODO realloc"): smooth content or few colours, not pure noise."""
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    m = rng.integers(0, 4)
    if m == 0:
        fb = synth.make("photo", w, h, seed=int(rng.integers(0, 1000)))
    elif m == 1:
        fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        fb[..., :3] = (fb[..., :3] // 128) * 128   # few colours: no dithering
        fb[..., 3] = 255
    elif m == 2:
        fb = synth.make("alpha", w, h, seed=int(rng.integers(0, 1000)))
    else:
        fb = np.zeros((h, w, 4), np.uint8)
        fb[..., 0] = np.linspace(0, 255, w, dtype=np.uint8)[None, :]
        fb[..., 1] = np.linspace(0, 255, h, dtype=np.uint8)[:, None]
        fb[..., 3] = 255
    return fb


def _fits_reference_buffer(stream, fb):
    h, w = fb.shape[:2]
    h6 = (h + 5) // 6 * 6
    return len(stream) + 64 < 1024 + w * h6 * 5


def test_sixel_canvas_send_matches_the_reference_class(oracle, ref):
    if not ref.has_sixel():
        pytest.skip("reference library built without the sixel canvas")
    rng = np.random.default_rng(21)
    checked = 0
    for i in range(150):
        fb = _sixel_frames(rng)
        bg = (*rng.integers(0, 256, 3), int(rng.choice([0, 255, 255])))
        pat = (*rng.integers(0, 256, 3), int(rng.choice([0, 255])))
        cx = int(rng.integers(1, 12))
        cy = int(rng.integers(2, 24))
        psize = int(rng.integers(1, 4))
        hg, broken = bool(rng.random() < 0.85), bool(rng.random() < 0.3)
        mode = int(rng.integers(0, 2))
        # the pattern cell of the pad rows: src/sixel-canvas.cc:115-118
        want = oracle.sixel_encode(fb, bg, pat, psize * cx, psize * cy // 2, has_getter=hg,
                                   broken_cursor=broken, lookup_mode=mode)
        if not _fits_reference_buffer(want, fb):
            continue
        checked += 1
        got = ref.sixel_send(fb, cell_x_px=cx, cell_y_px=cy, bg=bg, pattern=pat, pattern_size=psize,
                             has_getter=hg, broken_cursor=broken, lookup_mode=mode)
        assert got == want, (i, fb.shape, bg, pat, cx, cy, psize, hg, broken, mode)
    assert checked > 100


def test_sixel_canvas_cursor_prefix_and_cell_height(oracle, ref):
    if not ref.has_sixel():
        pytest.skip("reference library built without the sixel canvas")
    rng = np.random.default_rng(22)

    def round6(px):  # src/sixel-canvas.cc:91-94
        px += 5
        return px - px % 6

    for _ in range(40):
        fb = _sixel_frames(rng)
        h = fb.shape[0]
        cx, cy = int(rng.integers(1, 12)), int(rng.integers(2, 24))
        x = int(rng.integers(0, 60))
        full = bool(rng.random() < 0.5)
        bg = (30, 30, 46, 255)
        # src/sixel-canvas.cc:157-172
        rows = (round6(h) - 6) // cy + 1 if full else (round6(h) + cy - 1) // cy
        assert ref.sixel_cell_height(-h, cy, full) == -rows
        one = oracle.sixel_encode(fb, bg, (0, 0, 0, 0), cx, cy // 2, lookup_mode=1)
        if not _fits_reference_buffer(one, fb):
            continue
        right = b"\033[%dC" % (x // cx) if x // cx else b""   # src/terminal-canvas.cc:66-82
        up = b"\033[%dA" % rows
        want = right + one + up + right + one + up + right + one
        got = ref.sixel_send(fb, x=x, n_sends=3, cell_x_px=cx, cell_y_px=cy, bg=bg, full_cell_jump=full)
        assert got == want
