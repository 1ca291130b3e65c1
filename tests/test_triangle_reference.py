"""the bilinear (triangle) back-end against an INDEPENDENT float64 resampler.

timg's default build scales with libswscale SWS_BILINEAR (src/image-scaler.cc:45-72); libswscale is
not in the reference tree nor in this image, so that pairing cannot be pinned bit for bit.  What can
be checked is that TIMG_HIP_FILTER_TRIANGLE is the filter it claims to be: a separable triangle
kernel, widened to 1/scale source pixels when shrinking, clamp-to-edge, weights normalised, colour
filtered alpha-weighted.  The reference below is written from that definition alone (dense float64
matrices, no code shared with oracle/ or the product) and the tolerance is the rounding of the
8-bit result: |result - exact| <= 0.5 LSB (+1e-3 for fp32 arithmetic)."""
import numpy as np
import pytest

from timg_amd import synth

TOL = 0.5 + 1e-3
CASES = [("photo", 640, 480, 200, 113), ("alpha", 333, 250, 100, 75), ("noise", 200, 150, 67, 50),
         ("photo", 64, 48, 160, 120), ("alpha", 97, 61, 31, 200), ("photo", 1920, 1080, 400, 225)]


def tri_weights(n_in, n_out):
    scale = n_out / n_in
    W = np.zeros((n_out, n_in))
    for o in range(n_out):
        c = (o + 0.5) / scale  # centre of the output pixel in source coordinates
        stretch = scale if scale < 1 else 1.0
        r = 1.0 / stretch
        for i in range(int(np.floor(c - r - 1)), int(np.ceil(c + r + 1)) + 1):
            W[o, min(max(i, 0), n_in - 1)] += max(0.0, 1.0 - abs((i + 0.5 - c) * stretch))
        W[o] /= W[o].sum()
    return W


def float64_triangle(src, dw, dh):
    sh, sw = src.shape[:2]
    f = src.astype(np.float64) / 255.0
    a = f[..., 3:4]
    pm = np.concatenate([f[..., :3] * a, a], axis=2)
    t = np.einsum("oy,yxc->oxc", tri_weights(sh, dh), pm)
    t = np.einsum("px,oxc->opc", tri_weights(sw, dw), t)
    al = t[..., 3:4]
    rgb = np.where(al > 1e-30, t[..., :3] / np.maximum(al, 1e-30), 0.0)
    return np.concatenate([rgb, al], axis=2) * 255.0


def _frame(kind, sw, sh):
    src = synth.make(kind, sw, sh, seed=3)
    if kind == "noise":
        src[..., 3] = np.maximum(src[..., 3], 8)
    return src


def _check(got, want):
    d = np.abs(got.astype(np.float64) - want)
    assert d[..., 3].max() <= TOL
    visible = want[..., 3] > 1.0  # (the colour of an all-but-transparent pixel is ill-conditioned: 0/0)
    assert d[..., :3][visible].max() <= TOL


@pytest.mark.parametrize("kind,sw,sh,dw,dh", CASES)
def test_oracle_triangle_is_a_triangle_filter(oracle, kind, sw, sh, dw, dh):
    src = _frame(kind, sw, sh)
    _check(oracle.scale(src, dw, dh, filter=2), float64_triangle(src, dw, dh))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,sw,sh,dw,dh", CASES)
def test_hip_triangle_is_a_triangle_filter(hip, kind, sw, sh, dw, dh):
    src = _frame(kind, sw, sh)
    _check(hip.scale(src, dw, dh, filter=2), float64_triangle(src, dw, dh))


# ---- a THIRD-PARTY triangle filter that IS in the image: Pillow's Image.resize(BILINEAR) ------------------------------
# Not libswscale (absent: the a2 row stays "unpinned"), but an independent, widely deployed implementation of the same
# definition -- a triangle kernel widened to 1/scale source pixels when shrinking, weights normalised.  Where the two can
# differ by construction: Pillow filters in two 8-bit passes (the horizontal result is rounded to bytes before the
# vertical pass: up to 0.5 LSB more) with 22-bit fixed-point weights, and it TRUNCATES the kernel at the image border
# where stb's machinery clamps (folds the outside taps onto the edge pixel) -- so the comparison is of the interior, on
# opaque frames (Pillow filters straight alpha, the stb machinery alpha-weighted colour), with a bound of 1 LSB.
PIL_CASES = [("photo", 640, 480, 200, 113), ("noise", 200, 150, 67, 50), ("photo", 1920, 1080, 400, 225),
             ("photo", 64, 48, 160, 120), ("photo", 3840, 2160, 800, 450)]


def _pillow_bilinear(src, dw, dh):
    Image = pytest.importorskip("PIL.Image")
    return np.asarray(Image.fromarray(np.ascontiguousarray(src[..., :3])).resize((dw, dh), Image.BILINEAR)).astype(np.int32)


def _check_against_pillow(got, src, dw, dh):
    d = np.abs(got[..., :3].astype(np.int32) - _pillow_bilinear(src, dw, dh))
    assert d[2:-2, 2:-2].max() <= 1, "interior pixels: within 1 LSB of Pillow's BILINEAR"
    assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("kind,sw,sh,dw,dh", PIL_CASES)
def test_oracle_triangle_against_pillow_bilinear(oracle, kind, sw, sh, dw, dh):
    src = synth.make(kind, sw, sh, seed=3)
    src[..., 3] = 255
    _check_against_pillow(oracle.scale(src, dw, dh, filter=2), src, dw, dh)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,sw,sh,dw,dh", PIL_CASES)
def test_hip_triangle_against_pillow_bilinear(hip, kind, sw, sh, dw, dh):
    src = synth.make(kind, sw, sh, seed=3)
    src[..., 3] = 255
    _check_against_pillow(hip.scale(src, dw, dh, filter=2), src, dw, dh)
