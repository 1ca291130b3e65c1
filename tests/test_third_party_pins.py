"""Conditional pins for the three third-party libraries behind SURVEY.md 8 rows a2, a6, a13 and a14.

hzeller/timg does its sixel work in libsixel (src/sixel-canvas.cc:137-145), its default scaling in libswscale
(src/image-scaler.cc:45-72) and --auto-crop in GraphicsMagick's trim() (src/graphics-magick-source.cc:231-241).  None of
the three is vendored by the reference, none is in the build image, none can be fetched: the restatements under oracle/
that stand in for them are "parity unpinned" (DESIGN.md 2).  These tests are what turns each of those rows into a
number THE DAY A BOX HAS THE LIBRARY, without a code change here:

    libsixel      ctypes -> sixel_dither_new(256) / sixel_dither_initialize(RGBA8888, LARGE_LUM, REP_AVERAGE_COLORS,
                  QUALITY_AUTO) / sixel_encode -- the reference's own call sequence -- against oracle/sixel.c under
                  libsixel's lookup rule (lookup_mode 0), byte for byte; on a GPU box also against the device's
                  first-hit checker (libtimg_hip_debug.so) and, as a stated colour difference, the product's one rule.
    libswscale    ctypes -> sws_getContext(RGBA -> RGBA, SWS_BILINEAR) + sws_scale against the restatement's triangle
                  filter (TIMG_HIP_FILTER_TRIANGLE): the largest difference in LSB, reported and bounded.
    GraphicsMagick `gm convert -trim` against oracle/autocrop.c (and autocrop.hip on a GPU box): the same box.

Where a library is absent the test is SKIPPED with the reason (that is every box so far); recipes: oracle/README.md.
Environment overrides: TIMG_LIBSIXEL, TIMG_LIBSWSCALE (paths of the shared objects), TIMG_GM (the gm binary).
Report: gpurun_out/third_party_pins.txt (appended)."""
import ctypes
import ctypes.util
import os
import shutil
import subprocess

import numpy as np
import pytest

from timg_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BG, PAT = (30, 30, 46, 255), (96, 96, 128, 255)


def _find(env, name):
    p = os.environ.get(env) or ctypes.util.find_library(name)
    if not p:
        return None
    try:
        return ctypes.CDLL(p)
    except OSError:
        return None


def _report(line):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "third_party_pins.txt"), "a") as f:
        f.write(line + "\n")


def test_the_probe_says_which_libraries_this_box_has():
    """Never skipped: the record of WHY the three pins below did or did not run on this box."""
    have = {"libsixel": _find("TIMG_LIBSIXEL", "sixel") is not None,
            "libswscale": _find("TIMG_LIBSWSCALE", "swscale") is not None,
            "GraphicsMagick (gm)": bool(os.environ.get("TIMG_GM") or shutil.which("gm"))}
    _report("probe: " + ", ".join("%s %s" % (k, "FOUND" if v else "absent") for k, v in have.items()))
    assert set(have) == {"libsixel", "libswscale", "GraphicsMagick (gm)"}


# ---------------------------------------------------------------------------------------------------------------------
# libsixel (a13 / a14).  Constants and signatures: libsixel's sixel.h, the slice src/sixel-canvas.cc uses (the same
# values oracle/stub/sixel.h declares).
SIXEL_PIXELFORMAT_RGBA8888, SIXEL_LARGE_LUM, SIXEL_REP_AVERAGE_COLORS, SIXEL_QUALITY_AUTO = 0x11, 0x2, 0x2, 0x0
_WRITE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_char), ctypes.c_int, ctypes.c_void_p)


def real_libsixel_encode(lib, rgba):
    """src/sixel-canvas.cc:134-148 through ctypes: the bytes a libsixel-linked timg puts between its cursor strings."""
    h, w = rgba.shape[:2]
    assert h % 6 == 0 and rgba.dtype == np.uint8 and rgba.shape[2] == 4
    chunks = []

    def write(data, size, _priv):
        chunks.append(ctypes.string_at(data, size))
        return size

    cb = _WRITE_FN(write)
    for name, argt in (("sixel_output_new", [ctypes.POINTER(ctypes.c_void_p), _WRITE_FN, ctypes.c_void_p, ctypes.c_void_p]),
                       ("sixel_dither_new", [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
                       ("sixel_dither_initialize", [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6),
                       ("sixel_encode", [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p])):
        getattr(lib, name).argtypes = argt
        getattr(lib, name).restype = ctypes.c_int
    for name in ("sixel_output_destroy", "sixel_dither_destroy", "sixel_output_unref", "sixel_dither_unref"):
        if hasattr(lib, name):
            getattr(lib, name).argtypes = [ctypes.c_void_p]
            getattr(lib, name).restype = None
    out, dither = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.sixel_output_new(ctypes.byref(out), cb, None, None) == 0
    assert lib.sixel_dither_new(ctypes.byref(dither), 256, None) == 0
    px = np.ascontiguousarray(rgba).copy()  # (libsixel diffuses in place)
    ptr = px.ctypes.data_as(ctypes.c_void_p)
    assert lib.sixel_dither_initialize(dither, ptr, w, h, SIXEL_PIXELFORMAT_RGBA8888, SIXEL_LARGE_LUM,
                                       SIXEL_REP_AVERAGE_COLORS, SIXEL_QUALITY_AUTO) == 0
    assert lib.sixel_encode(ptr, w, h, 0, dither, out) == 0
    (getattr(lib, "sixel_dither_unref", None) or lib.sixel_dither_destroy)(dither)
    (getattr(lib, "sixel_output_unref", None) or lib.sixel_output_destroy)(out)
    return b"".join(chunks)


def restated_libsixel_encode(oracle, rgba, lookup_mode=0):
    h, w = rgba.shape[:2]
    cap = 4096 + w * (h + 6) * 8
    buf = ctypes.create_string_buffer(cap)
    fn = oracle.L.oracle_libsixel_encode
    fn.restype = ctypes.c_long
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_long]
    src = np.ascontiguousarray(rgba)
    n = fn(src.ctypes.data_as(ctypes.c_void_p), w, h, lookup_mode, buf, cap)
    assert n > 0
    return buf.raw[:n]


def pin_frames(oracle):
    """The twelve frames of tests/test_sixel_delta_e.py that need no device (the 8K-derived one is replaced by a second
    checkerboard geometry), heights padded to a multiple of 6 the way SixelCanvas::Send pads them."""
    from test_sixel_delta_e import gradient
    a = synth.alpha(800, 450, 5)
    fr = [("S-noise 800x450", synth.noise(800, 450, 1, opaque=True))]
    fr += [("S-photo 800x450 seed %d" % s, synth.photo(800, 450, s)) for s in (3, 11, 29)]
    fr += [("S-alpha over solid", oracle.alpha_compose(a, BG)[0]),
           ("S-alpha over checkerboard", oracle.alpha_compose(a, BG, PAT, 18, 18)[0]),
           ("S-alpha over fine checkerboard", oracle.alpha_compose(a, BG, PAT, 3, 2)[0]),
           ("2-colour ramp 800x450", gradient(800, 450, 2)), ("300-colour ramp 800x450", gradient(800, 450, 300)),
           ("S-photo 333x516 (odd width)", synth.photo(333, 516, 7)), ("S-photo 1365x30 (widest LDS band)", synth.photo(1365, 30, 9)),
           ("S-noise 64x36", synth.noise(64, 36, 2, opaque=True))]
    return [(n, np.ascontiguousarray(f[: f.shape[0] // 6 * 6])) for n, f in fr]


def test_the_libsixel_harness_itself_works_against_the_stub(oracle):
    """The ctypes call sequence above has never met a real libsixel.  It HAS met the one other implementation of that
    API there is: oracle/stub (the slice of sixel.h src/sixel-canvas.cc uses, forwarding to the restatement), which
    oracle/_ref/libtimg_ref.so exports.  Against it the pin must be green by construction -- output callback, in-place
    pixel buffer, argument order, object lifetimes all exercised; what a real library adds is only its own bytes."""
    path = os.path.join(ROOT, "oracle", "_ref", "libtimg_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libtimg_ref.so not built (needs /root/reference at build time)")
    lib = ctypes.CDLL(path)
    if not hasattr(lib, "sixel_encode"):
        pytest.skip("reference library built without the sixel canvas")
    lib.timg_stub_sixel_set_lookup_mode(0)
    for name, fb in pin_frames(oracle)[-3:]:
        assert real_libsixel_encode(lib, fb) == restated_libsixel_encode(oracle, fb, 0), name


def test_libsixel_pin_restatement_is_libsixel_byte_for_byte(oracle):
    """a13 + a14 on the CPU: the real library against oracle/sixel.c (lookup_mode 0 = libsixel's own cache rule).  Green
    here means the restatement IS libsixel on these frames, and everything pinned to the restatement (the device's
    histogram, median cut, diffusion under the first-hit rule, RLE bytes) is pinned to libsixel with it."""
    lib = _find("TIMG_LIBSIXEL", "sixel")
    if lib is None:
        pytest.skip("libsixel.so not found (set TIMG_LIBSIXEL; oracle/README.md \"Pinning the unpinned\")")
    bad = []
    for name, fb in pin_frames(oracle):
        real, ours = real_libsixel_encode(lib, fb), restated_libsixel_encode(oracle, fb, 0)
        same = real == ours
        first = next((i for i, (x, y) in enumerate(zip(real, ours)) if x != y), min(len(real), len(ours)))
        _report("libsixel pin, %-36s real %7d B, restatement %7d B: %s" % (name, len(real), len(ours),
                                                                       "identical" if same else "first difference at byte %d" % first))
        if not same:
            bad.append((name, first, len(real), len(ours)))
    assert not bad, bad


@pytest.mark.gpu
def test_libsixel_pin_device_first_hit_checker_and_the_products_tolerance(hip, oracle):
    """On a GPU box with libsixel: (1) the device encoder under libsixel's rule (libtimg_hip_debug.so) carries the real
    library's bytes between the canvas' cursor strings; (2) the PRODUCT's stream (one rule: cell centre) decodes to
    within the tolerance tests/test_sixel_delta_e.py states -- this time against the real library's picture."""
    lib = _find("TIMG_LIBSIXEL", "sixel")
    if lib is None:
        pytest.skip("libsixel.so not found (set TIMG_LIBSIXEL; oracle/README.md \"Pinning the unpinned\")")
    from test_sixel_delta_e import BOUNDS, delta_e_map, stats
    for name, fb in pin_frames(oracle):
        h, w = fb.shape[:2]
        real = real_libsixel_encode(lib, fb)
        first_hit = hip.sixel_encode_first_hit(fb, w, h)[0]
        assert real in first_hit and len(first_hit) - len(real) < 64, (name, len(real), len(first_hit))
        product = hip.sixel_encode(fb, w, h)[0]
        img_p, _ = oracle.sixel_decode(product)
        img_r, _ = oracle.sixel_decode(first_hit)
        mean, p99, mx = stats(delta_e_map(img_p[..., :3], img_r[..., :3]))
        _report("libsixel pin, %-36s product vs libsixel picture: mean %.2f p99 %.2f max %.2f CIE76" % (name, mean, p99, mx))
        assert mean <= BOUNDS["between_mean"] and p99 <= BOUNDS["between_p99"] and mx <= BOUNDS["between_max"], (name, mean, p99, mx)


# ---------------------------------------------------------------------------------------------------------------------
# libswscale (a2): the scaler a stock timg build uses.
AV_PIX_FMT_RGBA, SWS_BILINEAR = 26, 2


def real_swscale(lib, src, dw, dh):
    """src/image-scaler.cc:50-56,:64-65: sws_getContext(in, RGBA -> out, RGBA, SWS_BILINEAR) + one sws_scale."""
    sh, sw = src.shape[:2]
    lib.sws_getContext.restype = ctypes.c_void_p
    lib.sws_getContext.argtypes = [ctypes.c_int] * 3 + [ctypes.c_int] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.sws_scale.restype = ctypes.c_int
    lib.sws_scale.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
    lib.sws_freeContext.argtypes = [ctypes.c_void_p]
    ctx = lib.sws_getContext(sw, sh, AV_PIX_FMT_RGBA, dw, dh, AV_PIX_FMT_RGBA, SWS_BILINEAR, None, None, None)
    assert ctx
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw, 4), np.uint8)
    sp = (ctypes.c_void_p * 4)(src.ctypes.data, None, None, None)
    dp = (ctypes.c_void_p * 4)(dst.ctypes.data, None, None, None)
    ss, ds = (ctypes.c_int * 4)(sw * 4, 0, 0, 0), (ctypes.c_int * 4)(dw * 4, 0, 0, 0)
    assert lib.sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
    lib.sws_freeContext(ctx)
    return dst


SWS_GEOMS = [(640, 480, 67, 50), (1920, 1080, 400, 225), (3840, 2160, 800, 450), (320, 200, 100, 56), (50, 40, 120, 90)]
# libswscale filters in 14-bit fixed point with its own rounding and chroma-free RGBA path: the triangle filter in float
# cannot be bit-identical; what the pin states is HOW FAR apart the two are.  Bound: 3 LSB anywhere, 0.6 LSB in the mean
# (to be tightened to the measured figure by whoever first runs this with the library present).
SWS_MAX_LSB, SWS_MEAN_LSB = 3, 0.6


def test_swscale_pin_triangle_filter_against_sws_bilinear(oracle):
    lib = _find("TIMG_LIBSWSCALE", "swscale")
    if lib is None:
        pytest.skip("libswscale.so not found (set TIMG_LIBSWSCALE; oracle/README.md \"Pinning the unpinned\")")
    worst = 0
    for sw, sh, dw, dh in SWS_GEOMS:
        src = synth.photo(sw, sh, 17)
        real = real_swscale(lib, src, dw, dh)
        ours = oracle.scale(src, dw, dh, filter=2)  # TIMG_HIP_FILTER_TRIANGLE
        d = np.abs(real.astype(np.int32) - ours.astype(np.int32))
        _report("swscale pin, %dx%d -> %dx%d: max %d LSB, mean %.3f LSB, %.2f %% of bytes differ" % (
            sw, sh, dw, dh, int(d.max()), float(d.mean()), 100.0 * float((d != 0).mean())))
        worst = max(worst, int(d.max()))
        assert d.max() <= SWS_MAX_LSB and d.mean() <= SWS_MEAN_LSB, (sw, sh, dw, dh, int(d.max()), float(d.mean()))


@pytest.mark.gpu
def test_swscale_pin_device_triangle_filter(hip, oracle):
    lib = _find("TIMG_LIBSWSCALE", "swscale")
    if lib is None:
        pytest.skip("libswscale.so not found (set TIMG_LIBSWSCALE; oracle/README.md \"Pinning the unpinned\")")
    for sw, sh, dw, dh in SWS_GEOMS:
        src = synth.photo(sw, sh, 17)
        real = real_swscale(lib, src, dw, dh)
        got = hip.scale(src, dw, dh, filter=2)  # TIMG_HIP_FILTER_TRIANGLE
        d = np.abs(real.astype(np.int32) - got.astype(np.int32))
        _report("swscale pin (device), %dx%d -> %dx%d: max %d LSB, mean %.3f LSB" % (sw, sh, dw, dh, int(d.max()), float(d.mean())))
        assert d.max() <= SWS_MAX_LSB and d.mean() <= SWS_MEAN_LSB, (sw, sh, dw, dh, int(d.max()), float(d.mean()))


# ---------------------------------------------------------------------------------------------------------------------
# GraphicsMagick trim() (a6).
def _crop_cases():
    rng = np.random.RandomState(5)
    cases = []
    for (w, h, box, border) in [(120, 80, (10, 5, 90, 60), (20, 30, 40, 255)), (64, 64, (0, 0, 64, 64), (0, 0, 0, 255)),
                                (200, 50, (150, 40, 50, 10), (255, 255, 255, 255)), (90, 70, (1, 1, 88, 68), (7, 7, 7, 255)),
                                (33, 47, (16, 20, 1, 1), (200, 10, 10, 255))]:
        fb = np.empty((h, w, 4), np.uint8)
        fb[...] = np.array(border, np.uint8)
        x, y, bw, bh = box
        inner = rng.randint(0, 256, size=(bh, bw, 4)).astype(np.uint8)
        inner[..., 3] = 255
        # the corners of the box must differ from the border or trim() would cut further
        inner[0, 0] = inner[-1, -1] = inner[0, -1] = inner[-1, 0] = (np.array(border, np.int32) ^ 0x55).astype(np.uint8)
        inner[..., 3] = 255
        fb[y:y + bh, x:x + bw] = inner
        cases.append(fb)
    return cases


def real_gm_trim(gm, fb, tmp):
    """`gm convert in.png -trim out.png`: Image::trim() at fuzz 0, what src/graphics-magick-source.cc:239 calls."""
    from PIL import Image
    src, dst = os.path.join(tmp, "in.png"), os.path.join(tmp, "out.png")
    Image.fromarray(fb, "RGBA").save(src)
    r = subprocess.run([gm, "convert", src, "-trim", "+repage", dst], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return np.array(Image.open(dst).convert("RGBA"))


def test_graphicsmagick_pin_trim_against_the_autocrop_restatement(oracle, tmp_path):
    gm = os.environ.get("TIMG_GM") or shutil.which("gm")
    if not gm:
        pytest.skip("GraphicsMagick's gm not found (set TIMG_GM; oracle/README.md \"Pinning the unpinned\")")
    for i, fb in enumerate(_crop_cases()):
        trimmed = real_gm_trim(gm, fb, str(tmp_path))
        x, y, w, h = oracle.autocrop_bbox(fb, 0)
        ours = fb[y:y + h, x:x + w]
        _report("GraphicsMagick pin, case %d: trim() -> %dx%d, restatement box %dx%d+%d+%d" % (i, trimmed.shape[1], trimmed.shape[0], w, h, x, y))
        assert trimmed.shape == ours.shape and np.array_equal(trimmed, ours), (i, trimmed.shape, (x, y, w, h))


@pytest.mark.gpu
def test_graphicsmagick_pin_device_autocrop(hip, oracle, tmp_path):
    gm = os.environ.get("TIMG_GM") or shutil.which("gm")
    if not gm:
        pytest.skip("GraphicsMagick's gm not found (set TIMG_GM; oracle/README.md \"Pinning the unpinned\")")
    for i, fb in enumerate(_crop_cases()):
        trimmed = real_gm_trim(gm, fb, str(tmp_path))
        x, y, w, h = [int(v) for v in hip.autocrop_bbox(fb, fb.shape[1], fb.shape[0])[0]]
        assert trimmed.shape[:2] == (h, w) and np.array_equal(trimmed, fb[y:y + h, x:x + w]), (i, (x, y, w, h))
