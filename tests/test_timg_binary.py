"""A REAL timg with the MI355X back-end, beside a stock one (round 5; VERDICT r4 "missing" item 1).

integration/Makefile copies the reference tree into a scratch directory, applies integration/timg-hip.patch (the edits
INTEGRATION.md prints: ImageScaler::Create src/image-scaler.cc:101-116, the canvas switch of PresentImages
src/timg.cc:319-345, ImageSource::Create src/image-source.cc:155-221, the scale-then-compose pairs of the QOI and STB
loaders, src/CMakeLists.txt) and builds

    integration/build/timg-ref   the tree as it is
    integration/build/timg-hip   the patched tree with -DWITH_TIMG_HIP, linked with the twins and libtimg_hip.so

Both run main() -- option parsing, the loader pool, the renderer, the write sequencer: everything.  Here their terminal
streams are compared byte for byte on real files (a PNG through the STB loader, QOI files, one of them with alpha over
a solid background) for the block canvases, a grid with titles, and sixel.  The binaries are built in this container
(the GPU box has no /root/reference) and travel with the snapshot.

Sixel: libsixel is neither vendored nor installed, so both binaries are built over oracle/stub/sixel.h (test
infrastructure: the API's declarations over the oracle's restatement).  timg-ref then runs the reference's own
SixelCanvas, timg-hip the twin; with the restatement on the device's lookup rule (TIMG_STUB_SIXEL_LOOKUP=1) the streams
are identical -- parity against libsixel itself stays unpinned.
"""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
REF_BIN = os.path.join(INTEG, "build", "timg-ref")
HIP_BIN = os.path.join(INTEG, "build", "timg-hip")
PATCH = os.path.join(INTEG, "timg-hip.patch")

needs_binaries = pytest.mark.skipif(
    not (os.path.exists(REF_BIN) and os.path.exists(HIP_BIN)),
    reason="integration/build/timg-{ref,hip} not built (make -C integration; needs /root/reference at build time)")


def photo(w, h, seed, alpha=False):
    """A smooth picture with a little noise; alpha: a radial ramp with fully transparent and fully opaque parts."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.empty((h, w, 4), np.float64)
    img[..., 0] = 128 + 100 * np.sin(x / (31.0 + seed) + y / 91.0)
    img[..., 1] = 128 + 90 * np.cos(x / 53.0 - y / (23.0 + seed))
    img[..., 2] = 128 + 80 * np.sin((x + y) / 71.0 + seed)
    img[..., :3] += rng.normal(0, 5, (h, w, 3))
    if alpha:
        r = np.hypot(x - w / 2, y - h / 2) / (0.5 * np.hypot(w, h))
        img[..., 3] = np.clip(420 - 520 * r, 0, 255)
    else:
        img[..., 3] = 255
    return np.clip(img, 0, 255).astype(np.uint8)


def write_qoi(path, rgba):
    """A valid QOI stream of QOI_OP_RGBA chunks only (third_party/qoi/qoi.h reads any mix of ops)."""
    h, w, _ = rgba.shape
    body = np.empty((h * w, 5), np.uint8)
    body[:, 0] = 0xFF
    body[:, 1:] = rgba.reshape(-1, 4)
    with open(path, "wb") as f:
        f.write(b"qoif" + struct.pack(">IIBB", w, h, 4, 1))
        f.write(body.tobytes())
        f.write(b"\x00" * 7 + b"\x01")


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("timg_files")
    out = {}
    out["png"] = str(d / "c1.png")  # BASELINE config 1: 640x480 PNG -> -p half -g80x25 (67x25 cells)
    Image.fromarray(photo(640, 480, 1), "RGBA").save(out["png"], compress_level=1)
    out["qoi"] = str(d / "big.qoi")
    write_qoi(out["qoi"], photo(1920, 1080, 2))
    out["qoi_alpha"] = str(d / "alpha.qoi")
    write_qoi(out["qoi_alpha"], photo(1000, 700, 3, alpha=True))
    out["png_odd"] = str(d / "odd.png")
    Image.fromarray(photo(333, 517, 4, alpha=True), "RGBA").save(out["png_odd"], compress_level=1)
    return out


def run(binary, args, out_path, env_extra=None):
    env = dict(os.environ)
    env.pop("TIMG_HIP", None)
    env.update(env_extra or {})
    r = subprocess.run([binary] + args + ["-o", out_path], env=env, capture_output=True, text=True, timeout=300,
                       stdin=subprocess.DEVNULL)
    assert r.returncode == 0, (binary, args, r.stdout[-800:], r.stderr[-2000:])
    with open(out_path, "rb") as f:
        return f.read(), r.stderr


CASES = {
    # name: (args, files)
    "c1_half": (["-ph", "-g80x25"], ["png"]),
    "quarter_qoi": (["-pq", "-g80x25"], ["qoi"]),
    "quarter_alpha_over_bg": (["-pq", "-g100x40", "-b", "#1e1e2e"], ["qoi_alpha"]),
    "half_alpha_checkerboard": (["-ph", "-g90x40", "-b", "#1e1e2e", "-B", "#606080", "--pattern-size=2"], ["png_odd"]),
    "grid_2x2_titles": (["-pq", "-g120x60", "--grid=2x2", "--title", "-b", "#1e1e2e"], ["png", "qoi", "qoi_alpha", "png_odd"]),
    "half_256_colors_upper": (["-ph", "-g80x25", "--color8"], ["qoi"]),
}


def twin_counts(err):
    """The line the twins print when a traced process ends (timg_amd/twins/hip-context.cc: PrintTwinStats): frames
    produced by a device call and frames produced by the reference's classes, per twin."""
    import re
    m = re.search(r"timg_hip twins: frames on the device: scaler (\d+) block (\d+) sixel (\d+) graphics (\d+); on the CPU: "
                  r"scaler (\d+) block (\d+) sixel (\d+) graphics (\d+); degraded (\d)", err)
    assert m, err[-1500:]
    v = [int(x) for x in m.groups()]
    return {"device": dict(zip(("scaler", "block", "sixel", "graphics"), v[:4])),
            "cpu": dict(zip(("scaler", "block", "sixel", "graphics"), v[4:8])), "degraded": v[8]}


@needs_binaries
def test_binaries_run_and_say_what_they_are():
    for b in (REF_BIN, HIP_BIN):
        r = subprocess.run([b, "--version"], capture_output=True, text=True, timeout=60)
        text = r.stdout + r.stderr
        assert "Resize: STB resize" in text and "QOI image loading" in text and "STB image loading" in text, text
        assert "oracle/stub/sixel.h" in text  # (a sixel build over the stub says so)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference tree")
def test_patch_applies_to_the_reference_tree_and_touches_only_the_documented_files(tmp_path):
    tree = tmp_path / "tree"
    tree.mkdir()
    shutil.copytree("/root/reference/src", tree / "src")
    shutil.copy("/root/reference/CMakeLists.txt", tree / "CMakeLists.txt")
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    touched = sorted(l.split()[-1] for l in r.stdout.splitlines() if l.startswith("patching file"))
    assert touched == ["CMakeLists.txt", "src/CMakeLists.txt", "src/image-scaler.cc", "src/image-source.cc",
                       "src/qoi-image-source.cc", "src/stb-image-source.cc", "src/timg.cc"], touched
    # every C++ edit sits behind the build switch: without -DWITH_TIMG_HIP the patched sources are the reference's
    # (structure: the C++ additions are bracketed by #ifdef WITH_TIMG_HIP ... #endif / #else)
    depth, bad, in_cc = 0, [], False
    for l in open(PATCH):
        if l.startswith(("diff ", "--- ", "+++ ", "@@")):
            if l.startswith("diff "):
                in_cc = l.rstrip().endswith(".cc")
                depth = 0
            continue
        if not l.startswith("+") or not in_cc:
            continue
        t = l[1:].strip()
        if t.startswith("#ifdef WITH_TIMG_HIP"):
            depth += 1
        elif t.startswith("#endif") and depth:
            depth -= 1
        elif t.startswith("#else"):
            pass
        elif depth == 0 and t:
            bad.append(t)
    assert not bad, bad


@needs_binaries
@pytest.mark.parametrize("case", sorted(CASES))
def test_patched_timg_with_the_back_end_switched_off_is_the_reference(case, files, tmp_path):
    """TIMG_HIP=0 (or no device): every factory of the patch returns null and the reference's own classes run."""
    args, names = CASES[case]
    paths = [files[n] for n in names]
    want, _ = run(REF_BIN, args + paths, str(tmp_path / "ref.txt"))
    got, _ = run(HIP_BIN, args + paths, str(tmp_path / "hip.txt"), {"TIMG_HIP": "0"})
    assert len(want) > 2000
    assert got == want


@pytest.mark.gpu
@needs_binaries
@pytest.mark.parametrize("case", sorted(CASES))
def test_patched_timg_on_the_device_writes_the_references_bytes(case, files, tmp_path):
    args, names = CASES[case]
    paths = [files[n] for n in names]
    want, _ = run(REF_BIN, args + paths, str(tmp_path / "ref.txt"))
    got, err = run(HIP_BIN, args + paths, str(tmp_path / "hip.txt"), {"TIMG_HIP_TWIN_TRACE": "1"})
    # the device really did the work: context, one scaler per file, the block canvas twin
    assert "timg_hip twins: device context created" in err, err[-1500:]
    assert err.count("HipImageScaler:") >= len(paths), err[-1500:]
    assert "HipUnicodeBlockCanvas: created" in err, err[-1500:]
    # ... and KEPT doing it: a device call that fails after creation makes the twins go on with the reference's classes
    # (cpu-sibling.h) -- the comparison below would then be the reference against itself.  Every image was scaled and
    # every frame encoded by a device call, none by a CPU sibling.
    assert "continuing on the CPU" not in err, err[-1500:]
    counts = twin_counts(err)
    assert counts["degraded"] == 0 and not any(counts["cpu"].values()), counts
    assert counts["device"]["scaler"] == len(paths) and counts["device"]["block"] == len(paths), counts
    assert len(want) > 2000
    assert got == want, "first difference at byte %d of %d / %d" % (
        next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want))), len(got), len(want))


@pytest.mark.gpu
@needs_binaries
def test_patched_timg_sixel_is_the_reference_sixel_canvas_over_the_same_encoder(files, tmp_path):
    """-ps: src/sixel-canvas.cc (timg-ref, over the stub with the device's lookup rule) against HipSixelCanvas (timg-hip):
    cursor strings, pad rows over the background, prefix, the DCS stream -- one byte stream.  Headless, timg assumes 9x18
    cells and shows one frame (src/timg.cc:743-767): -g80x25 is a 600x450 picture, 75 bands."""
    for name, extra in (("png", []), ("qoi_alpha", ["-b", "#1e1e2e"])):
        args = ["-ps", "-g80x25"] + extra + [files[name]]
        want, _ = run(REF_BIN, args, str(tmp_path / "ref.txt"), {"TIMG_STUB_SIXEL_LOOKUP": "1"})
        got, err = run(HIP_BIN, args, str(tmp_path / "hip.txt"), {"TIMG_HIP_TWIN_TRACE": "1"})
        assert "HipSixelCanvas: created" in err and "timg_hip twins: device context created" in err, err[-1500:]
        assert "continuing on the CPU" not in err, err[-1500:]
        counts = twin_counts(err)
        assert counts["degraded"] == 0 and not any(counts["cpu"].values()), counts
        assert counts["device"]["scaler"] == 1 and counts["device"]["sixel"] == 1, counts
        assert b"\x1bPq" in want and len(want) > 20000
        assert got == want, (name, len(got), len(want))


@pytest.mark.gpu
@needs_binaries
@pytest.mark.parametrize("fail_at", [1, 3, 6])
def test_patched_timg_survives_a_device_failure(fail_at, files, tmp_path):
    """TIMG_HIP_FAIL_CALL=k: the k-th device call of the run fails.  The patched timg says once that it continues on the
    CPU and still writes the reference's bytes (still images: no frame differences in the stream)."""
    args, names = CASES["grid_2x2_titles"]
    paths = [files[n] for n in names]
    want, _ = run(REF_BIN, args + paths, str(tmp_path / "ref.txt"))
    got, err = run(HIP_BIN, args + paths, str(tmp_path / "hip.txt"), {"TIMG_HIP_FAIL_CALL": str(fail_at), "TIMG_HIP_TWIN_TRACE": "1"})
    assert err.count("continuing on the CPU") == 1, err[-1500:]
    counts = twin_counts(err)  # (what the device tests above must NOT see)
    assert counts["degraded"] == 1 and sum(counts["cpu"].values()) >= 1, counts
    assert got == want
