#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (hzeller/timg sources
compiled into oracle/_ref/libtimg_ref.so by oracle/Makefile).  Run in the
container that has /root/reference:

    python -c 'import __graft_entry__ as g; g.build()' && python tests/golden/make_golden.py

The reference has no tests or golden vectors of its own, so these
files ARE the pin: every array below is an output of the unmodified reference
code (ImageScaler::Scale with the STB back-end, Framebuffer::AlphaComposeBackground,
UnicodeBlockCanvas::Send) on seeded synthetic inputs from timg_amd/synth.py.
Small cases store input and output; BASELINE-size cases store the seed and a
SHA-256 of the output.  The sixel entries come from the oracle's restatement
(libsixel is not in the reference tree: parity unpinned) and are labelled so.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from timg_amd import synth  # noqa: E402

BG, PAT = (30, 30, 46, 255), (200, 190, 180, 255)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes() if isinstance(a, np.ndarray) else a).hexdigest()


def main():
    ref = oracle_lib.Ref.try_load()
    assert ref is not None, "build oracle/_ref first (needs /root/reference)"
    orc = oracle_lib.Oracle()

    # ---- scale: small cases with full data ------------------------------------------------
    out = {}
    cases = [("alpha", 64, 48, 20, 15, 0), ("noise", 50, 40, 120, 90, 0), ("alpha", 64, 64, 64, 64, 0),
             ("alpha", 97, 61, 31, 61, 1), ("noise", 33, 200, 7, 20, 0), ("photo", 160, 90, 40, 23, 0),
             ("alpha", 5, 5, 1, 1, 0), ("alpha", 1, 1, 5, 5, 0), ("alpha", 300, 300, 7, 3, 0)]
    for i, (kind, sw, sh, dw, dh, fmt) in enumerate(cases):
        src = synth.make(kind, sw, sh, seed=100 + i)
        out[f"in{i}"] = src
        out[f"out{i}"] = ref.scale(src, dw, dh, fmt)
        out[f"meta{i}"] = np.array([sw, sh, dw, dh, fmt])
    np.savez_compressed(os.path.join(HERE, "scale_small.npz"), **out)

    # ---- scale: BASELINE geometries, hashes only --------------------------------------------
    big = []
    for kind, sw, sh, dw, dh, seed in [("alpha", 640, 480, 67, 50, 1), ("photo", 3840, 2160, 800, 450, 9),
                                       ("alpha", 3840, 2160, 800, 450, 9), ("noise", 3840, 2160, 200, 56, 9),
                                       ("alpha", 3840, 2160, 200, 56, 9), ("alpha", 7680, 4320, 800, 450, 5)]:
        src = synth.make(kind, sw, sh, seed=seed)
        dst = ref.scale(src, dw, dh)
        blended, _ = ref.alpha_compose(dst, BG, PAT, 9, 9)
        big.append((kind, sw, sh, dw, dh, seed, sha(src), sha(dst), sha(blended)))
        print("scale", kind, sw, sh, dw, dh, sha(dst)[:16])
    np.savez_compressed(os.path.join(HERE, "scale_baseline_hashes.npz"),
                        rows=np.array([[str(v) for v in row] for row in big]))

    # ---- alpha compose -----------------------------------------------------------------------
    out = {}
    fb = synth.alpha(57, 33, seed=7)
    sub = fb[::3, ::5, 3]
    sub[...] = np.resize(np.array([0, 1, 0x5F, 0x60, 254, 255], np.uint8), sub.shape)
    out["fb"] = fb
    k = 0
    for pattern, pw, ph in [((0, 0, 0, 0), 0, 0), (PAT, 1, 1), (PAT, 9, 10), (PAT, 10, 9)]:
        for start_row in (0, 7, 33):
            res, calls = ref.alpha_compose(fb, BG, pattern, pw, ph, start_row)
            out[f"res{k}"] = res
            out[f"par{k}"] = np.array([*pattern, pw, ph, start_row, calls])
            k += 1
    np.savez_compressed(os.path.join(HERE, "blend.npz"), **out)

    # ---- block canvas bytes ----------------------------------------------------------------------
    out = {}
    k = 0
    for kind, w, h in [("alpha", 67, 50), ("photo", 100, 56), ("noise", 33, 21), ("alpha", 2, 3), ("photo", 65, 7)]:
        fb = synth.make(kind, w, h, seed=40 + k)
        if kind == "photo":
            fb[..., :3] = (fb[..., :3] // 64) * 64
        out[f"fb{k}"] = fb
        for flags in (0, 1, 2, 4, 5, 6):
            data = ref.block_encode(fb, quarter=bool(flags & 1), upper=bool(flags & 2), color256=bool(flags & 4),
                                    x=(k * 3) % 7)
            out[f"bytes{k}_{flags}"] = np.frombuffer(data, np.uint8)
        out[f"x{k}"] = np.array([(k * 3) % 7])
        k += 1
    # a frame-difference sequence through one canvas (5 Sends)
    rng = np.random.default_rng(5)
    for q in (0, 1):
        canvas = ref.block_canvas(bool(q), False, False)
        fb = synth.photo(40, 24, seed=3)
        sizes = []
        prev = 0
        for f in range(5):
            fb = fb.copy()
            if f in (1, 3):
                fb[rng.integers(0, 24), rng.integers(0, 40):] = [9, 200, 30, 255]
            if f == 4:
                fb[10:] = [1, 2, 3, 255]
            out[f"seq{q}_fb{f}"] = fb
            canvas.send(0, 0 if f == 0 else -24, fb)
            total = len(canvas.read_all())
            sizes.append(total - prev)
            prev = total
        out[f"seq{q}_bytes"] = np.frombuffer(canvas.read_all(), np.uint8)
        out[f"seq{q}_sizes"] = np.array(sizes)
        canvas.close()
    np.savez_compressed(os.path.join(HERE, "block.npz"), **out)

    # ---- sixel: ORACLE RESTATEMENT, parity unpinned ---------------------------------------------
    out = {}
    for k, (kind, w, h) in enumerate([("photo", 100, 56), ("alpha", 64, 45), ("noise", 33, 6), ("photo", 800, 450)]):
        fb = synth.make(kind, w, h, seed=5)
        data = orc.sixel_encode(fb, BG, PAT, 4, 4, lookup_mode=1)
        out[f"meta{k}"] = np.array([w, h, len(data)])
        out[f"kind{k}"] = np.array(kind)
        out[f"sha{k}"] = np.array(sha(data))
        if w * h < 10000:
            out[f"fb{k}"] = fb
            out[f"bytes{k}"] = np.frombuffer(data, np.uint8)
    np.savez_compressed(os.path.join(HERE, "sixel_unpinned.npz"), **out)
    # ---- graphics protocols at --compress=0: the REAL png::Encode (+ libdeflate) and the real
    # kitty / iTerm2 canvases --------------------------------------------------
    if ref.has_png():
        import re
        out, k = {}, 0
        rng = np.random.default_rng(11)
        for (w, h, kind) in [(1, 1, "noise"), (5, 3, "noise"), (67, 50, "noise"), (30, 26, "ramp"), (200, 56, "ramp"),
                             (200, 90, "ramp"), (129, 127, "ramp")]:
            if kind == "noise":
                fb = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            else:
                y, x = np.mgrid[0:h, 0:w]
                fb = np.stack([(x * 3 + y) & 255, (x + y * 5) & 255, (x * y) & 255, 255 - ((x + y) & 127)], -1).astype(np.uint8)
            for with_alpha in (True, False):
                if not with_alpha and k % 4 != 1:
                    continue  # (a few RGB-only cases are enough)
                real = ref.graphics_send(0, fb, level=0, local_alpha=not with_alpha)
                image_id = int(re.search(rb"i=(\d+),", real).group(1))
                kitty = real[real.index(b"\x1b_G"):]
                iterm = ref.graphics_send(1, fb, level=0, local_alpha=not with_alpha)
                iterm = iterm[iterm.index(b"\x1b]1337"):]
                out[f"fb{k}"] = fb
                out[f"alpha{k}"] = np.array(with_alpha)
                out[f"id{k}"] = np.array(image_id, np.uint32)
                out[f"png{k}"] = np.frombuffer(ref.png_encode(fb, 0, with_alpha), np.uint8)
                out[f"kitty{k}"] = np.frombuffer(kitty, np.uint8)
                out[f"iterm{k}"] = np.frombuffer(iterm, np.uint8)
                k += 1
        out["count"] = np.array(k)
        np.savez_compressed(os.path.join(HERE, "png.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
