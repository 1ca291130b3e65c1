"""The sixel tolerance as a TABLE (round 5; VERDICT r4 "next" item 3).

north_star allows the device a stated colour-difference tolerance for sixel palette selection; libsixel itself is
neither vendored nor installed (parity unpinned: DESIGN.md 2), so what CAN be measured is measured on many frames
instead of one: for every frame below the HIP encoder's stream (the product's ONE rule: a 15-bit cell answers with the
palette entry nearest to its centre) and the restatement under libsixel's own rule (lookup_mode 0: first hit in raster
order -- src/sixel-canvas.cc:137-145 calls exactly that code) are decoded by the independent decoder and compared

    device  vs source      mean / p99 / max CIE76 of 5x5-averaged pictures (what a diffusion preserves)
    libsixel-rule vs source
    device  vs libsixel-rule
    palette: identical between the two rules by construction (same histogram + median cut) -- asserted

against the numbers stated in BOUNDS, and written to gpurun_out/r5/sixel_delta_e.txt (copied to profiles/r5/).
"""
import os

import numpy as np
import pytest

import timg_amd
from timg_amd import synth
from test_sixel_oracle import _box_blur, _lab

pytestmark = pytest.mark.gpu

BG, PAT = (30, 30, 46, 255), (96, 96, 128, 255)

# Stated tolerances (CIE76 between 5x5 box averages), from the table of the first run (profiles/r5/sixel_delta_e.txt)
# with margin.  Against the SOURCE a 256-colour palette is what limits both rules alike -- saturated photographic content
# leaves 1 % of the neighbourhoods 20..27 away and single neighbourhoods up to 43, for libsixel's rule exactly as for
# the device's (the two columns agree to 0.1 in the mean and 1.5 in the tail on every frame): mean < 4.0 (the figure
# the tests of rounds 1-4 held one frame to), p99 < 30, max < 48, and the device never worse than libsixel's rule by
# more than 0.5 in the mean / 1.0 at p99.  Device against libsixel's rule: mean < 1.6, p99 < 5.0, max < 40.  Frames
# of at most 256 colours are not dithered and decode to the same picture under both rules.
BOUNDS = {"mean": 4.0, "p99": 30.0, "max": 48.0, "between_mean": 1.6, "between_p99": 5.0, "between_max": 40.0}


def delta_e_map(a, b):
    return np.linalg.norm(_lab(_box_blur(a)) - _lab(_box_blur(b)), axis=-1)


def stats(m):
    return float(m.mean()), float(np.percentile(m, 99)), float(m.max())


def gradient(w, h, ncolors):
    """A horizontal ramp of `ncolors` distinct colours (2: two halves)."""
    x = (np.arange(w) * ncolors // w).astype(np.int64)
    fb = np.zeros((h, w, 4), np.uint8)
    fb[..., 0] = (x * 255 // max(1, ncolors - 1))[None, :]
    fb[..., 1] = (255 - x * 200 // max(1, ncolors - 1))[None, :]
    fb[..., 2] = ((x * 7) % 256)[None, :]
    fb[..., 3] = 255
    return fb


def frames(hip, oracle):
    """(name, RGBA frame as the canvas receives it -- already composed where it had alpha)."""
    out = []
    out.append(("S-noise 800x450", synth.noise(800, 450, 1, opaque=True)))
    for seed in (3, 11, 29):
        out.append(("S-photo 800x450 seed %d" % seed, synth.photo(800, 450, seed)))
    a = synth.alpha(800, 450, 5)
    out.append(("S-alpha over solid", oracle.alpha_compose(a, BG)[0]))
    out.append(("S-alpha over checkerboard", oracle.alpha_compose(a, BG, PAT, 18, 18)[0]))
    # BASELINE config 5's frame: 7680x4320 S-alpha -> 800x450 composed over the checkerboard, scaled on the device
    import torch
    src = torch.empty((4320, 7680, 4), dtype=torch.uint8, device="cuda")
    hip.synth_frames("alpha", 7680, 4320, 0, 0, 1, dst=src.data_ptr())
    dst = torch.empty((450, 800, 4), dtype=torch.uint8, device="cuda")
    sc = hip.scaler(7680, 4320, 800, 450)
    hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), 1, timg_amd.Blend.make(BG, PAT, 18, 18))
    hip.sync()
    torch.cuda.synchronize()
    out.append(("c5: 8K S-alpha -> 800x450 over checkerboard", dst.cpu().numpy().copy()))
    sc.close()
    out.append(("2-colour ramp 800x450", gradient(800, 450, 2)))
    out.append(("300-colour ramp 800x450", gradient(800, 450, 300)))
    out.append(("S-photo 333x517 (odd)", synth.photo(333, 517, 7)))
    out.append(("S-noise 97x61 (odd)", synth.noise(97, 61, 2, opaque=True)))
    out.append(("S-photo 1365x96 (widest LDS band)", synth.photo(1365, 96, 13)))
    return out


def test_sixel_delta_e_table(hip, oracle):
    rows, failures = [], []
    worst = {k: 0.0 for k in BOUNDS}
    for name, fb in frames(hip, oracle):
        h, w = fb.shape[:2]
        got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG))[0]
        # (the device's bytes ARE the restatement's under the same rule -- tests/test_gpu_parity.py; asserted here
        # too, so that the table is about the product and nothing else)
        assert got == oracle.sixel_encode(fb, BG, lookup_mode=1), name
        like_libsixel = oracle.sixel_encode(fb, BG, lookup_mode=0)
        dev = oracle.sixel_decode(got)[0][:h, :, :3]
        ref = oracle.sixel_decode(like_libsixel)[0][:h, :, :3]
        pal, dither_off = oracle.sixel_palette(fb)
        src = fb[..., :3]
        s_dev, s_ref, s_between = stats(delta_e_map(dev, src)), stats(delta_e_map(ref, src)), stats(delta_e_map(dev, ref))
        rows.append((name, len(pal), dither_off, s_dev, s_ref, s_between))
        if dither_off:  # <= 256 colours: the palette IS the picture's colours (on the 5:5:5 grid), nothing is diffused
            assert np.array_equal(dev, ref), name
            # (15-bit cells lose 3 bits per channel, the palette's 0..100 per cent units another step of 2.55)
            assert np.abs(dev.astype(int) - src.astype(int)).max() <= 10, name
            continue
        for key, val in (("mean", max(s_dev[0], s_ref[0])), ("p99", max(s_dev[1], s_ref[1])), ("max", max(s_dev[2], s_ref[2])),
                         ("between_mean", s_between[0]), ("between_p99", s_between[1]), ("between_max", s_between[2])):
            worst[key] = max(worst[key], val)
            if val >= BOUNDS[key]:
                failures.append((name, key, round(val, 2), BOUNDS[key]))
        # the device's rule is not further from the source than libsixel's own
        if not (s_dev[0] < s_ref[0] + 0.5 and s_dev[1] < s_ref[1] + 1.0):
            failures.append((name, "device further from the source", s_dev, s_ref))
    lines = ["# CIE76 between 5x5 box averages (mean / p99 / max); device = libtimg_hip.so (cell-centre lookup), "
             "libsixel rule = restatement lookup_mode 0 (first hit in raster order); palettes identical by construction",
             "%-46s %7s  %-22s %-22s %-22s" % ("frame", "colours", "device vs source", "libsixel rule vs source",
                                              "device vs libsixel rule")]
    for name, ncol, off, a, b, c in rows:
        fmt = lambda s: "%5.2f /%6.2f /%6.2f" % s
        lines.append("%-46s %4d%s  %-22s %-22s %-22s" % (name, ncol, " * " if off else "   ", fmt(a), fmt(b), fmt(c)))
    lines.append("(*: at most 256 distinct 15-bit colours -- not dithered, both rules decode to the same picture)")
    lines.append("worst over the dithered frames: " + ", ".join("%s %.2f (bound %.1f)" % (k, worst[k], BOUNDS[k]) for k in BOUNDS))
    text = "\n".join(lines) + "\n"
    print(text)
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out", "r5"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "r5", "sixel_delta_e.txt"), "w") as f:
            f.write(text)
    except OSError:
        pass
    assert not failures, failures
