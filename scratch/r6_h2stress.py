"""scratch/r6_h2stress.py [seconds] -- random horizontal-first geometries around the two-column kernel's range (ratios 8 to
10.6 (the range of the two-column kernel: 17 to 40 taps), odd widths, windows near the row buffer's bound, frames with opaque / alpha / fully transparent regions, composed and
not) against the restatement, byte for byte; counts how many ran on ScaleStreamH2Kernel."""
import os, sys, time, random
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import timg_amd, oracle_lib
from timg_amd import synth
hip, oracle = timg_amd.TimgHip(0), oracle_lib.Oracle()
rng = random.Random(int(os.environ.get("SEED", "6")))
t_end = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 60)
n = n_h2 = n_hf = 0
BG, PAT = (30, 30, 46, 255), (96, 96, 128, 255)
while time.time() < t_end:
    dw = rng.randint(40, 1100)
    ratio = rng.uniform(4.2, 10.6)
    sw = max(dw + 1, int(dw * ratio) + rng.randint(-3, 3))
    dh = rng.randint(6, 60)
    sh = max(dh, int(dh * (rng.uniform(8.0, 11.0) if ratio > 8 else rng.uniform(1.0, 12.0))))
    info = oracle.plan_info(sw, sh, dw, dh)
    kind = rng.choice(["photo", "alpha", "mixed"])
    src = synth.photo(sw, sh, seed=n) if kind == "photo" else synth.alpha(sw, sh, seed=n)
    if kind == "mixed":
        src[:, : sw // 3] = synth.photo(sw, sh, seed=n + 1000)[:, : sw // 3]
        src[sh // 4: sh // 2, sw // 2: sw // 2 + sw // 5, 3] = 0
    fmt = rng.randint(0, 1)
    sc = hip.scaler(sw, sh, dw, dh, in_fmt=fmt)
    si = sc.info()
    want = oracle.scale(src, dw, dh, in_fmt=fmt)
    got = np.empty((dh, dw, 4), np.uint8)
    blend = None if rng.random() < 0.5 else timg_amd.Blend.make(BG, PAT, rng.randint(1, 20), rng.randint(1, 20))
    hip.scale_blend(sc, src, got, 1, blend)
    if blend is not None:
        want = oracle.alpha_compose(want, BG, PAT, blend.pattern_w, blend.pattern_h)[0]
    if not np.array_equal(got, want):
        bad = np.argwhere((got != want).any(axis=2))
        print("MISMATCH", sw, sh, dw, dh, kind, fmt, si, "first", bad[0], "count", len(bad), flush=True)
        sys.exit(1)
    sc.close()
    n += 1
    n_hf += 1 - si["vertical_first"]
    n_h2 += si["two_column_kernel"]
print("h2 stress: %d geometries byte-exact, %d horizontal-first, %d on the two-column kernel" % (n, n_hf, n_h2))
