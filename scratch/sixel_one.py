# one small and one 800x450 sixel encode against the oracle; run under `ulimit -c 0; timeout 60`
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
o = oracle_lib.Oracle()
hip = timg_amd.TimgHip(0)
for (w, h, sw, sh) in ((64, 36, 320, 200), (800, 450, 3840, 2160)):
    fb = o.scale(synth.make("photo", sw, sh, seed=0), w, h)
    print("encode", w, h, flush=True)
    got = hip.sixel_encode(fb, w, h)[0]
    want = o.sixel_encode(fb, has_getter=False, lookup_mode=1)
    if got == want:
        print("  identical,", len(got), "bytes", flush=True)
    else:
        n = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        print("  MISMATCH at", n, "of", len(got), len(want), got[max(0, n - 30):n + 30], want[max(0, n - 30):n + 30], flush=True)
