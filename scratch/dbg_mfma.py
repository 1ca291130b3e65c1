import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
hip = timg_amd.TimgHip(0); orc = oracle_lib.Oracle()
cases = [("photo", 160, 90, 40, 23), ("alpha", 160, 90, 40, 23), ("photo", 640, 480, 200, 113), ("photo", 3840, 2160, 800, 450), ("alpha", 3840, 2160, 800, 450),
         ("noise", 1920, 1080, 400, 225), ("photo", 1000, 700, 333, 99)]
for kind, sw, sh, dw, dh in cases:
    src = synth.make(kind, sw, sh, seed=9)
    want = orc.scale(src, dw, dh)
    sc = hip.scaler(sw, sh, dw, dh)
    print(kind, sw, sh, dw, dh, sc.info())
    for kernel in (2,):
        try:
            sc.set_kernel(kernel)
        except Exception as e:
            print("  kernel", kernel, "refused", e); continue
        got = np.empty((dh, dw, 4), np.uint8)
        hip.scale_blend(sc, src, got)
        bad = np.argwhere((got != want).any(axis=2))
        print("  kernel", kernel, "bad px", len(bad))
        if len(bad):
            ys = np.unique(bad[:, 0]); xs = np.unique(bad[:, 1])
            print("    rows", ys[:30], "... cols", xs[:20], xs[-5:])
            for y, x in bad[:5]:
                print("     ", y, x, got[y, x], want[y, x])
