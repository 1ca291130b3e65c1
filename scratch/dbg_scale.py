import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
hip = timg_amd.TimgHip(0); orc = oracle_lib.Oracle()
for kind, dw, dh in [("alpha", 800, 450), ("alpha", 1280, 720)]:
    src = synth.make(kind, 3840, 2160, seed=9)
    want = orc.scale(src, dw, dh)
    sc = hip.scaler(3840, 2160, dw, dh)
    for kernel in (2, 3, 4, 1):
        sc.set_kernel(kernel)
        got = np.empty((dh, dw, 4), np.uint8)
        hip.scale_blend(sc, src, got)
        bad = np.argwhere((got != want).any(axis=2))
        print(kind, dw, dh, "kernel", kernel, "bad px", len(bad))
        if len(bad):
            ys = np.unique(bad[:, 0]); xs = np.unique(bad[:, 1])
            print("  rows", ys[:20], "... cols", xs[:20], xs[-5:])
            for y, x in bad[:5]:
                print("   ", y, x, got[y, x], want[y, x])
