"""scratch/r6_dbg.py -- where do two builds of the library differ?  LIBS=main,<tag>: per-channel mismatch counts and the first few."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, timg_amd, timg_amd.hip as H
here = os.path.dirname(os.path.abspath(H.__file__))
tags = os.environ.get("LIBS", "main,txyz").split(",")
n, sw, sh, dw, dh = 2, 3840, 2160, 800, 450
outs = []
src = None
for t in tags:
    H._lib = None
    os.environ["TIMG_HIP_LIB"] = os.path.join(here, "libtimg_hip.so" if t == "main" else f"libtimg_hip_{t}.so")
    hip = timg_amd.TimgHip(0)
    if src is None:
        src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
        hip.synth_frames("photo", sw, sh, 0, 0, n, dst=src.data_ptr())
        hip.sync()
    sc = hip.scaler(sw, sh, dw, dh)
    d = torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
    hip.scale_blend(sc, src.data_ptr(), d.data_ptr(), n, None)
    torch.cuda.synchronize()
    outs.append(d.cpu())
a, b = outs[0], outs[1]
ne = a != b
print("mismatches per channel:", [int(ne[..., c].sum()) for c in range(4)])
idx = ne.nonzero()[:12]
for i in idx.tolist():
    f, y, x, c = i
    print(i, int(a[f, y, x, c]), int(b[f, y, x, c]))
print("columns with mismatches (first 40):", sorted(set(ne.nonzero()[:, 2].tolist()))[:40])
print("max abs diff:", int((a.int() - b.int()).abs().max()))
