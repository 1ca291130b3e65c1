"""How does DitherKernel's time per step depend on the number of waves that share a CU?
64 frames of 800 x H for several H (one workgroup = one frame = one CU; one wave per 32 rows):
run under rocprofv3 --kernel-trace --stats and read DitherKernel's average."""
import sys, os
sys.path.insert(0, '/root/repo')
import torch, timg_amd
hip = timg_amd.TimgHip(0)
w = 800
n = int(os.environ.get("N", "64"))
for h in [int(x) for x in os.environ.get("HS", "96,120,216,450").split(",")]:
    src = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    hip.synth_frames("photo", w, h, 0, 0, n, dst=src.data_ptr())
    cap = hip.sixel_max_bytes(w, h)
    out = torch.empty(cap * n, dtype=torch.uint8, device="cuda")
    for _ in range(int(os.environ.get("REPS", "5"))):
        hip.sixel_encode(src.data_ptr(), w, h, n_frames=n, out=out.data_ptr(), out_cap=cap)
    hip.sync()
    print("done", h, flush=True)
