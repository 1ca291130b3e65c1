"""scratch/r4_dither_sweep.py -- DitherKernel's time as a function of the frame geometry: 64 frames of W x H for a list of
(W, H) (env GEOMS="800x96,400x96,..."), run under rocprofv3 --kernel-trace; scratch/r4_dither_sweep.sh reads the trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, timg_amd
hip = timg_amd.TimgHip(0)
n = int(os.environ.get("N", "64"))
for geom in os.environ.get("GEOMS", "800x96,400x96,800x450,400x450").split(","):
    w, h = (int(v) for v in geom.split("x"))
    src = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    hip.synth_frames("photo", w, h, 0, 0, n, dst=src.data_ptr())
    cap = hip.sixel_max_bytes(w, h)
    out = torch.empty(cap * n, dtype=torch.uint8, device="cuda")
    for _ in range(int(os.environ.get("REPS", "5"))):
        hip.sixel_encode(src.data_ptr(), w, h, n_frames=n, out=out.data_ptr(), out_cap=cap)
    hip.sync()
    print("done", geom, flush=True)
