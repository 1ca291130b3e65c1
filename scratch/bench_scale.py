import sys, os, time
sys.path.insert(0, '/root/repo')
import torch, timg_amd
from timg_amd.pipeline import synth_frames_on_device
n = int(os.environ.get("N", "64"))
kind = os.environ.get("KIND", "photo")
dw, dh = int(os.environ.get("DW", "800")), int(os.environ.get("DH", "450"))
sw, sh = int(os.environ.get("SW", "3840")), int(os.environ.get("SH", "2160"))
hip = timg_amd.TimgHip(0)
src = synth_frames_on_device(min(4, n), sw, sh, kind, seed=0)
src = src.repeat((n + 3) // 4, 1, 1, 1)[:n].contiguous() if n > 4 else src[:n].contiguous()
dst = torch.empty((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
sc = hip.scaler(sw, sh, dw, dh)
print(sc.info())
blend = timg_amd.Blend.make((30, 30, 46, 255))
st = torch.cuda.Stream()
for kernel in [int(k) for k in os.environ.get("KERNELS", "2").split(",")]:
    sc.set_kernel(kernel)
    for _ in range(2):
        hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record(st)
    for _ in range(reps):
        hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = sc.algorithmic_bytes() * n / 1e9
    print(f"kernel {kernel} band={os.environ.get('TIMG_HIP_BAND_ROWS','-')} {kind} {sw}x{sh}->{dw}x{dh} n={n}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s  ({gb/ms*1e3/8000*100:.1f}% of 8TB/s)")
