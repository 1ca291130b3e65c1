"""Single-frame, host-resident path (what the C++ twins do per image): PCIe-inclusive times."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, timg_amd
from timg_amd import synth
hip = timg_amd.TimgHip(0)
src = synth.photo(3840, 2160, seed=1)
blend = timg_amd.Blend.make((30, 30, 46, 255))
sc = hip.scaler(3840, 2160, 800, 450)
dst = np.empty((450, 800, 4), np.uint8)
def t(f, n=10):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
ms_scale = t(lambda: hip.scale_blend(sc, src, dst, 1, blend))
ms_sixel = t(lambda: hip.sixel_encode(dst, 800, 450, pad_blend=blend))
ms_quarter = t(lambda: hip.block_encode(dst, 800, 450, flags=1))
print(f"host->host single 4K frame: scale+blend {ms_scale:.3f} ms ({3840*2160/ms_scale/1e3:.0f} Mpx/s), "
      f"sixel 800x450 {ms_sixel:.3f} ms, quarter 800x450 {ms_quarter:.3f} ms")
# batch of 16 host-resident frames
frames = np.stack([src] * 16)
dst16 = np.empty((16, 450, 800, 4), np.uint8)
ms16 = t(lambda: hip.scale_blend(sc, frames, dst16, 16, blend), 5)
print(f"host->host 16 frames: {ms16:.2f} ms = {16*3840*2160/ms16/1e3:.0f} Mpx/s (PCIe-bound)")
