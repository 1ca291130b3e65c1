import sys, time, threading
sys.path.insert(0, '/root/repo')
import torch, timg_amd
from timg_amd.pipeline import GridPipeline, synth_frames_on_device
n = 64
src = synth_frames_on_device(4, 3840, 2160, "photo", seed=0).repeat(16, 1, 1, 1).contiguous()
blend = timg_amd.Blend.make((30, 30, 46, 255))
def make():
    hip = timg_amd.TimgHip(0)
    return hip, GridPipeline(hip, n, 3840, 2160, 800, 450, "sixel", blend)
pipes = [make() for _ in range(3)]
torch.cuda.synchronize()
def run(p, k):
    for _ in range(k):
        p.step(src)
for hip, p in pipes:
    p.stream.wait_stream(torch.cuda.current_stream())
    run(p, 2)
torch.cuda.synchronize()
for nthreads in (1, 2, 3):
    K = 12
    ths = [threading.Thread(target=run, args=(pipes[i][1], K // nthreads)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} concurrent pipelines: {dt / K * 1e3:.3f} ms per step  ({n * 3840 * 2160 * K / dt / 1e6:.0f} Mpx/s)")
