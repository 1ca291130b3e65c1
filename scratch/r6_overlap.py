"""scratch/r6_overlap.py -- the metric step with the scale call and the sixel chain on TWO streams of one context
(scale of step k + 1 beside the chain of step k; `scaled` and `out` double-buffered), with and without stream
priorities, against the one-stream form bench.py times.  Bytes of the last step compared between the forms."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, timg_amd
from timg_amd.pipeline import synth_frames_on_device

n, iw, ih, ow, oh = 64, 3840, 2160, 800, 450
hip = timg_amd.TimgHip(0)
src = torch.empty((n, ih, iw, 4), dtype=torch.uint8, device="cuda")
hip.synth_frames("photo", iw, ih, seed=0, first_frame=0, n_frames=n, dst=src.data_ptr())
hip.sync()
blend = timg_amd.Blend.make((30, 30, 46, 255))
cap = hip.sixel_max_bytes(ow, oh)
scaler = hip.scaler(iw, ih, ow, oh)
scaled = [torch.empty((n, oh, ow, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [torch.empty((n, cap), dtype=torch.uint8, device="cuda") for _ in range(2)]
jobs = [hip.sixel_job(n) for _ in range(2)]
torch.cuda.synchronize()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)


def run(K, s_scale, s_enc, two):
    """K steps; returns (seconds, lengths of the last step)"""
    lens = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        sl = k & 1
        hip.scale_blend(scaler, src.data_ptr(), scaled[sl].data_ptr(), n, blend, stream=s_scale.cuda_stream)
        if two:
            ev = torch.cuda.Event()
            ev.record(s_scale)
            s_enc.wait_event(ev)
        hip.sixel_encode_async(jobs[sl], scaled[sl].data_ptr(), ow, oh, outs[sl].data_ptr(), cap, n_frames=n,
                               pad_blend=blend, stream=s_enc.cuda_stream)
        if k >= 1:
            lens = hip.sixel_encode_wait(jobs[sl ^ 1], n)  # (frees scaled / out of the other slot for step k + 1)
    lens = hip.sixel_encode_wait(jobs[(K - 1) & 1], n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, lens


def snapshot(K, lens):
    o = outs[(K - 1) & 1]
    return [bytes(o[i, :lens[i]].cpu().numpy().tobytes()) for i in (0, 17, 63)]


K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
forms = [("one stream", None, None)]
forms += [("two streams, equal priority", 0, 0), ("two streams, chain high", 0, -1), ("two streams, scale high", -1, 0)]
if len(sys.argv) > 2:  # one form only (for a kernel trace)
    forms = [forms[int(sys.argv[2])]]
ref = None
for rep in range(2 if len(sys.argv) <= 2 else 1):
    for name, ps, pe in forms:
        if ps is None:
            s1 = torch.cuda.Stream()
            s2, two = s1, False
        else:
            s1, s2, two = torch.cuda.Stream(priority=ps), torch.cuda.Stream(priority=pe), True
        run(6, s1, s2, two)
        dt, lens = run(K, s1, s2, two)
        snap = snapshot(K, lens)
        if ref is None:
            ref = snap
        print("%-32s %.3f ms per step  %.1f Gpx/s  bytes equal to the first form: %s" %
              (name, dt / K * 1e3, n * iw * ih * K / dt / 1e9, snap == ref), flush=True)
