"""scratch/r6_cumask2.py [steps [reserved_per_xcd]] -- r6_cumask.py with the chains of consecutive steps on TWO contexts (each
its own sixel scratch, its own high-priority encode stream): chain k + 1 may start while chain k is still running; the host
is two steps ahead (it waits for step k - 2 before it enqueues step k).  TIMG_HIP_DITHER_PARTS from the environment."""
import sys, time, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, timg_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reserved = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n, iw, ih, ow, oh = 64, 3840, 2160, 800, 450
hips = [timg_amd.TimgHip(0), timg_amd.TimgHip(0)]
hip = hips[0]
src = torch.empty((n, ih, iw, 4), dtype=torch.uint8, device="cuda")
hip.synth_frames("photo", iw, ih, seed=0, first_frame=0, n_frames=n, dst=src.data_ptr())
hip.sync()
blend = timg_amd.Blend.make((30, 30, 46, 255))
cap = hip.sixel_max_bytes(ow, oh)
scaler = hip.scaler(iw, ih, ow, oh)
scaled = [torch.empty((n, oh, ow, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [torch.empty((n, cap), dtype=torch.uint8, device="cuda") for _ in range(2)]
jobs = [hips[i].sixel_job(n) for i in range(2)]
torch.cuda.synchronize()
s_scale = hip.stream_create(reserved_cus_per_xcd=reserved) if reserved else hip.stream_create()
s_enc = [hips[i].stream_create(high_priority=True) for i in range(2)]


def run(K, two_chains):
    lens = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        sl = k & 1
        c = sl if two_chains else 0
        if two_chains and k >= 2:
            lens = hips[c].sixel_encode_wait(jobs[sl], n)
        hip.scale_blend(scaler, src.data_ptr(), scaled[sl].data_ptr(), n, blend, stream=s_scale)
        hip.stream_wait_stream(s_enc[c], s_scale)
        hips[c].sixel_encode_async(jobs[sl] if two_chains else jobs0[sl], scaled[sl].data_ptr(), ow, oh, outs[sl].data_ptr(), cap,
                                   n_frames=n, pad_blend=blend, stream=s_enc[c])
        if not two_chains and k >= 1:
            lens = hip.sixel_encode_wait(jobs0[sl ^ 1], n)
    if two_chains:
        if K >= 2:
            hips[(K - 2) & 1].sixel_encode_wait(jobs[(K - 2) & 1], n)
        lens = hips[(K - 1) & 1].sixel_encode_wait(jobs[(K - 1) & 1], n)
    else:
        lens = hip.sixel_encode_wait(jobs0[(K - 1) & 1], n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, lens


jobs0 = [hip.sixel_job(n) for _ in range(2)]
ref = None
for rep in range(2):
    for name, two in (("one chain at a time", False), ("two chains (two contexts)", True)):
        run(6, two)
        dt, lens = run(K, two)
        o = outs[(K - 1) & 1]
        snap = [bytes(o[i, :lens[i]].cpu().numpy().tobytes()) for i in (0, 17, 63)]
        if ref is None:
            ref = snap
        print("reserved %2d parts %s  %-28s %.3f ms per step  %.1f Gpx/s  bytes equal: %s" %
              (reserved, os.environ.get("TIMG_HIP_DITHER_PARTS", "-"), name, dt / K * 1e3, n * iw * ih * K / dt / 1e9, snap == ref), flush=True)
