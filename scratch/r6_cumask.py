"""scratch/r6_cumask.py [steps [reserved_per_xcd [layout]]] -- the metric step with the scale call on a stream whose CU mask
leaves `reserved_per_xcd` CUs of every XCD to the sixel chain (hipExtStreamCreateWithCUMask), the scale of step k + 1
gated behind the END of chain k - 1, so that it runs beside Hist / MedianCut / BuildLut of step k (64-workgroup,
latency-bound kernels that fit the reserved CUs) and the chain's chip-wide kernels run alone.  Against the one-stream form.
layout: 0 = mask bit n is CU n / 8 of XCD n % 8 (interleaved), 1 = bit n is CU n % 32 of XCD n / 32."""
import sys, time, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, timg_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reserved = int(sys.argv[2]) if len(sys.argv) > 2 else 4
layout = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, iw, ih, ow, oh = 64, 3840, 2160, 800, 450
hip = timg_amd.TimgHip(0)
rt = ctypes.CDLL("libamdhip64.so")
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


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


def masked_stream(reserved_per_xcd):
    bits = [1] * 256
    for b in range(256):
        cu = b // 8 if layout == 0 else b % 32
        if cu >= 32 - reserved_per_xcd:
            bits[b] = 0
    words = (ctypes.c_uint32 * 8)(*[sum(bits[32 * w + i] << i for i in range(32)) for w in range(8)])
    s = ctypes.c_void_p()
    check(rt.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words), "hipExtStreamCreateWithCUMask")
    return s


def new_event():
    e = ctypes.c_void_p()
    check(rt.hipEventCreateWithFlags(ctypes.byref(e), 2), "hipEventCreateWithFlags")  # hipEventDisableTiming
    return e


def run(K, s_scale, s_enc, two, gate):
    ev_scale, ev_chain = [new_event() for _ in range(2)], [new_event() for _ in range(2)]
    lens = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        sl = k & 1
        if two and gate and k >= 2:
            check(rt.hipStreamWaitEvent(s_scale, ev_chain[sl], 0), "wait chain")  # (chain k - 2 ... see below)
        hip.scale_blend(scaler, src.data_ptr(), scaled[sl].data_ptr(), n, blend, stream=s_scale.value)
        if two:
            check(rt.hipEventRecord(ev_scale[sl], s_scale), "record scale")
            check(rt.hipStreamWaitEvent(s_enc, ev_scale[sl], 0), "wait scale")
        hip.sixel_encode_async(jobs[sl], scaled[sl].data_ptr(), ow, oh, outs[sl].data_ptr(), cap, n_frames=n,
                               pad_blend=blend, stream=s_enc.value)
        if two:
            check(rt.hipEventRecord(ev_chain[sl], s_enc), "record chain")
        if k >= 1:
            lens = hip.sixel_encode_wait(jobs[sl ^ 1], n)
    lens = hip.sixel_encode_wait(jobs[(K - 1) & 1], n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, lens


def snapshot(K, lens):
    o = outs[(K - 1) & 1]
    return [bytes(o[i, :lens[i]].cpu().numpy().tobytes()) for i in (0, 17, 63)]


# (the host is one step ahead: scale k + 1 is enqueued while chain k runs; its gate is the end of chain k - 1, whose
# event sits in slot (k + 1) & 1 -- the slot the loop above waits on)
plain = ctypes.c_void_p(torch.cuda.Stream().cuda_stream)
enc = ctypes.c_void_p(torch.cuda.Stream(priority=-1).cuda_stream)
forms = [("one stream", plain, plain, False, False),
         ("scale on %d CUs, ungated" % (256 - 8 * reserved), masked_stream(reserved), enc, True, False),
         ("scale on %d CUs, gated behind chain k-1" % (256 - 8 * reserved), masked_stream(reserved), enc, True, True),
         ("scale on 256 CUs, gated behind chain k-1", ctypes.c_void_p(torch.cuda.Stream().cuda_stream), enc, True, True)]
if len(sys.argv) > 4:
    forms = [forms[int(sys.argv[4])]]
ref = None
for rep in range(2 if len(sys.argv) <= 4 else 1):
    for name, s1, s2, two, gate in forms:
        run(6, s1, s2, two, gate)
        dt, lens = run(K, s1, s2, two, gate)
        snap = snapshot(K, lens)
        if ref is None:
            ref = snap
        print("%-44s %.3f ms per step  %.1f Gpx/s  bytes equal to the first form: %s" %
              (name, dt / K * 1e3, n * iw * ih * K / dt / 1e9, snap == ref), flush=True)
