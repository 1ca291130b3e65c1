"""Times the scale kernels alone on device-resident batches.

Environment: N, SW/SH, DW/DH, KIND, KERNELS (forced kernel ids), TIMG_HIP_* tuning variables, and
VARIANTS = "name:ENV=VAL,ENV=VAL;name2:..." -- scalers created under different tuning environments,
timed INTERLEAVED in one process (rounds of REPS launches each, median per variant): the only way to
compare variants that differ by a few percent, since boxes and clock states differ by more."""
import sys, os, time, statistics
sys.path.insert(0, '/root/repo')
import torch, timg_amd
n = int(os.environ.get("N", "64"))
kind = os.environ.get("KIND", "photo")
dw, dh = int(os.environ.get("DW", "800")), int(os.environ.get("DH", "450"))
sw, sh = int(os.environ.get("SW", "3840")), int(os.environ.get("SH", "2160"))
reps, rounds = int(os.environ.get("REPS", "20")), int(os.environ.get("ROUNDS", "7"))
hip = timg_amd.TimgHip(0)
src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
hip.synth_frames(kind, sw, sh, 0, 0, n, dst=src.data_ptr())
hip.sync()
dst = torch.empty((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
blend = timg_amd.Blend.make((30, 30, 46, 255))
st = torch.cuda.Stream()
variants = []
spec = os.environ.get("VARIANTS", "")
if spec:
    for item in spec.split(";"):
        name, _, envs = item.partition(":")
        variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
else:
    variants.append(("default", {}))
scalers = []
for name, env in variants:
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    sc = hip.scaler(sw, sh, dw, dh)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    scalers.append(sc)
print(scalers[0].info())
for kernel in [int(k) for k in os.environ.get("KERNELS", "2").split(",")]:
    for sc in scalers:
        sc.set_kernel(kernel)
    t_end = time.time() + float(os.environ.get("WARM_S", "0.3"))   # warm-up to a steady clock
    while time.time() < t_end:
        for sc in scalers:
            hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend, stream=st.cuda_stream)
        torch.cuda.synchronize()
    times = [[] for _ in scalers]
    for _ in range(rounds):
        for i, sc in enumerate(scalers):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend, stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / reps)
    for (name, _), sc, t in zip(variants, scalers, times):
        ms = statistics.median(t)
        gb = sc.algorithmic_bytes() * n / 1e9
        print(f"kernel {kernel} {name} band={os.environ.get('TIMG_HIP_BAND_ROWS','-')} {kind} {sw}x{sh}->{dw}x{dh} n={n}: "
              f"{ms:.3f} ms (min {min(t):.3f} max {max(t):.3f})  {gb/ms*1e3:.0f} GB/s  ({gb/ms*1e3/8000*100:.1f}% of 8TB/s)")
