"""Times the scale kernel of SEVERAL builds of the library (scratch/build_variant.sh) interleaved in ONE process -- boxes and
clock states differ by more than the variants do -- and compares every build's output bytes with the first one's.

Environment: LIBS="tag,tag,..." (timg_amd/libtimg_hip_<tag>.so; "main" = libtimg_hip.so), N, SW/SH, DW/DH, KIND,
REPS, ROUNDS, BLEND=0/1.  A build with -DTIMG_M_TRACE prints where its waves' time goes."""
import ctypes, os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import timg_amd
import timg_amd.hip as H

n = int(os.environ.get("N", "64"))
kind = os.environ.get("KIND", "photo")
dw, dh = int(os.environ.get("DW", "800")), int(os.environ.get("DH", "450"))
sw, sh = int(os.environ.get("SW", "3840")), int(os.environ.get("SH", "2160"))
reps, rounds = int(os.environ.get("REPS", "20")), int(os.environ.get("ROUNDS", "7"))
tags = os.environ.get("LIBS", "main").split(",")
here = os.path.dirname(os.path.abspath(H.__file__))


def open_lib(tag):
    H._lib = None
    os.environ["TIMG_HIP_LIB"] = os.path.join(here, "libtimg_hip.so" if tag == "main" else f"libtimg_hip_{tag}.so")
    return timg_amd.TimgHip(0)


hips = [open_lib(t) for t in tags]
src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
hips[0].synth_frames(kind, sw, sh, 0, 0, n, dst=src.data_ptr())
hips[0].sync()
blend = timg_amd.Blend.make((30, 30, 46, 255)) if os.environ.get("BLEND", "1") == "1" else None
st = torch.cuda.Stream()
scalers = [h.scaler(sw, sh, dw, dh) for h in hips]
print(scalers[0].info(), flush=True)
dsts = [torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in hips]
for h, sc, d in zip(hips, scalers, dsts):
    h.scale_blend(sc, src.data_ptr(), d.data_ptr(), n, blend, stream=st.cuda_stream)
torch.cuda.synchronize()
for t, d in zip(tags[1:], dsts[1:]):
    same = bool(torch.equal(d, dsts[0]))
    print(f"bytes {t} == {tags[0]}: {same}" + ("" if same else f"  ({int((d != dsts[0]).sum())} bytes differ)"), flush=True)

t_end = time.time() + float(os.environ.get("WARM_S", "0.3"))
while time.time() < t_end:
    for h, sc, d in zip(hips, scalers, dsts):
        h.scale_blend(sc, src.data_ptr(), d.data_ptr(), n, blend, stream=st.cuda_stream)
    torch.cuda.synchronize()
times = [[] for _ in hips]
for _ in range(rounds):
    for i, (h, sc, d) in enumerate(zip(hips, scalers, dsts)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            h.scale_blend(sc, src.data_ptr(), d.data_ptr(), n, blend, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / reps)
for t, sc, tm in zip(tags, scalers, times):
    ms = statistics.median(tm)
    gb = sc.algorithmic_bytes() * n / 1e9
    print(f"{t:10s} {kind} {sw}x{sh}->{dw}x{dh} n={n}: {ms:.4f} ms (min {min(tm):.4f} max {max(tm):.4f})  "
          f"{gb / ms * 1e3:.0f} GB/s  ({gb / ms * 1e3 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)

for t, h, sc, d in zip(tags, hips, scalers, dsts):
    if not hasattr(h.L, "timg_hip_debug_mtrace"):
        continue
    try:
        fn = h.L.timg_hip_debug_mtrace
    except AttributeError:
        continue
    out = (ctypes.c_ulonglong * 8)()
    fn(out, 1)
    h.scale_blend(sc, src.data_ptr(), d.data_ptr(), n, blend, stream=st.cuda_stream)
    torch.cuda.synchronize()
    fn(out, 0)
    v = list(out)
    waves = max(v[5], 1)
    names = ["wait row", "vertical", "stage+bar", "horizontal", "bar2", "waves", "prologue", "kernel"]
    print(f"trace {t}: " + ", ".join(f"{nm} {x / waves:.0f}" for nm, x in zip(names, v) if nm != "waves")
          + f" ticks per wave ({waves} waves)", flush=True)
