"""Random sixel geometries / contents against the oracle (byte equality), plus run-to-run
determinism of a batch: `ulimit -c 0; timeout 80 python scratch/sixel_stress.py [seconds [seed [widest]]]`
(widest: the last width class reaches up to it, default 2200; 4095 = the device's limit)."""
import sys, time, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
o = oracle_lib.Oracle()
hip = timg_amd.TimgHip(0)
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
widest = int(sys.argv[3]) if len(sys.argv) > 3 else 2200
BG, PAT = (30, 30, 46, 255), (200, 190, 180, 255)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = random.choice(["photo", "noise", "alpha", "photo"])
    w = random.choice([random.randint(1, 40), random.randint(41, 400), random.randint(401, 1365), random.randint(1366, widest)])
    h = random.choice([random.randint(1, 30), random.randint(31, 200), random.randint(201, 700)])
    if w * h > 500_000: h = max(1, 500_000 // w)
    fb = synth.make(kind, w, h, seed=random.randint(0, 1 << 30))
    pw, ph = random.randint(1, 9), random.randint(1, 9)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, pw, ph),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = o.sixel_encode(fb, BG, PAT, pw, ph, lookup_mode=1)
    n += 1
    if got != want:
        bad += 1
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        print("MISMATCH", kind, w, h, pw, ph, len(got), len(want), "first diff at", k, flush=True)
print("sixel stress:", n, "cases,", bad, "mismatches", flush=True)
# determinism of a batch under repetition
import torch
fbs = torch.randint(0, 256, (8, 300, 500, 4), dtype=torch.uint8, device="cuda"); fbs[..., 3] = 255
ref = None
for it in range(6):
    out = torch.empty((8, hip.sixel_max_bytes(500, 300) * 2), dtype=torch.uint8, device="cuda")
    lens = hip.sixel_encode(fbs.data_ptr(), 500, 300, n_frames=8, out=out.data_ptr(), out_cap=out.shape[1])
    sig = [hash(out[i, :lens[i]].cpu().numpy().tobytes()) for i in range(8)]
    if ref is None: ref = sig
    elif sig != ref:
        bad += 1; print("NONDETERMINISM at", it, flush=True)
print("stress done, problems:", bad, flush=True)
