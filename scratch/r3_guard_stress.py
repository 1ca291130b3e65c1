import sys, os
sys.path.insert(0, ".")
import numpy as np, timg_amd
hip = timg_amd.TimgHip(0)
rng = np.random.default_rng(1)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
big = rng.integers(0, 256, 40_000_000, dtype=np.uint8)
for i in range(N):
    n = int(rng.integers(1, 3_000_000)) if i % 10 else int(rng.integers(3_000_000, 36_000_000))
    off = int(rng.integers(0, 4096))
    a = big[off:off + n]
    p = hip.upload(a)
    b = hip.download(p, n)
    if not np.array_equal(a, b):
        d = np.nonzero(a != b)[0]
        print(f"iter {i}: n={n} off={off} ptr={p:#x} mismatches={len(d)} first={d[0]} last={d[-1]} vals={b[d[:8]]} again_equal={np.array_equal(hip.download(p, n), a)}", flush=True)
        bad += 1
        if bad > 5: break
    try:
        hip.free(p)
    except Exception as e:
        print("free:", e)
print("stress done, bad =", bad, "of", N)
