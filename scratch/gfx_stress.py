"""Random sizes / contents: png, kitty, iTerm2 at --compress=0 (RGBA and RGB) and sixel in first-hit mode against the
oracle, byte for byte, for `seconds` (argv[1], default 60)."""
import sys, random, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
o = oracle_lib.Oracle()
hip = timg_amd.TimgHip(0)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
t_end = time.time() + seconds
bad = cases = 0
while time.time() < t_end:
    w, h = random.choice([(random.randint(1, 60), random.randint(1, 40)), (random.randint(1, 900), random.randint(1, 500)),
                          (random.randint(3000, 6000), random.randint(1, 12))])
    kind = random.choice(["photo", "alpha", "noise"])
    fb = synth.make(kind, w, h, seed=cases)
    rgb24 = random.random() < 0.5
    iid = random.choice([1, 77, 4_000_000_123])
    ok = hip.gfx_encode("png", fb, w, h, rgb24=rgb24)[0] == o.png_encode(fb, not rgb24)
    ok = ok and hip.gfx_encode("kitty", fb, w, h, rgb24=rgb24, image_ids=[iid])[0] == o.kitty_encode(fb, iid, not rgb24)
    ok = ok and hip.gfx_encode("iterm2", fb, w, h, rgb24=rgb24)[0] == o.iterm2_encode(fb, not rgb24)
    if w <= 300 and h <= 200 and cases % 3 == 0:
        got = hip.sixel_encode_first_hit(fb, w, h, out_cap=hip.sixel_max_bytes(w, h) * 4)[0]  # (the test-only debug library)
        ok = ok and got == o.sixel_encode(fb, has_getter=False, lookup_mode=0)
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", w, h, kind, rgb24, flush=True)
print("gfx / first-hit stress:", cases, "cases,", bad, "mismatches")
