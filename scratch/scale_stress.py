"""Random geometries, contents and batch sizes: streaming kernels (matrix-core / all-VALU / horizontal-first, chosen
by the plan) against the generic gather kernel (itself pinned to the oracle by the test suite), bit for bit, for
`seconds` (argv[1], default 100).  Prints the instantiations that were exercised."""
import sys, random, time, collections
sys.path.insert(0, '/root/repo')
import torch, timg_amd
hip = timg_amd.TimgHip(0)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t_end = time.time() + seconds
seen = collections.Counter()
bad = cases = 0
while time.time() < t_end:
    sw, sh = random.randint(5, 4200), random.randint(5, 2400)
    if sw * sh > 9_000_000:
        continue
    rx, ry = random.choice([1.0, 1.3, 2.0, 3.7, 4.8, 9.6, 19.2]), random.choice([1.0, 1.3, 2.0, 3.7, 4.8, 9.6, 38.0])
    dw, dh = max(1, int(sw / rx) - random.randint(0, 2)), max(1, int(sh / ry) - random.randint(0, 2))
    n = random.choice([1, 2, 5])
    kind = random.choice(["photo", "alpha", "noise", "mixed"])
    src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
    hip.synth_frames("alpha" if kind == "mixed" else kind, sw, sh, seed=cases, first_frame=0, n_frames=n, dst=src.data_ptr())
    if kind == "mixed":  # opaque except one block: tiles of both channel sets in one frame
        src[..., 3] = 255
        y0, x0 = random.randrange(sh), random.randrange(sw)
        src[:, y0:y0 + max(1, sh // 7), x0:x0 + max(1, sw // 5), 3] = 77
    try:
        sc = hip.scaler(sw, sh, dw, dh)
    except Exception as e:
        print("scaler", (sw, sh, dw, dh), e)
        continue
    info = sc.info()
    cases += 1
    blend = random.choice([None, timg_amd.Blend.make((30, 30, 46, 255)), timg_amd.Blend.make((30, 30, 46, 255), (200, 190, 180, 255), 5, 7)])
    outs = []
    for kernel in ((1, 2) if info["streaming_ok"] else (1,)):
        sc.set_kernel(kernel)
        dst = torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # (torch fills on ITS stream; the library's stream is non-blocking: unordered without this)
        hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
        hip.sync()
        outs.append(dst)
    seen[(info["streaming_ok"], info["vertical_first"], info["matrix_kernel"], info["matrix_overflow_row"], kind)] += 1
    if len(outs) == 2 and not torch.equal(outs[0], outs[1]):
        bad += 1
        print("MISMATCH", (sw, sh, dw, dh), n, kind, "seed", cases - 1, "blend", None if blend is None else (blend.pattern_w, blend.pattern_h),
              info, int((outs[0] != outs[1]).sum()), flush=True)
        if bad <= 6:  # which of the two is wrong, and where?
            sys.path.insert(0, '/root/repo/tests')
            import numpy as np, oracle_lib
            o = oracle_lib.Oracle()
            host = src.cpu().numpy()
            for i in range(n):
                want = o.scale(host[i], dw, dh)
                if blend is not None:
                    pat = (200, 190, 180, 255) if blend.pattern_w else (0, 0, 0, 0)
                    want = o.alpha_compose(want, (30, 30, 46, 255), pat, blend.pattern_w, blend.pattern_h, 0)[0]
                for name, t in (("generic", outs[0]), ("streaming", outs[1])):
                    d = np.any(t[i].cpu().numpy() != want, axis=2)
                    if d.any():
                        ys, xs = np.nonzero(d)
                        print("   frame", i, name, "differs from the oracle in", int(d.sum()), "px, rows", ys.min(), ys.max(),
                              "cols", xs.min(), xs.max(), flush=True)
            # and again: is it reproducible?
            for kernel in (1, 2):
                sc.set_kernel(kernel)
                dst = torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
                hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
                hip.sync()
                print("   rerun kernel", kernel, "equal to its first run:", bool(torch.equal(dst, outs[kernel - 1])), flush=True)
    sc.close()
    del src
    if bad >= 6 and '--all' not in sys.argv:
        break
print("scale stress:", cases, "cases,", bad, "mismatches")
for k, v in sorted(seen.items()):
    print("  streaming=%d vertical_first=%d matrix=%d overflow_row=%d %-6s %d" % (*k, v))
