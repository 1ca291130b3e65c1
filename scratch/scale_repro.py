"""Which kernel is wrong?  Geometries that scale_stress.py reported: generic (1) and streaming (2) kernels, twice each,
against the oracle, per frame, with the bounding box of the differing pixels."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, timg_amd, oracle_lib
from timg_amd import synth
o = oracle_lib.Oracle()
hip = timg_amd.TimgHip(0)
cases = [(2474, 2269, 2474, 2268, "photo"), (2159, 1541, 2157, 1541, "photo"), (3284, 1151, 886, 1151, "photo"),
         (2705, 769, 2705, 589, "alpha"), (1427, 1393, 1427, 1392, "photo"), (3288, 2395, 3287, 2395, "photo")]
for sw, sh, dw, dh, kind in cases:
    for n in (1, 5):
        src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
        hip.synth_frames(kind, sw, sh, seed=3, first_frame=0, n_frames=n, dst=src.data_ptr())
        host = src.cpu().numpy()
        want = np.stack([o.scale(host[i], dw, dh) for i in range(n)])
        sc = hip.scaler(sw, sh, dw, dh)
        for blend in (None, timg_amd.Blend.make((30, 30, 46, 255))):
            w2 = want if blend is None else np.stack([o.alpha_compose(want[i], (30, 30, 46, 255), (0, 0, 0, 0), 0, 0, 0)[0] for i in range(n)])
            for kernel in (1, 2, 2):
                sc.set_kernel(kernel)
                dst = torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
                hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
                hip.sync()
                got = dst.cpu().numpy()
                for i in range(n):
                    d = np.any(got[i] != w2[i], axis=2)
                    if d.any():
                        ys, xs = np.nonzero(d)
                        print((sw, sh, dw, dh), kind, "n", n, "blend", blend is not None, "kernel", kernel, "frame", i,
                              "px", int(d.sum()), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), flush=True)
        sc.close()
print("done")
