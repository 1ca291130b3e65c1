#!/usr/bin/env python3
"""scratch/dither_isa.py [kernel-substring] -- instructions between two ring waits of the sixel diffusion's unrolled
body, from the assembly the build keeps (timg_amd/csrc/build/sixel_canvas-hip-amdgcn-amd-amdhsa-gfx950.s)."""
import re, sys
path = "timg_amd/csrc/build/sixel_canvas-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[1] if len(sys.argv) > 1 else "DitherKernelILb0ELb1EE"
show = len(sys.argv) > 2
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + want + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
waits = [i for i in range(start, end) if "vmcnt(14) ; ring" in lines[i]]
for a, b in zip(waits, waits[1:] ):
    body = [l.strip() for l in lines[a:b] if l.strip() and not l.strip().startswith((";", ".", "s_waitcnt vmcnt(14)")) and not re.match(r"^[.\w$]+:", l.strip())]
    kinds = {}
    for ins in body:
        op = ins.split()[0]
        k = "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith("global_") else "valu"
        kinds[k] = kinds.get(k, 0) + 1
    print(len(body), kinds)
    if show:
        print("\n".join(body)); show = False
