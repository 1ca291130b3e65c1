#!/usr/bin/env python3
"""scratch/dither_steps.py [kernel-substring] [first-span last-span] -- instructions between consecutive ring waits of a sixel
diffusion kernel (one step each), from the assembly the build keeps; with two span numbers, those spans' instructions."""
import re, sys
path = "timg_amd/csrc/build/sixel_canvas-hip-amdgcn-amd-amdhsa-gfx950.s"
name = sys.argv[1] if len(sys.argv) > 1 else "DitherKernelILb0ELb1ELb1ELb1E"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + name + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
waits = [i for i in range(start, end) if "; ring" in lines[i]]
def body(a, b):
    return [l.strip() for l in lines[a:b] if l.strip() and not l.strip().startswith((";", ".")) and not re.match(r"^[.\w$]+:", l.strip())]
print(name, [len(body(a, b)) for a, b in zip(waits, waits[1:])])
if len(sys.argv) > 3:
    print("\n".join(body(waits[int(sys.argv[2])], waits[int(sys.argv[3])])))
