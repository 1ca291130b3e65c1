#!/usr/bin/env python3
"""scratch/isa_mix.py [file.s ...] -- instruction mix of the hot loops, from the assembly the build keeps
(timg_amd/csrc/build/*-gfx950.s): for every kernel with a prefetch ring, the instructions between two consecutive
"; ring" waits (= one source row of a scale kernel, one step of the sixel diffusion), by issue class.  The kernels of
this repository are bound by what a wave issues, so this is the number a change should move BEFORE a GPU minute is
spent on it (the rows of DESIGN.md 4.1 / 10 were counted by hand from the same files).

The span between two waits contains everything the compiler placed there, taken or not (completion paths, the row-end
shuffle): read `branches` with it -- a span with many is not all executed per row."""
import collections
import re
import sys

CLASSES = [
    ("mfma", re.compile(r"^v_mfma")),
    ("valu_pk", re.compile(r"^v_pk_")),
    ("valu_dpp", re.compile(r"^v_\w+_dpp|^v_mov_b32_dpp")),
    ("valu_cvt", re.compile(r"^v_cvt")),
    ("valu_xlane", re.compile(r"^v_(readlane|writelane|readfirstlane|perm)")),
    ("valu", re.compile(r"^v_")),
    ("lds_read", re.compile(r"^ds_read")),
    ("lds_write", re.compile(r"^ds_write|^ds_add|^ds_\w+_rtn")),
    ("vmem", re.compile(r"^(global|flat|buffer)_")),
    ("branches", re.compile(r"^s_cbranch|^s_branch")),
    ("waits", re.compile(r"^s_waitcnt|^s_nop|^s_barrier|^s_sleep")),
    ("salu", re.compile(r"^s_")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(line)
        if "s_endpgm" in line:
            yield name, body
            name = None


def short(mangled):
    m = re.search(r"\d+(ScaleStream\w*Kernel|DitherKernel)I(.*?)EEv", mangled)
    if not m:
        return mangled[:60]
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    return "%s<%s>" % (m.group(1), ",".join(args))


def main(paths):
    for path in paths:
        for name, body in kernels(path):
            marks = [i for i, l in enumerate(body) if "; ring" in l and "ring all" not in l]
            if len(marks) < 2:
                continue
            span = body[marks[0]:marks[1]]
            mix = collections.Counter()
            for l in span:
                l = l.strip()
                if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
                    continue
                mix[classify(l.split()[0])] += 1
            total = sum(v for k, v in mix.items())
            cols = " ".join("%s %d" % (k, mix[k]) for k, _ in CLASSES if mix[k])
            print("%-34s %4d instr between two ring waits: %s" % (short(name), total, cols))


if __name__ == "__main__":
    import glob
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1:] or sorted(glob.glob(os.path.join(here, "..", "timg_amd", "csrc", "build", "*-gfx950.s"))))
