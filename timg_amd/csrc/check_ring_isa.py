#!/usr/bin/env python3
"""check_ring_isa.py <device assembly of scale_stream.hip or sixel_canvas.hip>

ScaleStreamMKernel keeps source rows in flight in a ring of register sets that are loaded by inline
asm (`global_load_dwordx4`; round 6: four typed `buffer_load_format_xyzw` a row) and released by an inline-asm `s_waitcnt vmcnt(n) ; ring v[a:b]`; the sixel
DitherKernel does the same with source pixels (`global_load_dword`, `; ring vN`).  The
compiler does not know that such a load completes later: this script proves, on the generated code,
that on no path between a set's load and its wait an instruction outside the asm statements names one
of its registers (a copy for a tied operand, a live-range split, a back-edge copy into another register
set would read data that has not arrived, or be overwritten when it does).

Method: basic blocks from labels and branches, forward may-analysis of "sets in flight" (union over
predecessors, to a fixed point), then every instruction is checked against the sets in flight before it.

Second property, same assembly: the matrix kernel writes the A operand of its v_mfma products with inline-asm
`v_writelane_b32` (four weights into lanes 0-3) and pads the VALU-write -> MFMA-read hazard by hand (`s_nop 3`): the
compiler does not see that write, so it inserts nothing.  Missing wait states showed as wrong red channels in every
sixth output row (round 2).  Here: on every path from a `v_writelane_b32 vN` to a `v_mfma*` that reads vN there are at
least MFMA_WAIT_STATES wait states (an instruction counts one, `s_nop n` counts n + 1) -- a forward analysis of
"wait states since the last v_writelane" per register, minimum over predecessors.

Exit status 0: every kernel that uses the ring is clean.  1: a violation (printed)."""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
# Accumulation registers (the matrix kernel's ring since round 5 lives in a[..], named ONLY inside asm statements): they are
# tracked in the same sets as the vector registers, numbered from AGPR_BASE.
AGPR_BASE = 1000
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
LOAD = re.compile(r"(?:global_load_dwordx[24]|buffer_load_dwordx4|buffer_load_format_xyzw?) v\[(\d+):(\d+)\]")  # (x2: the sixel diffusion's pixel pairs; buffer_load_format: the typed rows of the matrix scale kernel, round 6)
LOAD_A = re.compile(r"global_load_dwordx4 a\[(\d+):(\d+)\]")
LOAD1 = re.compile(r"global_load_(?:dword|ubyte) v(\d+),")  # (the sixel diffusion's one-pixel ring)
WAIT = re.compile(r"s_waitcnt vmcnt\(\d+\) ; ring (.*)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    for m in AREG.finditer(text):
        if m.group(1) is not None:
            out.add(AGPR_BASE + int(m.group(1)))
        else:
            out.update(range(AGPR_BASE + int(m.group(2)), AGPR_BASE + int(m.group(3)) + 1))
    return out


class Block:
    def __init__(self, label):
        self.label, self.ops, self.succ_labels, self.falls = label, [], [], True
        self.succ, self.entry = [], frozenset()


def parse_blocks(lines):
    """ops: (line number, kind, payload) with kind in load / wait / waitall / instr."""
    blocks = [Block(None)]
    in_asm = False
    for no, line in lines:
        code = line.strip()
        if code.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if code.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm:
            m = LOAD.match(code)
            if m:
                blocks[-1].ops.append((no, "load", (int(m.group(1)), int(m.group(2)))))
                if code.endswith("; reserved"):
                    blocks[-1].ops.append((no, "reserve", (int(m.group(1)), int(m.group(2)))))
            m = LOAD_A.match(code)
            if m:
                blocks[-1].ops.append((no, "load", (AGPR_BASE + int(m.group(1)), AGPR_BASE + int(m.group(2)))))
            m = LOAD1.match(code)
            if m:
                blocks[-1].ops.append((no, "load", (int(m.group(1)), int(m.group(1)))))
            m = WAIT.match(code)
            if m:
                what = m.group(1).strip()
                blocks[-1].ops.append((no, "waitall", None) if what == "all" else (no, "wait", regs_of(what)))
            continue
        m = re.match(r"^([.\w$]+):", code)
        if m:
            blocks.append(Block(m.group(1)))
            continue
        if not code or code[0] in ";.":
            continue
        instr = code.split(";")[0].strip()
        op = instr.split()[0]
        if op == "s_branch":
            blocks[-1].succ_labels.append(instr.split()[1])
            blocks[-1].falls = False
            blocks.append(Block(None))
        elif op.startswith("s_cbranch"):
            blocks[-1].succ_labels.append(instr.split()[1])
            blocks.append(Block(None))
        elif op == "s_endpgm":
            blocks[-1].falls = False
            blocks.append(Block(None))
        else:
            blocks[-1].ops.append((no, "instr", instr))
    by_label = {b.label: b for b in blocks if b.label}
    for i, b in enumerate(blocks):
        b.succ = [by_label[l] for l in b.succ_labels if l in by_label]
        if b.falls and i + 1 < len(blocks):
            b.succ.append(blocks[i + 1])
    return blocks


def transfer(state, op, report=None):
    no, kind, payload = op
    if kind == "load":
        return state | {payload}
    if kind == "wait":
        return frozenset(t for t in state if not (set(range(t[0], t[1] + 1)) & payload))
    if kind == "waitall":
        return frozenset()
    if kind == "reserve":
        return state
    if report is not None and state:
        used = regs_of(payload)
        for lo, hi in state:
            if used & set(range(lo, hi + 1)):
                report.append((no, payload, "v[%d:%d]" % (lo, hi)))
    return state


def check_kernel(lines):
    blocks = parse_blocks(lines)
    loads = sum(1 for b in blocks for o in b.ops if o[1] == "load")
    waits = sum(1 for b in blocks for o in b.ops if o[1] in ("wait", "waitall"))
    if loads == 0:
        return 0, 0, []
    work = [blocks[0]]
    seen = {id(blocks[0])}
    while work:
        b = work.pop()
        state = b.entry
        for op in b.ops:
            state = transfer(state, op)
        for s in b.succ:
            merged = s.entry | state
            if merged != s.entry or id(s) not in seen:
                s.entry = merged
                seen.add(id(s))
                work.append(s)
    bad = []
    for b in blocks:
        if id(b) not in seen:
            continue
        state = b.entry
        for op in b.ops:
            state = transfer(state, op, bad)
    # a ring in accumulation registers belongs to the asm statements alone: no compiler-placed instruction may name one
    # of its registers ANYWHERE in the kernel (in flight or not)
    # ... and so does a ring in vector registers reserved from the compiler (loads marked "; reserved")
    ring_agprs = set()
    for b in blocks:
        for _, kind, payload in b.ops:
            if (kind == "load" and payload[0] >= AGPR_BASE) or kind == "reserve":
                ring_agprs.update(range(payload[0], payload[1] + 1))
    if ring_agprs:
        for b in blocks:
            for no, kind, payload in b.ops:
                if kind == "instr" and regs_of(payload) & ring_agprs:
                    bad.append((no, payload, "a reserved ring register (outside the asm statements)"))
    return loads, waits, bad


MFMA_WAIT_STATES = 4  # what `s_nop 3` provides between the last v_writelane and the first v_mfma (scale_stream.hip)
WRITELANE = re.compile(r"v_writelane_b32 v(\d+),")
NOP = re.compile(r"s_nop (\d+)")


def parse_all_blocks(lines):
    """Like parse_blocks, but every instruction counts -- those inside asm statements too (the v_writelane and its
    s_nop live there)."""
    blocks = [Block(None)]
    for no, line in lines:
        code = line.strip()
        m = re.match(r"^([.\w$]+):", code)
        if m:
            blocks.append(Block(m.group(1)))
            continue
        if not code or code[0] in ";.":
            continue
        instr = code.split(";")[0].strip()
        if not instr:
            continue
        op = instr.split()[0]
        blocks[-1].ops.append((no, "instr", instr))
        if op == "s_branch":
            blocks[-1].succ_labels.append(instr.split()[1])
            blocks[-1].falls = False
            blocks.append(Block(None))
        elif op.startswith("s_cbranch"):
            blocks[-1].succ_labels.append(instr.split()[1])
            blocks.append(Block(None))
        elif op == "s_endpgm":
            blocks[-1].falls = False
            blocks.append(Block(None))
    by_label = {b.label: b for b in blocks if b.label}
    for i, b in enumerate(blocks):
        b.succ = [by_label[l] for l in b.succ_labels if l in by_label]
        if b.falls and i + 1 < len(blocks):
            b.succ.append(blocks[i + 1])
    return blocks


def mfma_step(state, instr, no, report):
    """state: {vgpr: wait states since a v_writelane wrote it} (only registers below MFMA_WAIT_STATES are kept)."""
    op = instr.split()[0]
    if op.startswith("v_mfma") and state:
        operands = instr[len(op):].split(",")
        read = regs_of(",".join(operands[1:]))  # (everything but the destination: A, B and the accumulator)
        for r in read:
            if r in state and report is not None:
                report.append((no, instr, "v%d" % r, state[r]))
    m = NOP.match(instr)
    states = int(m.group(1)) + 1 if m else 1
    out = {r: n + states for r, n in state.items() if n + states < MFMA_WAIT_STATES}
    m = WRITELANE.match(instr)
    if m:
        out[int(m.group(1))] = 0
    return out


def check_mfma_waits(lines):
    if not any("v_mfma" in l for _, l in lines) or not any("v_writelane_b32" in l for _, l in lines):
        return 0, []
    blocks = parse_all_blocks(lines)
    for b in blocks:
        b.entry = None  # None: not reached yet; else {vgpr: wait states}
    blocks[0].entry = {}
    work = [blocks[0]]
    while work:
        b = work.pop()
        state = dict(b.entry)
        for no, _, instr in b.ops:
            state = mfma_step(state, instr, no, None)
        for s in b.succ:
            if s.entry is None:
                merged = dict(state)
            else:  # the minimum over the predecessors: a register is "recent" if it is on any path
                merged = dict(s.entry)
                for r, n in state.items():
                    merged[r] = min(n, merged.get(r, MFMA_WAIT_STATES))
            if merged != s.entry:
                s.entry = merged
                work.append(s)
    bad = []
    n_mfma = 0
    for b in blocks:
        if b.entry is None:
            continue
        state = dict(b.entry)
        for no, _, instr in b.ops:
            if instr.startswith("v_mfma"):
                n_mfma += 1
            state = mfma_step(state, instr, no, bad)
    return n_mfma, bad


def main(path):
    kernels, cur = [], None
    for no, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = []
            kernels.append((m.group(1), cur))
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append((no, line.rstrip("\n")))
    status, checked, mfma_kernels = 0, 0, 0
    for name, lines in kernels:
        n_mfma, late = check_mfma_waits(lines)
        if n_mfma:
            mfma_kernels += 1
        for no, instr, reg, n in late[:20]:
            print("%s:%d: '%s' reads %s %d wait state(s) after a v_writelane_b32 wrote it (needs %d)"
                  % (path, no, instr, reg, n, MFMA_WAIT_STATES))
        if late:
            print("%s: %d v_mfma reads too close behind a v_writelane" % (name, len(late)))
            status = 1
        loads, waits, bad = check_kernel(lines)
        if loads == 0:
            continue
        checked += 1
        if waits == 0:
            print("%s: asm loads but no ring waits" % name)
            status = 1
        for no, instr, ring in bad[:20]:
            print("%s:%d: '%s' names %s while its load is in flight" % (path, no, instr, ring))
        if bad:
            print("%s: %d instructions touch a register set in flight" % (name, len(bad)))
            status = 1
    if checked == 0:
        print("%s: no kernel with a register ring found" % path)
        status = 1
    if status == 0:
        print("check_ring_isa: %s: %d kernels, no instruction touches a register set in flight%s"
              % (path.split("/")[-1], checked,
                 "; %d kernels with v_writelane -> v_mfma, all %d wait states apart" % (mfma_kernels, MFMA_WAIT_STATES)
                 if mfma_kernels else ""))
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
