"""The matrix scale kernel keeps source rows in flight in registers the compiler believes are already
loaded (timg_amd/csrc/scale_stream.hip, issue_next_row).  timg_amd/csrc/check_ring_isa.py proves on the
generated assembly that nothing touches such a register between its load and its wait; the Makefile runs
it on every build.  Here: the checker itself catches the failure modes seen in round 2, and the
assembly of the library that was actually built passes."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "timg_amd", "csrc", "check_ring_isa.py")
BUILT = os.path.join(ROOT, "timg_amd", "csrc", "build", "scale_stream-hip-amdgcn-amd-amdhsa-gfx950.s")

HEAD = """\
_Z6kernelv:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v[20:21], off
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[6:9], v[20:21], off
\t;;#ASMEND
.LBB0_1:
"""
TAIL = """\
\ts_cbranch_scc1 .LBB0_1
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; ring all
\t;;#ASMEND
\tv_mov_b32_e32 v30, v6
\ts_endpgm
.Lfunc_end0:
"""
STEP = """\
\t;;#ASMSTART
\ts_waitcnt vmcnt(1) ; ring v[{a}:{b}]
\t;;#ASMEND
\tv_cvt_f32_ubyte0_e32 v40, v{a}
\t;;#ASMSTART
\tglobal_load_dwordx4 v[{a}:{b}], v[20:21], off
\t;;#ASMEND
"""


def run(tmp_path, body):
    path = tmp_path / "k.s"
    path.write_text(HEAD + textwrap.dedent(body) + TAIL)
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_clean_ring_passes(tmp_path):
    rc, out = run(tmp_path, STEP.format(a=2, b=5) + STEP.format(a=6, b=9))
    assert rc == 0, out


def test_copy_before_the_wait_is_caught(tmp_path):
    # (what a tied asm operand compiled into: the set is copied, THEN waited for)
    rc, out = run(tmp_path, "\tv_mov_b64_e32 v[30:31], v[2:3]\n" + STEP.format(a=2, b=5) + STEP.format(a=6, b=9))
    assert rc == 1 and "v[2:5]" in out


def test_back_edge_copy_is_caught(tmp_path):
    # (a fifth register set: the reloaded set is copied into another one at the end of the loop body)
    rc, out = run(tmp_path, STEP.format(a=2, b=5) + STEP.format(a=6, b=9) + "\tv_mov_b64_e32 v[10:11], v[6:7]\n")
    assert rc == 1 and "v[6:9]" in out


def test_spill_of_a_set_in_flight_is_caught(tmp_path):
    rc, out = run(tmp_path, STEP.format(a=2, b=5) + "\tscratch_store_dwordx4 off, v[2:5], off offset:56\n" + STEP.format(a=6, b=9))
    assert rc == 1 and "scratch_store" in out


def test_clobber_on_a_side_path_is_caught(tmp_path):
    body = STEP.format(a=2, b=5) + "\ts_cbranch_vccz .LBB0_9\n\tv_mov_b32_e32 v3, 0\n.LBB0_9:\n" + STEP.format(a=6, b=9)
    rc, out = run(tmp_path, body)
    assert rc == 1 and "v_mov_b32_e32 v3, 0" in out


def test_typed_buffer_loads_are_ring_loads_too(tmp_path):
    """Round 6 (scratch/experiments/r6_typed_loads.patch): four buffer_load_format_xyzw per source row; a register of a row
    in flight that the compiler touches is the same bug whatever instruction loads it."""
    typed_head = HEAD.replace("global_load_dwordx4 v[2:5], v[20:21], off", "buffer_load_format_xyzw v[2:5], v20, s[4:7], s8 offen") \
                     .replace("global_load_dwordx4 v[6:9], v[20:21], off", "buffer_load_format_xyz v[6:8], v21, s[4:7], s8 offen")
    step = ("\t;;#ASMSTART\n\ts_waitcnt vmcnt(1) ; ring v[2:5]\n\t;;#ASMEND\n\tv_pk_mul_f32 v[2:3], v[2:3], s[10:11]\n"
            "\t;;#ASMSTART\n\tbuffer_load_format_xyzw v[2:5], v20, s[4:7], s8 offen\n\t;;#ASMEND\n"
            "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1) ; ring v[6:8]\n\t;;#ASMEND\n\tv_mul_f32_e32 v6, v6, v31\n"
            "\t;;#ASMSTART\n\tbuffer_load_format_xyz v[6:8], v21, s[4:7], s8 offen\n\t;;#ASMEND\n")
    path = tmp_path / "k.s"
    path.write_text(typed_head + step + TAIL)
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    path.write_text(typed_head + "\tv_min3_f32 v40, v40, v5, v8\n" + step + TAIL)  # alpha looked at before its load has landed
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and "v_min3_f32" in r.stdout, r.stdout


def test_loads_without_ring_markers_fail(tmp_path):
    path = tmp_path / "k.s"
    path.write_text(HEAD + "\ts_endpgm\n.Lfunc_end0:\n")
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and "no ring waits" in r.stdout


def test_built_library_passes():
    if not os.path.exists(BUILT):
        pytest.skip("no kept assembly (library built elsewhere)")
    r = subprocess.run([sys.executable, CHECK, BUILT], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    # 4 instantiations of the matrix kernel + 18 of the horizontal-first kernel (3 channel sets x 3 tap counts x 2 load widths)
    # + 24 of the two-column kernel (2 channel sets x first steps 2 to 5 x 2 / 3 / 4 loads a row) + 6 of a 28-tap one-column kernel (round 6)
    assert "52 kernels" in r.stdout
    # ... and the four matrix instantiations write their A operand by v_writelane, four wait states ahead of the first v_mfma
    assert "4 kernels with v_writelane -> v_mfma" in r.stdout
    # the sixel diffusion keeps eight source pixels (and, in its one-trip forms, eight palette indices) in flight the
    # same way: seven instantiations (two of them request the pixels in pairs: global_load_dwordx2)
    sixel = BUILT.replace("scale_stream", "sixel_canvas")
    r = subprocess.run([sys.executable, CHECK, sixel], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "7 kernels" in r.stdout


def test_single_register_ring_is_checked_too(tmp_path):
    body = """\
_Z6kernelv:
\t;;#ASMSTART
\tglobal_load_dword v7, v[20:21], off
\t;;#ASMEND
\tv_add_u32_e32 v8, 1, v7
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; ring v7
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""
    path = tmp_path / "k.s"
    path.write_text(body)
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and "v[7:7]" in r.stdout


# ---- v_writelane -> v_mfma wait states (the hand-counted s_nop of the matrix kernel) -----------------------------
MFMA = """\
_Z6kernelv:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v[20:21], off
\t;;#ASMEND
{pre}\t;;#ASMSTART
\tv_writelane_b32 v157, s48, 0
\tv_writelane_b32 v157, s49, 1
\tv_writelane_b32 v157, s50, 2
\tv_writelane_b32 v157, s51, 3
{nop}\t;;#ASMEND
{mid}\tv_mfma_f32_4x4x1_16b_f32 v[54:57], v157, v58, 0 cbsz:4
\tv_mfma_f32_4x4x1_16b_f32 v[50:53], v157, v59, 0 cbsz:4
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; ring all
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""


def run_mfma(tmp_path, pre="", nop="\ts_nop 3\n", mid=""):
    path = tmp_path / "m.s"
    path.write_text(MFMA.format(pre=pre, nop=nop, mid=mid))
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_pair_ring_is_checked_too(tmp_path):
    """The diffusion's pixel pairs: global_load_dwordx2 into v[a:b], released by a wait that names the pair -- either
    half named in between is a violation."""
    body = """\
_Z6kernelv:
\t;;#ASMSTART
\tglobal_load_dwordx2 v[8:9], v[20:21], off
\t;;#ASMEND
{mid}\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; ring v[8:9] v30
\t;;#ASMEND
\tv_add_u32_e32 v10, v8, v9
\ts_endpgm
.Lfunc_end0:
"""
    path = tmp_path / "k.s"
    path.write_text(body.format(mid="\tv_add_u32_e32 v22, 4, v22\n"))
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    path.write_text(body.format(mid="\tv_mov_b32_e32 v31, v9\n"))
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and "v[8:9]" in r.stdout


def test_writelane_then_mfma_with_the_nop_passes(tmp_path):
    rc, out = run_mfma(tmp_path)
    assert rc == 0 and "all 4 wait states apart" in out, out


def test_missing_or_short_nop_is_caught(tmp_path):
    rc, out = run_mfma(tmp_path, nop="")
    assert rc == 1 and "reads v157 0 wait state(s)" in out, out
    rc, out = run_mfma(tmp_path, nop="\ts_nop 1\n", mid="\ts_bitcmp1_b32 s98, 0\n")  # 2 + 1 = 3 < 4
    assert rc == 1 and "reads v157 3 wait state(s)" in out, out
    # ... and other instructions count as wait states like the nop does
    rc, out = run_mfma(tmp_path, nop="\ts_nop 1\n", mid="\ts_bitcmp1_b32 s98, 0\n\ts_cselect_b64 s[4:5], -1, 0\n")
    assert rc == 0, out


def test_short_path_around_the_nop_is_caught(tmp_path):
    # a branch from right behind the v_writelane to the v_mfma skips the padding: the minimum over the paths counts
    pre = ""
    nop = "\ts_nop 3\n"
    body = MFMA.format(pre=pre, nop="", mid="\ts_cbranch_scc1 .LBB0_7\n\ts_nop 3\n.LBB0_7:\n")
    path = tmp_path / "m.s"
    path.write_text(body)
    r = subprocess.run([sys.executable, CHECK, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and "reads v157 1 wait state(s)" in r.stdout, r.stdout


def test_sgpr_spill_writelanes_do_not_disturb(tmp_path):
    # the compiler spills scalar registers into lanes of a VGPR with the same instruction: no v_mfma reads those
    rc, out = run_mfma(tmp_path, mid="\tv_writelane_b32 v214, s12, 0\n")
    assert rc == 0, out
