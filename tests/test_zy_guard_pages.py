"""Red zones (Next 1d): the GPU parity suites once more with every device buffer of the
library AND of the tests (timg_hip_malloc, the staging of host buffers, scaler tables, canvas scratch) on an
exact-size allocation between two UNMAPPED granules, slack poisoned (timg_amd/csrc/dev_alloc.h,
TIMG_HIP_GUARD=start|end16|end4).  A kernel that reads or writes a byte outside a buffer -- a prefetch
ring running ahead, a 16-byte load over the last pixels of a frame, a schedule table read one record too
far -- dies with a GPU memory access fault (or aborts at free when it wrote into the slack) instead of
passing on a box whose neighbouring pages happen to be mapped.

Child processes: a fault abort()s the process; the parent reports it as a failed test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SUITES = ["tests/test_gpu_parity.py", "tests/test_golden.py"]


def _child(code_or_args, mode, timeout=1500):
    env = dict(os.environ, TIMG_HIP_GUARD=mode, TIMG_SKIP_CANARY="1")
    return subprocess.run([sys.executable] + code_or_args, capture_output=True, text=True, timeout=timeout, env=env,
                          cwd=ROOT)


_LIE = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np, timg_amd
hip = timg_amd.TimgHip(0)
w, h, dw, dh = 256, 128, 64, 32
src = hip.upload(np.zeros((h, w, 4), np.uint8))
what = %(what)r
if what == "read":      # the scaler is told the frame has one more row than the buffer holds
    sc = hip.scaler(w, h + 1, dw, dh)
    dst = hip.malloc(dw * dh * 4)
else:                   # the destination is one row short
    sc = hip.scaler(w, h, dw, dh)
    dst = hip.malloc(dw * (dh - 1) * 4)
hip.scale_blend(sc, src, dst, 1)
hip.sync()
hip.free(dst); hip.free(src)
print("survived")
'''


def test_the_guard_itself_catches_an_over_read_and_an_over_write():
    """The instrument is only worth something if it fires: a scale call that is lied to about its source
    height must fault in mode end4; one whose destination is a row short must be caught in mode start
    (poisoned slack, checked at free) and fault in mode end4."""
    r = _child(["-c", _LIE % {"root": ROOT, "what": "read"}], "end4", 300)
    assert r.returncode != 0 and "survived" not in r.stdout, (r.stdout, r.stderr[-400:])
    # (how the fault surfaces is the box's: the runtime aborts the process with "Memory access fault by GPU ...", or -- seen on
    # a lease of round 4 -- the synchronising call returns hipErrorIllegalAddress and the library reports it)
    assert "Memory access fault" in r.stderr + r.stdout or "illegal memory access" in r.stderr + r.stdout, r.stderr[-600:]
    r = _child(["-c", _LIE % {"root": ROOT, "what": "write"}], "start", 300)
    assert r.returncode != 0 and "timg_hip GUARD" in r.stderr, (r.returncode, r.stderr[-600:])
    r = _child(["-c", _LIE % {"root": ROOT, "what": "write"}], "end4", 300)
    assert r.returncode != 0 and "survived" not in r.stdout, (r.stdout, r.stderr[-400:])
    # ... and the same call, told the truth, runs clean in every mode
    ok = (_LIE % {"root": ROOT, "what": "read"}).replace("h + 1", "h")
    for mode in ("start", "end16", "end4"):
        r = _child(["-c", ok], mode, 300)
        assert r.returncode == 0 and "survived" in r.stdout, (mode, r.stdout, r.stderr[-600:])


@pytest.mark.parametrize("mode", ["start", "end16", "end4"])
def test_gpu_parity_suites_under_guard_pages(mode):
    r = _child(["-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SUITES, mode)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-1500:])
    try:  # the whole log, for whoever has to find the kernel (gpurun_out/ travels back from the GPU box)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"guard_{mode}.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    assert r.returncode == 0, f"TIMG_HIP_GUARD={mode}: rc {r.returncode}\n{tail}"
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail


def test_smoke_in_a_fresh_process_like_the_driver_runs_it():
    """__graft_entry__.smoke() exactly as the driver invokes it (its own process, host buffers)."""
    code = 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")'
    for mode in ("", "end4"):
        env = dict(os.environ, TIMG_SKIP_CANARY="1")
        if mode:
            env["TIMG_HIP_GUARD"] = mode
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0 and "__SMOKE_OK__" in r.stdout, (mode, r.stdout[-500:], r.stderr[-800:])
