"""The RCCL gather behind its C-ABI (include/timg_hip_comm.h) on the GPU box.

Everything that initialises RCCL runs in a CHILD process: a communication library that cannot
come up (no usable librccl, a box whose fabric/IPC set-up it rejects) may abort() the process, and
under `pytest -x` that would take every other GPU test with it (round 2's driver run died exactly
there).  The file sorts last for the same reason.  When RCCL is unavailable the tests SKIP with
RCCL's own text; a wrong gather still FAILS."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import timg_amd
from timg_amd import comm
out = {}
try:
    out["rccl"] = comm.rccl_info()
except comm.NoRccl as e:
    print(json.dumps({"skip": str(e)})); sys.exit(0)
hip = timg_amd.TimgHip(0)
c = comm.Comm(0, 1, 0, comm.Comm.unique_id())
rng = np.random.default_rng(5)
lens = rng.integers(1, 5000, 9)
data = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
src = hip.upload(data)
dst = hip.malloc(int(lens.sum()))            # exact size: nothing may be written past it
# one call
all_len, total = c.gather_to_root(src, lens, n_frames_max=12, recv_ptr=dst, recv_cap=int(lens.sum()))
assert total == lens.sum() and (all_len[0, :9] == lens).all() and (all_len[0, 9:] == 0).all()
assert np.array_equal(hip.download(dst, total), data)
# two steps, as HipGatherWriter drives them
al = c.gather_lengths(lens, 12)
assert (al == all_len).all()
hip.upload(np.zeros(int(lens.sum()), np.uint8), dst)
assert c.gather_payload(src, al, dst, int(lens.sum())) == lens.sum()
assert np.array_equal(hip.download(dst, total), data)
# ... and with TIMG_HIP_COMM_PAYLOAD_READY: "I have waited for the payload's producer" -- the exchange runs on the
# communicator's stream without idling the device first (what bench.py's gather passes: ADVICE r4)
hip.upload(np.zeros(int(lens.sum()), np.uint8), dst)
hip.sync()
assert c.gather_payload(src, al, dst, int(lens.sum()), stream=comm.PAYLOAD_READY) == lens.sum()
assert np.array_equal(hip.download(dst, total), data)
# a short receive buffer is an error on every rank (here: the only one), not a hang, and moves nothing
hip.upload(np.full(int(lens.sum()), 7, np.uint8), dst)
try:
    c.gather_payload(src, al, dst, int(lens.sum()) - 1)
    raise SystemExit("short buffer accepted")
except RuntimeError as e:
    assert e.code == -3, e
assert (hip.download(dst, total) == 7).all()
# empty shard (a rank that owns no frame of a ragged stream)
al0 = c.gather_lengths(np.zeros(0, np.uint64), 3)
assert (al0 == 0).all() and c.gather_payload(0, al0, 0, 0) == 0
hip.free(src); hip.free(dst); c.close(); hip.close()
out["ok"] = True
print(json.dumps(out))
'''


def _run_child(extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], capture_output=True, text=True,
                       timeout=300, env=env)
    return r


@pytest.mark.gpu
def test_rccl_gather_through_the_c_abi_world_1():
    """timg_hip_gather_to_root / _lengths / _payload with one rank (all this box has): the same calls a
    multi-GPU job makes -- ncclCommInitRank, ncclAllGather of the lengths, the payload hand-over -- in a
    process that also holds torch's HIP runtime and torch's RCCL (the library binds that one)."""
    r = _run_child()
    # (RCCL prints its own banner to stdout, possibly after ours)
    last = ([ln for ln in r.stdout.splitlines() if ln.startswith("{")] or [""])[-1]
    if r.returncode != 0:
        tail = (r.stderr or "")[-1500:]
        # RCCL could not come up on this box (abort()/signal inside the library or an init error): not a
        # property of the gather -- but an assertion of the child is
        if r.returncode < 0 or "ncclCommInitRank" in tail or "ncclGetUniqueId" in tail:
            pytest.skip(f"RCCL unavailable on this box (child rc {r.returncode}): {tail}")
        pytest.fail(f"child rc {r.returncode}\n{r.stdout[-800:]}\n{tail}")
    assert last, r.stdout[-800:]
    out = json.loads(last)
    if "skip" in out:
        pytest.skip("no librccl: " + out["skip"])
    assert out.get("ok")
    path, version = out["rccl"]
    # the process holds torch => its bundled RCCL must be the one that served (never a second copy)
    assert "already mapped" in path and version > 0, out
    print("RCCL:", path, version)


@pytest.mark.gpu
def test_rccl_library_named_by_the_environment(tmp_path):
    """A process that maps no RCCL yet (no PyTorch: plain ctypes) binds the file TIMG_HIP_RCCL_LIB names before it
    searches the loader's path: `timg_hip_comm_rccl_info` says so.  A name that cannot be loaded is not an error: the
    search goes on."""
    rocm = os.path.realpath("/opt/rocm/lib/librccl.so.1")
    if not os.path.exists(rocm):
        pytest.skip("no /opt/rocm/lib/librccl.so.1")
    named = tmp_path / "the_rccl_of_this_deployment.so"
    os.symlink(rocm, named)
    child = (
        "import ctypes, sys\n"
        "L = ctypes.CDLL(%r)\n"
        "buf, ver = ctypes.create_string_buffer(512), ctypes.c_int(0)\n"
        "rc = L.timg_hip_comm_rccl_info(buf, ctypes.c_size_t(512), ctypes.byref(ver))\n"
        "print(rc, ver.value, buf.value.decode())\n"
    ) % os.path.join(ROOT, "timg_amd", "libtimg_hip_comm.so")
    for value, expect in ((str(named), "loaded as " + str(named)), (str(tmp_path / "absent.so"), "loaded as librccl")):
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TIMG_HIP_RCCL_LIB=value))
        assert r.returncode == 0, r.stdout + r.stderr
        rc, version, how = r.stdout.strip().splitlines()[-1].split(" ", 2)
        assert int(rc) == 0 and int(version) > 0, r.stdout
        assert expect in how or (expect == "loaded as librccl" and "loaded as /opt/rocm" in how), how


def test_comm_library_does_not_link_rccl():
    """libtimg_hip_comm.so binds RCCL at run time (dlopen, RTLD_NOLOAD first): no DT_NEEDED on librccl, so
    the host program's RCCL is never shadowed by a second one."""
    so = os.path.join(ROOT, "timg_amd", "libtimg_hip_comm.so")
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "NEEDED" in dyn and "rccl" not in dyn, dyn


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--frames", "4"], ["--config", "c4", "--frames", "10", "--chunk", "4"]])
def test_bench_exchanges_through_the_c_abi_gather(extra):
    """bench.py's exchange step is the PRODUCT one (timg_hip_gather_lengths / _payload behind the C-ABI), not a second
    implementation.  Only one GPU is reachable here: TIMG_BENCH_FORCE_GATHER=1 makes the single rank run the very
    calls a multi-GPU job makes per step (all-gather of the lengths, payload hand-over to the root).  The line must say
    which RCCL served, how many gathers ran and that the root received exactly the step's bytes -- and carry a green
    parity_check of the timed output."""
    env = dict(os.environ, TIMG_BENCH_FORCE_GATHER="1", TIMG_SKIP_CANARY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--prewarm", "0.05",
                        "--no-cpu-baseline", "--no-extras"] + extra, capture_output=True, text=True, timeout=600, env=env,
                       cwd=ROOT)
    line = ([ln for ln in r.stdout.splitlines() if ln.startswith("{")] or [""])[-1]
    if r.returncode != 0 and not line and r.returncode < 0:
        pytest.skip(f"bench.py died with signal {-r.returncode} while RCCL came up: {r.stderr[-800:]}")
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-600:], r.stderr[-1200:])
    out = json.loads(line)
    x = out["rccl"]
    assert "C-ABI" in x["via"] and x["rccl_version"] > 0 and "librccl" in x["lib"], x
    assert x["gathers"] >= 2 and x["bytes_at_root"] > 0, x
    # per step of the timed region: how long the exchange took on rank 0's clock and what arrived at the root
    assert len(x["gather_ms_per_step"]) == 2 and all(ms > 0 for ms in x["gather_ms_per_step"]), x
    if "--config" not in extra:  # (one exchange a step: exactly the step's bytes; c4's ragged last launch carries fewer)
        assert x["exchanges_per_step"] == 1 and x["bytes_at_root_per_step"] == [out["output_bytes_per_step"]] * 2, x
    else:
        assert x["exchanges_per_step"] == 3 and all(0 < b <= out["output_bytes_per_step"] for b in x["bytes_at_root_per_step"]), x
    assert out["parity_check"]["ok"], out["parity_check"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,extra", [
    (2, []),                                                   # the metric configuration: 64 frames per rank, weak scaling
    (2, ["--config", "c5", "--frames", "9", "--no-parity"]),   # strong scaling, contiguous blocks of 5 + 4 8K frames
    (4, ["--config", "c5", "--frames", "3", "--no-parity"]),   # ... of 1 + 1 + 1 + 0: a rank with an empty shard
    (4, ["--config", "c4", "--frames", "10", "--chunk", "2"]), # round-robin video frames, ragged: 3 + 3 + 2 + 2
])
def test_multi_rank_control_flow_on_one_gpu(ranks, extra):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), with the ranks
    SHARING the one reachable GPU and the collectives over gloo (TIMG_DIST_BACKEND=gloo): the multi-rank control flow --
    pre-warm decided by rank 0, barrier + max-over-ranks timing, per-step exchange beside the next step's kernels,
    strong-scaling shards incl. ragged and empty ones -- runs before the first real 8-GPU run does.  The numbers of
    such a run mean nothing; it must complete on every rank and report the whole job's frames."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, TIMG_DIST_BACKEND="gloo", TIMG_SKIP_CANARY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2",
           "--warmup", "1", "--prewarm", "0.05", "--no-cpu-baseline", "--no-extras", "--no-dropin"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    line = ([ln for ln in r.stdout.splitlines() if ln.startswith("{")] or [""])[-1]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-600:], r.stderr[-2000:])
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["value"] > 0 and out["rccl"]["ranks"] == ranks, out
    assert "torch.distributed over gloo" in out["rccl"]["via"], out["rccl"]
    if "--no-parity" not in extra:
        assert out["parity_check"]["ok"], out["parity_check"]
