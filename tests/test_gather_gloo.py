"""The N>1 path on CPU: world_size-2 (and 3) `gloo` runs of the only exchange
step the hot path has -- the ordered gather of variable-length escape-sequence
buffers to rank 0 (timg_amd/gather.py; `nccl` = RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from timg_amd.gather import gather_frames_to_root, shard_frames


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(rank, i):
    rng = np.random.default_rng(1000 * rank + i)
    n = int(rng.integers(0, 5000)) if i % 4 else 0  # zero-length frames occur (identical animation frames)
    return rng.integers(0, 256, n, dtype=np.uint8)


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = [_frame(rank, i) for i in range(n_frames)]
        lens = torch.tensor([len(f) for f in frames], dtype=torch.int64)
        payload = torch.from_numpy(np.concatenate(frames + [np.zeros(64, np.uint8)]))  # slack behind the data
        got = gather_frames_to_root(payload, lens)
        if rank == 0:
            ok = got is not None and len(got) == world
            for r in range(world):
                for i in range(n_frames):
                    ok = ok and np.array_equal(got[r][i].numpy(), _frame(r, i))
            q.put(bool(ok))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 8), (3, 5), (2, 1)])
def test_gather_to_root_in_rank_and_frame_order(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def test_single_process_gather_is_identity():
    payload = torch.arange(10, dtype=torch.uint8)
    out = gather_frames_to_root(payload, torch.tensor([3, 0, 7]))
    assert [t.tolist() for t in out[0]] == [[0, 1, 2], [], [3, 4, 5, 6, 7, 8, 9]]


def test_shard_frames_partitions_exactly():
    for n in (1, 7, 64, 600):
        for world in (1, 2, 3, 8):
            for rr in (False, True):
                owned = [shard_frames(n, world, r, rr) for r in range(world)]
                flat = sorted(i for o in owned for i in o)
                assert flat == list(range(n))
                if rr:
                    assert all(o == list(range(r, n, world)) for r, o in enumerate(owned))
                else:
                    assert all(o == sorted(o) and (not o or o[-1] - o[0] == len(o) - 1) for o in owned)


# ---- the multi-pipeline step runner of bench.py on CPU (fake pipelines, gloo) -----------------
class _FakePipe:
    """Stands in for GridPipeline: 'encodes' a batch of 4 frames whose bytes depend on
    (rank, call number) so that ordering mistakes show."""

    def __init__(self, rank, pipe_id):
        self.rank, self.pipe_id, self.calls, self.stream = rank, pipe_id, 0, None
        self._out = None

    def scale(self, src):
        pass

    def encode(self):
        import time
        time.sleep(0.002 * ((self.pipe_id + self.calls) % 3))  # uneven speeds
        step = self.pipe_id + 3 * self.calls  # the global step this pipeline is running (3 pipelines)
        self.calls += 1
        frames = [np.full(1 + (step + i + self.rank) % 7, (step * 4 + i + 50 * self.rank) % 251, np.uint8)
                  for i in range(4)]
        self._out = frames

    def packed_output(self):
        lens = torch.tensor([len(f) for f in self._out], dtype=torch.int64)
        return torch.from_numpy(np.concatenate(self._out)), lens


def _runner_worker(rank, world, port, n_steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from timg_amd.pipeline import run_batched_streams
        pipes = [_FakePipe(rank, i) for i in range(3)]
        seen = []

        def gather(payload, lens):
            got = gather_frames_to_root(payload, lens)
            if rank == 0:
                seen.append([[t.numpy().copy() for t in frames] for frames in got])

        run_batched_streams(pipes, None, n_steps, 3, world, gather)
        if rank == 0:
            ok = len(seen) == n_steps
            for step, per_rank in enumerate(seen):
                for r in range(world):
                    for i in range(4):
                        want = np.full(1 + (step + i + r) % 7, (step * 4 + i + 50 * r) % 251, np.uint8)
                        ok = ok and np.array_equal(per_rank[r][i], want)
            q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_steps", [(2, 7), (2, 2)])
def test_batched_streams_gather_in_step_order(world, n_steps):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_runner_worker, args=(r, world, port, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def test_batched_streams_single_rank_runs_every_step():
    from timg_amd.pipeline import run_batched_streams
    pipes = [_FakePipe(0, i) for i in range(3)]
    run_batched_streams(pipes, None, 8, 3)
    assert sorted(p.calls for p in pipes) == [2, 3, 3]


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` as ONE plain process (no torchrun around it) used to parse --gpus and never read it:
    WORLD_SIZE was unset, the line said n_gpus 1.  It now starts torch.distributed.run itself; --dry-launch
    stops after the rendezvous (no device here): both ranks joined one gloo group."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TIMG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-1500:]
    assert "without a launcher: starting" in r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"launched_ranks": 2, "n_gpus": 2}


def test_bench_refuses_a_world_that_is_not_what_gpus_asked_for():
    """--gpus 4 under a launcher that started ONE rank: no number for another job than the one asked for."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-launch"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "refusing to report" in r.stderr, r.stdout + r.stderr
