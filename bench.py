#!/usr/bin/env python3
"""bench.py -- the hot path of hzeller/timg on MI355X, BASELINE.json's metric:

    Mpixels/s scale+sixel-encode, 4K -> 800 px, grid = 8x8
    (+ achieved HBM GB/s of the scale+blend kernel)

One "step" = one pass of the hot path over one batch of synthetic RGBA frames that are
already resident in HBM (generated there by timg_hip_synth_frames; timg_amd.synth.hash_frame
is the same function on the host): scale(+alpha-compose) every frame, encode every frame,
lengths back on the host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config metric|c2|c3|c4|c5]

  metric (default)  64x 3840x2160 S-photo per GPU -> 800x450 -> sixel      weak scaling
  c2                1x  3840x2160 -> 800x450 -> sixel                       (BASELINE config 2)
  c3                64x 3840x2160 -> 200x56 -> quarter blocks, grid 8x8     (config 3)
  c4                600-frame 4K stream -> sixel, frames round-robin over the GPUs, strong scaling (config 4)
  c5                256x 7680x4320 S-alpha, checkerboard -> 800x450 -> sixel, contiguous blocks, strong (config 5)

With N > 1 ranks the variable-length outputs are gathered to rank 0 over RCCL for ordered
emission -- the only exchange step the path has.  Prints ONE JSON line (rank 0).  See DESIGN.md
"Measurement" for the byte accounting behind `roofline`.
"""
from __future__ import annotations

import argparse
import faulthandler
import json
import os
import sys
import threading
import time

faulthandler.enable()  # a fatal signal (a GPU memory fault aborts the process) leaves the Python stack in the log

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The fused step forks its pieces onto side streams: with ROCm's default of 4 hardware queues a process that holds
# more streams than that multiplexes them, and an event wait between two streams that share a queue costs
# milliseconds (measured in round 2: 1.15 ms per extra stream per batch; gone with 8 queues).  Runtime
# configuration, set before the HIP runtime comes up; INTEGRATION.md tells C++ hosts the same.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    # name: (frames in total or per GPU, in_w, in_h, out_w, out_h, mode, kind, scaling, round_robin, checker)
    "metric": dict(frames=64, per_gpu=True, in_w=3840, in_h=2160, out_w=800, out_h=450, mode="sixel", kind="photo"),
    "c2": dict(frames=1, per_gpu=True, in_w=3840, in_h=2160, out_w=800, out_h=450, mode="sixel", kind="photo"),
    "c3": dict(frames=64, per_gpu=True, in_w=3840, in_h=2160, out_w=200, out_h=56, mode="quarter", kind="photo"),
    "c4": dict(frames=600, per_gpu=False, round_robin=True, in_w=3840, in_h=2160, out_w=800, out_h=450,
               mode="sixel", kind="photo"),
    "c5": dict(frames=256, per_gpu=False, round_robin=False, in_w=7680, in_h=4320, out_w=800, out_h=450,
               mode="sixel", kind="alpha", checker=True),
}
BG = (0x1E, 0x1E, 0x2E, 0xFF)
PATTERN = (0x45, 0x47, 0x5A, 0xFF)


def cpu_baseline(in_w, in_h, out_w, out_h, blend_args, cores, frames, target_seconds=12.0):
    """The reference's CPU path on the host cores, on a bounded sample of the same workload:
    the REAL reference (oracle/_ref, hzeller/timg sources) for scale + alpha-compose where it
    was built, the oracle's restatement for the sixel encode (libsixel is not in the reference
    tree).  Checker code used as a yardstick only.

    `value` = end-to-end on all cores.  `stages` = the three stages timed separately on ONE
    thread (ms per frame), `sixel_only` = the sixel stage alone on 1 / 5 (the reference's
    encoder pool, src/timg.cc:333) / all threads, so that "x host sixel encode" can be read."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.Oracle()
    ref = oracle_lib.Ref.try_load()
    cores = max(1, cores)
    distinct = [np.ascontiguousarray(f) for f in frames]  # a few of the GPU's own input frames
    scaler = ref if ref is not None else orc
    bg, pattern, pw, ph = blend_args

    def work(idx_list):
        for i in idx_list:
            fb = scaler.scale(distinct[i % len(distinct)], out_w, out_h)
            fb, _ = scaler.alpha_compose(fb, bg, pattern, pw, ph)
            orc.sixel_encode(fb, bg, pattern, pw, ph, lookup_mode=0)

    # -- one thread, stage by stage (also the warm-up)
    stage = {"scale": 0.0, "blend": 0.0, "sixel": 0.0}
    scaled = []
    n1 = 2
    for i in range(n1):
        t0 = time.perf_counter()
        fb = scaler.scale(distinct[i % len(distinct)], out_w, out_h)
        t1 = time.perf_counter()
        fb, _ = scaler.alpha_compose(fb, bg, pattern, pw, ph)
        t2 = time.perf_counter()
        orc.sixel_encode(fb, bg, pattern, pw, ph, lookup_mode=0)
        t3 = time.perf_counter()
        stage["scale"] += t1 - t0
        stage["blend"] += t2 - t1
        stage["sixel"] += t3 - t2
        scaled.append(fb)
    single = n1 * in_w * in_h / 1e6 / sum(stage.values())
    # the same stage under the rule the GPU path runs (cell centre, lookup_mode 1) -- the ratio below compares
    # the device against libsixel's OWN rule (first hit, lookup_mode 0: what the reference executes); this
    # figure says what the comparison would be rule for rule
    t0 = time.perf_counter()
    for fb in scaled:
        orc.sixel_encode(fb, bg, pattern, pw, ph, lookup_mode=1)
    sixel_same_rule_ms = (time.perf_counter() - t0) / len(scaled) * 1e3

    def run(fn, n, threads):
        shards = [list(range(t, n, threads)) for t in range(threads)]
        ts = [threading.Thread(target=fn, args=(s,)) for s in shards]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    def sixel_only(idx_list):
        for i in idx_list:
            orc.sixel_encode(scaled[i % len(scaled)], bg, pattern, pw, ph, lookup_mode=0)

    sixel_rates = {"1": round(n1 * in_w * in_h / 1e6 / stage["sixel"], 1)}
    for threads in sorted({min(5, cores), cores}):
        n = threads * max(1, min(8, int(2.0 / max(stage["sixel"] / n1, 1e-3))))
        sixel_rates[str(threads)] = round(n * in_w * in_h / 1e6 / run(sixel_only, n, threads), 1)

    # bounded sample: one calibration round, then enough frames for ~target_seconds of wall time
    t_round = run(work, cores, cores)
    rounds = max(1, min(40, int(target_seconds / max(t_round, 1e-3))))
    n_sample_frames = cores * rounds
    dt = run(work, n_sample_frames, cores)
    mpx = n_sample_frames * in_w * in_h / 1e6 / dt
    return {
        "value": round(mpx, 2), "unit": "Mpixels/s", "cores": cores,
        "kind": "reference" if ref is not None else "port",
        "sample": (f"{n_sample_frames} frames {in_w}x{in_h}->{out_w}x{out_h} on {cores} threads: "
                   f"scale+blend = {'hzeller/timg sources (oracle/_ref)' if ref is not None else 'oracle port'}, "
                   "sixel = oracle restatement of libsixel (parity unpinned)"),
        "seconds": round(dt, 3),
        "single_thread_value": round(single, 2),
        "stages_ms_per_frame_1_thread": {k: round(v / n1 * 1e3, 2) for k, v in stage.items()},
        "sixel_only_source_mpx_per_s_by_threads": sixel_rates,
        "sixel_lookup_rule": ("host figures: libsixel's first-hit cache (restatement, parity unpinned) = what the reference "
                              "runs; the device runs the cell-centre rule (DESIGN 2) -- same stage on one host thread "
                              f"under the device's rule: {sixel_same_rule_ms:.2f} ms per frame"),
    }


def parity_check(pipe, src_batch, out_w, out_h, mode, blend_args, chunk):
    """Outside the timed region: frames 0 and N-1 of the batch the LAST timed step left in HBM -- the scaled
    + composed frame and its escape bytes -- against the checkers (oracle/_ref = the compiled reference for
    scale + compose where it travelled with the repo, else its restatement; the restatement for the canvas
    bytes).  A number whose bytes were never looked at is a claim; a mismatch fails the run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.Oracle()
    ref = oracle_lib.Ref.try_load()
    scaler = ref if ref is not None else orc
    bg, pattern, pw, ph = blend_args
    frames = sorted({0, chunk - 1})
    res = {"frames": frames, "scale_checker": "reference (oracle/_ref)" if ref is not None else "oracle restatement",
           "scale": "bit-exact", "canvas": "byte-exact vs restatement"
           + (" (lookup rule of the device: cell centre; libsixel itself unpinned)" if mode == "sixel" else "")}
    ok = True
    for i in frames:
        src = src_batch[i].cpu().numpy()
        want = scaler.scale(src, out_w, out_h)
        want, _ = scaler.alpha_compose(want, bg, pattern, pw, ph)
        got = pipe.scaled[i].cpu().numpy()
        if not np.array_equal(got, want):
            res["scale"] = f"MISMATCH frame {i}: {int(np.count_nonzero(got != want))} bytes"
            ok = False
        got_bytes = pipe.frame_bytes(i)
        if mode == "sixel":
            want_bytes = orc.sixel_encode(got, bg, pattern, pw, ph, lookup_mode=1)
        elif mode in ("quarter", "half"):
            want_bytes = orc.block_encode(got, quarter=(mode == "quarter"), x=(i % 8) * (out_w + 2))
        else:
            want_bytes = None
            res["canvas"] = "not checked in bench.py for this canvas (tests/test_gpu_parity.py does)"
        if want_bytes is not None and got_bytes != want_bytes:
            res["canvas"] = f"MISMATCH frame {i}: {len(got_bytes)} vs {len(want_bytes)} bytes"
            ok = False
    res["ok"] = ok
    return res


def cpp_dropin():
    """The drop-in path measured as a drop-in (north_star: "drops into the existing C++ renderer unchanged"):
    tests/twins/build/twin_bench drives HipRawRGBASource -> the reference's Renderer -> the Hip canvases -> the
    reference's BufferedWriteSequencer the way src/timg.cc:311-396,948-968 does (loader pool, one Send per image,
    queue of 4 / a grid row / the whole grid), and the reference's own classes on the host cores beside it.
    Not the contract's `value`: one process, one Send per image, bytes delivered to the host and written."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "twins", "build", "twin_bench")
    if not os.path.exists(exe):
        return {"skipped": "tests/twins/build/twin_bench not built (needs the reference's headers at build time)"}
    try:
        r = subprocess.run([exe, "--config", "metric,c4,c3", "--repeat", "2", "--cpu-frames", "64"], capture_output=True,
                           text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"skipped": "twin_bench timed out"}
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    out = {"unit": "Mpixels/s", "what": "timg's own loop (loader pool, Renderer, one Send per image, BufferedWriteSequencer on "
                                        "/dev/null); gpu = C++ twins on frames that are born in HBM, host = C++ twins on frames in HOST memory "
                                        "(what real files take: HipImageScaler uploads, scales, composes, downloads; PCIe Gen5 x16 "
                                        "bounds it at ~15 Gpx/s of 4K sources), cpu = the reference's classes on the host cores"}
    for x in rows:
        out.setdefault(x["config"], {}).setdefault(x["path"], {})["queue_%d" % x["queue_len"]] = {
            "mpx_per_s": x["mpx_per_s"], "ms_per_frame": x["ms_per_frame"], "frames": x["frames"],
            "loader_threads": x["loader_threads"]}
    return out


def under_a_profiler():
    """rocprofv3 preloads its tool library and configures it through the environment"""
    return (any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ)
            or "rocprofiler" in os.environ.get("LD_PRELOAD", ""))


def live_hbm_traffic(args):
    """roofline.traffic measured by THIS invocation: the same command line (short, without the extras) twice more as a
    subprocess under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- each counter in its own
    pass, no trace domain but --kernel-trace, as MI355X_MICROARCH.md's HBM section prescribes -- and the dominant scale
    kernel's average per dispatch, FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16-byte-a-lane reads as 64
    bytes), unit KiB.  None when rocprofv3 is missing or a pass fails (the bench line then falls back to the counter file
    profiles/collect_pmc.sh left, and says so)."""
    import csv, glob, shutil, subprocess, tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not tool or os.environ.get("TIMG_BENCH_NO_LIVE_PMC"):
        return None
    # (a run that is itself being profiled -- rocprofv3 preloads its tool library and configures it through the
    # environment -- does not start a profiler inside the profiler: the file figure serves)
    if under_a_profiler():
        return None
    inner = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
             "--no-extras", "--no-dropin", "--no-parity", "--no-live-pmc"]
    if args.kind:
        inner += ["--kind", args.kind]
    if args.frames:
        inner += ["--frames", str(args.frames)]
    if args.chunk:
        inner += ["--chunk", str(args.chunk)]
    env = dict(os.environ, TMPDIR="/tmp", TIMG_BENCH_NO_LIVE_PMC="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got = {}
    work = tempfile.mkdtemp(prefix="timg_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            try:
                subprocess.run([tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + inner,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=170, check=True)
            except Exception:
                return None
            per_kernel = {}
            for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r["Counter_Name"] == counter and "ScaleStream" in r["Kernel_Name"]:
                        k = "ScaleStream" + r["Kernel_Name"].split("ScaleStream")[1].split("(")[0]
                        per_kernel.setdefault(k, []).append(float(r["Counter_Value"]))
            if not per_kernel:
                return None
            got[counter] = {k: sum(v) / len(v) for k, v in per_kernel.items()}
        dom = max(got["FETCH_SIZE"], key=lambda k: got["FETCH_SIZE"][k])
        fetch, write = got["FETCH_SIZE"][dom], got["WRITE_SIZE"].get(dom, 0.0)
        return {"hbm_bytes_per_launch": int((fetch * 2 + write) * 1024),
                "counters": {"kernel": dom, "FETCH_SIZE_KiB_raw": round(fetch, 1), "WRITE_SIZE_KiB_raw": round(write, 1),
                             "correction": "FETCH_SIZE x2 (gfx950: 128-byte requests tallied as 64), WRITE_SIZE as reported; KiB"},
                "source": "measured in THIS run: bench.py re-ran its own command line (3 steps, no extras) under rocprofv3 --kernel-trace "
                          "--pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE after the timed region; average per dispatch of " + dom}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS),
                    help="metric = BASELINE.json's metric configuration (default); c2..c5 = BASELINE configs 2..5")
    ap.add_argument("--pipelines", type=int, default=1,
                    help="independent batched streams per GPU in the timed region (1 = one batch at a time: "
                         "kernel durations are undisturbed, which the roofline object needs)")
    ap.add_argument("--batched-streams", type=int, default=3,
                    help="after the timed region, time the same K steps again on this many concurrent batched "
                         "streams and report it as `batched_streams` (0 = skip; metric configuration only)")
    ap.add_argument("--partitioned", type=int, default=12,
                    help="CUs of every XCD kept free of the scale kernel in the `partitioned_streams` extra (0 = skip; "
                         "metric configuration, one rank)")
    ap.add_argument("--frames", type=int, default=0, help="override the configuration's frame count")
    ap.add_argument("--chunk", type=int, default=0,
                    help="frames per batched launch (c4, c5).  0 = the fewest equal launches of at most 256 frames: the sixel "
                         "chain is latency-bound (one frame costs 0.54 ms of kernels, sixty-four 0.80: profiles/r5/"
                         "chain_by_frames.txt) and with up to 256 frames a call every frame's diffusion still has a CU of "
                         "its own -- c4 as 3 x 200 frames 11.2 ms a step against 13.8 as 64s, c5 as 1 x 256 17.2 against 19.7 "
                         "(profiles/r6/chunk_sweep.txt)")
    ap.add_argument("--kind", default="", choices=["", "photo", "noise", "alpha"])
    ap.add_argument("--mode", default="", choices=["", "sixel", "quarter", "half", "kitty", "iterm2", "png"],
                    help="canvas override: sixel, half/quarter blocks, or a graphics protocol at --compress=0")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 streaming")
    ap.add_argument("--pieces", type=int, default=0,
                    help="sixel: 0 / 1 = scale and encode as two calls (kernels undisturbed; default: profiles/r3/"
                         "fused_pieces.txt); N > 1 = timg_hip_scale_sixel_encode with N pieces")
    ap.add_argument("--prewarm", type=float, default=0.4,
                    help="seconds of untimed steps before the warm-up steps (clock ramp-up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed output")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip tests/twins/build/twin_bench (the C++ drop-in path as src/timg.cc drives it)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_alpha / d2h / batched_streams")
    ap.add_argument("--sync-encode", action="store_true",
                    help="sixel on one GPU: every step ends with timg_hip_sixel_encode's blocking read-back of the byte counts "
                         "(rounds 1-4) instead of timg_hip_sixel_encode_async (two jobs alternate, counts read one step late)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-time budget of the CPU sample")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of this same command as "
                         "subprocesses, after the timed region); the figure then comes from profiles/hbm_traffic_*.json")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch / rendezvous check only: every rank joins the process group, one all-reduce, rank 0 prints "
                         '{"launched_ranks": N}; no device is touched (tests/test_gather_gloo.py runs it on the CPU)')
    args = ap.parse_args()

    # --gpus N MEANS N ranks.  The driver starts N > 1 under torch.distributed.run; started as ONE plain process
    # (`python bench.py --gpus 8`) this script used to read WORLD_SIZE alone and report n_gpus 1 without a word
    # (VERDICT r5): it now starts the launcher itself -- same command line, one rank per GPU on this node.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd[1:9])))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s rank(s) (WORLD_SIZE): refusing to report a number for "
                         "another job than the one asked for" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist
    if args.dry_launch:
        import datetime
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("TIMG_DIST_BACKEND", "gloo"), timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"launched_ranks": seen, "n_gpus": args.gpus}), flush=True)
        return
    import timg_amd
    from timg_amd.gather import gather_frames_to_root, shard_frames
    from timg_amd.pipeline import GridPipeline, run_batched_streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # TIMG_DIST_BACKEND=gloo: the collectives travel through host memory.  With it, several ranks can share one
    # GPU (LOCAL_RANK modulo the device count) -- the multi-rank control flow of this script can then be exercised
    # on a single-GPU box; the numbers of such a run mean nothing.
    backend = os.environ.get("TIMG_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (a collective that cannot complete fails after three minutes instead of holding the node)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180))

    # -- the exchange step: THE PRODUCT ONE.  timg_hip_gather_lengths / timg_hip_gather_payload (include/timg_hip_comm.h,
    # libtimg_hip_comm.so: RCCL behind the C-ABI -- what HipGatherWriter feeds the reference's sequencer from).  Bootstrapped
    # like any RCCL program: rank 0's unique id travels through torch.distributed's store.  timg_amd/gather.py (the same
    # exchange written with torch.distributed calls) remains for the gloo tests and as the fallback when the C-ABI
    # communicator cannot be created -- which the JSON line then says.  TIMG_BENCH_FORCE_GATHER=1: gather at world 1 too
    # (the same calls with one rank), so that the code path runs where only one GPU is reachable.
    force_gather = bool(os.environ.get("TIMG_BENCH_FORCE_GATHER")) and world == 1
    exchange = {"ranks": world, "via": "none (one rank)"}
    cabi = None
    if (world > 1 and backend == "nccl") or force_gather:
        from timg_amd import comm as tcomm
        box = [None]
        if rank == 0:
            try:
                box[0] = (tcomm.Comm.unique_id(), tcomm.rccl_info())
            except Exception as exc:  # no librccl, ...
                box[0] = ("error", str(exc))
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[0][0] == "error":
            exchange = {"ranks": world, "via": "torch.distributed (C-ABI communicator unavailable: %s)" % box[0][1]}
        else:
            # Every rank enters ncclCommInitRank together.  Then ONE small exchange is made and checked before any timing --
            # this path has never run on more than one GPU (only one is reachable from the build container): if the
            # communicator cannot be created or the self-test does not deliver every rank's bytes to the root in rank
            # order, ALL ranks agree (an all-reduce over torch.distributed) to exchange through torch.distributed
            # instead, and the JSON line says so -- a scaling run then still produces its numbers.
            import numpy as np
            ok, why = 1, ""
            try:
                cabi = tcomm.Comm(local_rank, world, rank, box[0][0])
                probe = torch.full((16,), 0x40 + rank, dtype=torch.uint8, device="cuda")
                back = torch.zeros(16 * world, dtype=torch.uint8, device="cuda") if rank == 0 else None
                torch.cuda.synchronize()
                al = cabi.gather_lengths(np.array([16], np.uint64), 1)
                got = cabi.gather_payload(probe.data_ptr(), al, back.data_ptr() if rank == 0 else 0,
                                          16 * world if rank == 0 else 0, stream=tcomm.PAYLOAD_READY)
                if rank == 0:
                    want = torch.arange(world, dtype=torch.uint8).repeat_interleave(16) + 0x40
                    if got != 16 * world or not torch.equal(back.cpu(), want):
                        ok, why = 0, "self-test: the root received %d bytes, not every rank's 16 in rank order" % got
            except Exception as exc:
                ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                agreed = int(flag.item())
            else:
                agreed = ok
            if agreed:
                exchange = {"ranks": world, "via": "timg_hip_gather_lengths + timg_hip_gather_payload (libtimg_hip_comm.so, C-ABI)",
                            "lib": box[0][1][0], "rccl_version": box[0][1][1], "gathers": 0, "bytes_at_root": 0,
                            "self_test": "16 bytes per rank delivered to the root in rank order before the timed region"}
            else:
                if cabi is not None:
                    try:
                        cabi.close()
                    except Exception:
                        pass
                cabi = None
                exchange = {"ranks": world, "via": "torch.distributed (the C-ABI communicator failed its start-up self-test on "
                                                   "some rank%s)" % (": " + why if why else "")}
    elif world > 1:
        exchange = {"ranks": world, "via": "torch.distributed over %s (testing configuration)" % backend}
    # (a curve measured on the fallback is a curve of torch.distributed's collectives, not of the product's exchange: said
    # in the line, and first thing on stderr where a log reader sees it)
    exchange["exchange_is_product"] = cabi is not None or world == 1
    if rank == 0:
        sys.stderr.write("bench.py: %d rank(s); exchange via %s%s\n" % (world, exchange["via"],
                         "" if exchange["exchange_is_product"] else "  [NOT the product's exchange]"))
        sys.stderr.flush()
    recv_buf = [None]

    def gather(payload, lens):
        """One exchange step: this rank's frames (payload: device bytes back to back, lens: byte counts) to rank 0."""
        if cabi is None:
            return gather_frames_to_root(payload, lens)
        import numpy as np
        lens_h = lens.cpu().numpy().astype(np.uint64)
        all_len = cabi.gather_lengths(lens_h, max(1, len(lens_h)))
        total = int(all_len.sum())
        recv_ptr = recv_cap = 0
        if rank == 0:
            if recv_buf[0] is None or recv_buf[0].numel() < total:
                recv_buf[0] = torch.empty(max(total + total // 4, 1 << 20), dtype=torch.uint8, device="cuda")
            recv_ptr, recv_cap = recv_buf[0].data_ptr(), recv_buf[0].numel()
        # (packed_output() handed over a snapshot it has synchronised: PAYLOAD_READY -- the exchange runs on the communicator's
        # stream beside the next step's kernels instead of waiting for the whole device)
        got = cabi.gather_payload(payload.data_ptr() if payload.numel() else 0, all_len, recv_ptr, recv_cap,
                                  stream=tcomm.PAYLOAD_READY)
        if rank == 0:
            exchange["gathers"] += 1
            exchange["bytes_at_root"] = got
        return None

    gather_log = []  # (ms, bytes at the root) of every exchange of the timed region, rank 0's clock

    def timed_gather(payload, lens):
        t0 = time.perf_counter()
        gather(payload, lens)
        if rank == 0:
            gather_log.append(((time.perf_counter() - t0) * 1e3, int(exchange.get("bytes_at_root", 0))))

    cfg = dict(CONFIGS[args.config])
    base_kind = cfg["kind"]
    if args.frames:
        cfg["frames"] = args.frames
    if args.kind:
        cfg["kind"] = args.kind
    if args.mode:
        cfg["mode"] = args.mode
        if args.mode in ("quarter", "half"):
            cfg["out_w"], cfg["out_h"] = 200, 56  # grid cell of an 800-cell canvas
    in_w, in_h, out_w, out_h, mode, kind = (cfg[k] for k in ("in_w", "in_h", "out_w", "out_h", "mode", "kind"))
    strong = not cfg["per_gpu"]
    # frames this rank owns (global indices: they enter the frame hash, so every frame of a
    # sharded stream is the frame a single GPU would have produced)
    if strong:
        mine = shard_frames(cfg["frames"], world, rank, round_robin=cfg.get("round_robin", False))
    else:
        mine = list(range(rank * cfg["frames"], (rank + 1) * cfg["frames"]))
    n_mine = len(mine)
    if strong:
        share = (cfg["frames"] + world - 1) // world  # the largest rank's frames
        if args.chunk > 0:
            chunk = max(1, min(args.chunk, share))
        else:
            launches = (share + 255) // 256
            chunk = max(1, (share + launches - 1) // launches)
    else:
        chunk = n_mine
    pw, ph = (18, 18) if cfg.get("checker") else (0, 0)  # -B pattern: pattern_size * cell px (9 x 18 cells)
    blend = timg_amd.Blend.make(BG, PATTERN if cfg.get("checker") else (0, 0, 0, 0), pw, ph)

    n_pipes = max(1, args.pipelines) if not strong else 1
    n_extra = 0 if (strong or args.no_extras or args.config != "metric") else max(0, args.batched_streams)
    # (a run under rocprofv3 is usually there for the per-kernel averages of the TIMED region: the two extras that run the same
    # kernels CONCURRENTLY -- several batches in flight, the partitioned chip -- would mix their stretched launches into them)
    profiled = under_a_profiler()
    if profiled:
        n_extra = 0
    hips = [timg_amd.TimgHip(local_rank) for _ in range(max(n_pipes, n_extra, 1))]
    pipes = [GridPipeline(h, chunk, in_w, in_h, out_w, out_h, mode, blend, pieces=args.pieces) for h in hips]
    if args.kernel:
        for p in pipes:
            p.scaler.set_kernel(args.kernel)
    pipe = pipes[0]

    def make_frames(kind, indices):
        t = torch.empty((len(indices), in_h, in_w, 4), dtype=torch.uint8, device="cuda")
        runs, start = [], 0  # runs of consecutive frame indices -> one generator call each
        for i in range(1, len(indices) + 1):
            if i == len(indices) or indices[i] != indices[i - 1] + 1:
                runs.append((start, i))
                start = i
        for a, b in runs:
            hips[0].synth_frames(kind, in_w, in_h, seed=0, first_frame=indices[a], n_frames=b - a,
                                 dst=t[a].data_ptr())
        hips[0].sync()
        return t

    src = make_frames(kind, mine)
    torch.cuda.synchronize()
    for p in pipes:
        p.stream.wait_stream(torch.cuda.current_stream())
    chunks = [src[i:i + chunk] for i in range(0, n_mine, chunk)]
    counts = ([len(shard_frames(cfg["frames"], world, r, round_robin=cfg.get("round_robin", False))) for r in range(world)]
              if strong else [n_mine] * world)
    n_launches_max = max((n + chunk - 1) // chunk for n in counts) if strong else 1
    if chunks and chunks[-1].shape[0] != chunk:  # a ragged tail gets its own pipeline
        tail_pipe = GridPipeline(hips[0], chunks[-1].shape[0], in_w, in_h, out_w, out_h, mode, blend, pieces=args.pieces)
    else:
        tail_pipe = None

    def record(stream):
        # HIP events on the stream the kernels are launched on
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def run_steps(n_steps, timed_events=None, n_pipes=n_pipes):
        """n_steps passes of the hot path; with several ranks the outputs are gathered to rank 0
        in step order by this (the main) thread."""
        if not strong:
            run_batched_streams(pipes, src, n_steps, n_pipes, 2 if force_gather else world, timed_gather, timed_events, record,
                                async_encode=not args.sync_encode)
            return
        # (one rank, sixel: a chunk is enqueued -- timg_hip_sixel_encode_async -- and its byte counts are read after the
        # next chunk's launches; with several ranks the gather needs a chunk's counts at once: the blocking form)
        use_async = (world == 1 and not force_gather and not args.sync_encode and pipe.can_async())
        for _ in range(n_steps):  # a step = the whole sharded stream, chunk by chunk
            pending = None  # (pipeline, job slot) whose byte counts have not been read yet
            for k in range(n_launches_max):
                c = chunks[k] if k < len(chunks) else None
                if c is not None:
                    p = tail_pipe if (tail_pipe is not None and c.shape[0] != chunk) else pipe
                    if pending is not None and pending[0] is not p:
                        # (the ragged tail has a pipeline -- a stream -- of its own on the SAME context: the scratch the two
                        # share is only safe in stream order, so the chunk in flight is waited for before the switch)
                        pending[0].finish(slot=pending[1])
                        pending = None
                    e0 = record(p.stream)
                    if p.fused:
                        p.step(c)
                        e1 = None
                    elif use_async:
                        p.scale(c)
                        e1 = record(p.stream)
                        p.encode_begin(slot=k & 1)
                    else:
                        p.scale(c)
                        e1 = record(p.stream)
                        p.encode()
                    # (one event between two launches: the next launch's e0 ends this one's encode stage; pipeline.py)
                    e2 = record(p.stream) if (p.fused or k == n_launches_max - 1 or k + 1 >= len(chunks)) else None
                    if timed_events is not None:
                        timed_events.append((e0, e1, e2, p.last_scale_ms))
                    if use_async:
                        if pending is not None:
                            pending[0].finish(slot=pending[1])
                        pending = (p, k & 1)
                if world > 1 or force_gather:
                    # every rank takes part in every gather; ranks that own fewer frames pad their
                    # lengths with zeros (the gather wants the same frame count everywhere)
                    if c is not None:
                        payload, lens = p.packed_output()
                    else:
                        payload = torch.empty(0, dtype=torch.uint8, device="cuda")
                        lens = torch.empty(0, dtype=torch.int64, device="cuda")
                    want = max(min(chunk, max(0, n_r - k * chunk)) for n_r in counts)
                    if lens.numel() < want:
                        lens = torch.cat([lens, torch.zeros(want - lens.numel(), dtype=torch.int64, device="cuda")])
                    timed_gather(payload, lens)
            if pending is not None:
                pending[0].finish(slot=pending[1])

    def timed(n_steps, n_pipes, timed_events=None):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run_steps(n_steps, timed_events, n_pipes)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # The shader clock ramps up over the first tens of milliseconds of load (a cold 64-frame scale launch
    # reads 0.85-0.94 ms, the same launch 0.70 ms once the clock has settled: profiles/r2): run untimed
    # steps for --prewarm seconds before the W warm-up steps, so that K timed steps see a settled clock.
    # (With several ranks a step contains collectives: rank 0's clock decides for everybody how long the
    # loop runs -- ranks that counted their own seconds would leave it after different numbers of steps and
    # the next collective would never complete.)
    t_pre = time.perf_counter()
    while True:
        go = time.perf_counter() - t_pre < args.prewarm
        if world > 1:
            t = torch.tensor([1 if go else 0], dtype=torch.int32, device=comm_dev)
            dist.broadcast(t, src=0)
            go = bool(t.item())
        if not go:
            break
        run_steps(1)
        torch.cuda.synchronize()
    run_steps(args.warmup)
    events = []
    del gather_log[:]
    elapsed = timed(args.steps, n_pipes, events)  # THE timed region: exactly K steps
    if rank == 0 and gather_log:
        # the exchange of the timed region, per step (a step of a strong-scaling configuration is several exchanges):
        # rank 0's wall clock around the two C-ABI calls (they end synchronised), the bytes that arrived at the root
        per_step = max(1, len(gather_log) // max(1, args.steps))
        steps_ms = [round(sum(ms for ms, _ in gather_log[i:i + per_step]), 3) for i in range(0, len(gather_log), per_step)]
        steps_by = [sum(b for _, b in gather_log[i:i + per_step]) for i in range(0, len(gather_log), per_step)]
        exchange["exchanges_per_step"] = per_step
        exchange["gather_ms_per_step"] = steps_ms[:16]
        exchange["bytes_at_root_per_step"] = steps_by[:16]
        exchange["gather_ms_mean"] = round(sum(steps_ms) / len(steps_ms), 3)
    parity = None
    if rank == 0 and not args.no_parity and n_mine > 0:
        # what the last timed step left behind (pipeline (K-1) % n_pipes holds it; strong configs: the last chunk)
        if strong:
            last_pipe = tail_pipe if (tail_pipe is not None and chunks[-1].shape[0] != chunk) else pipe
            last_src = chunks[-1]
        else:
            last_pipe, last_src = pipes[(args.steps - 1) % n_pipes], src
        parity = parity_check(last_pipe, last_src, out_w, out_h, mode,
                              (BG, PATTERN if cfg.get("checker") else (0, 0, 0, 0), pw, ph), last_src.shape[0])

    launches_per_step = len(chunks) if strong else 1
    # per launch of the hot path: device time of the scale kernels (two calls: events around the scale call; fused call:
    # the library's own events around every piece's scale launches, summed) and the rest of the step
    fused = pipe.fused
    scale_ms = [ms if e1 is None else e0.elapsed_time(e1) for e0, e1, _, ms in events]
    # (a launch without an end event of its own ends where the next launch on its stream begins)
    encode_ms = [e0.elapsed_time(e2 if e2 is not None else events[i + 1][0]) - sc
                 for i, ((e0, _, e2, _), sc) in enumerate(zip(events, scale_ms))]
    sizes = [c.shape[0] for c in chunks] * args.steps if strong else [chunk] * len(events)
    full = [s for s, n in zip(scale_ms, sizes) if n == chunk]  # (a ragged tail launch is not the roofline's launch)
    roof_frames = chunk
    if not full:  # this rank's shard is smaller than a batch (or empty): what it launched, as it is
        full = scale_ms
        roof_frames = max(sizes) if sizes else chunk
    scale_avg_ms = sum(full) / max(1, len(full))
    # (fused call: a batch is `pieces` scale launches, the library's events sum their device time; bytes and time are
    # divided by the same count, so `achieved` is the same number either way)
    launches_per_batch = pipe.pieces if fused else 1
    alg_bytes = pipe.scaler.algorithmic_bytes() * roof_frames // launches_per_batch  # per launch
    scale_avg_ms /= launches_per_batch
    achieved = alg_bytes / (scale_avg_ms * 1e-3) / 1e9 if scale_avg_ms > 0 else 0.0
    frames_total = cfg["frames"] if strong else world * cfg["frames"]
    total_px = frames_total * in_w * in_h * args.steps
    value = total_px / 1e6 / elapsed
    info = pipe.scaler.info()
    out_bytes = sum(pipe.lengths or [])  # (None on a rank whose shard is empty)

    canvas_name = {"sixel": "sixel", "quarter": "quarter-block", "half": "half-block"}.get(mode, mode + " (--compress=0)")
    result = {
        "metric": "Mpixels/s scale+sixel-encode, 4K->800px grid=8x8" if args.config == "metric" and mode == "sixel"
                  else f"Mpixels/s scale+{canvas_name}-encode, {in_w}x{in_h}->{out_w}x{out_h} ({args.config})",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_s": args.prewarm,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.config}: {cfg['frames']}x {in_w}x{in_h} RGBA8 S-{kind} frames "
                         f"{'in total' if strong else 'per GPU'}, generated and resident in HBM -> scale+alpha-compose"
                         f"{' over a checkerboard' if cfg.get('checker') else ''} to {out_w}x{out_h} -> {canvas_name} encode"
                         + (" (BASELINE metric config: 4K->800px grid=8x8)" if args.config == "metric" else "")),
            "frames_per_gpu": n_mine,
            "frames_per_launch": chunk,
            "parallelism": ((f"frames sharded {world} way(s) "
                             f"({'round-robin' if cfg.get('round_robin') else 'contiguous blocks'}), "
                             "RCCL gather of output bytes to rank 0" if world > 1 else "single GPU, batched launches")
                            + f"; {n_pipes} batch(es) in flight per GPU"),
            "pipelines": n_pipes,
            "pieces": pipe.pieces,
            "encode_call": ("timg_hip_sixel_encode_async: a step is enqueued, its byte counts are read after the next step's "
                            "launches (two jobs alternate)"
                            if (pipe.can_async() and world == 1 and not force_gather and not args.sync_encode)
                            else "blocking (the call returns the frames' byte counts)"),
            "scale_kernel": "streaming" if (info["streaming_ok"] and args.kernel != 1) else "generic",
            "pass_order": "vertical-first" if info["vertical_first"] else "horizontal-first",
        },
        "roofline": {
            "kernel": "scale+alpha-compose (reads every source byte once); "
                      + ("all-opaque frames: the compose epilogue finds nothing to blend" if kind == "photo"
                         else "frames with alpha: every output pixel is composed"),
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(scale_avg_ms, 4),
            "limiter": "no counters collected for this configuration (profiles/collect_pmc.sh)",
            "launches_per_step": launches_per_batch * (launches_per_step if strong else 1),
            "measured": ("HIP events of libtimg_hip.so around every piece's scale launches on the piece's own stream "
                         "(timg_hip_scale_sixel_encode: the kernels run beside the serial sixel stages of the pieces in front "
                         "-- durations are those of the timed region, not of a kernel alone on the chip; --pieces 1 times it alone)"
                         if fused else "HIP events around the scale call on the pipeline's stream; the kernel is alone on the chip"),
        },
        "rccl": exchange,
        "stages_ms": {"scale_blend": round(sum(scale_ms) / args.steps, 3),
                      "encode": round(sum(encode_ms) / args.steps, 3)},
        "output_bytes_per_step": out_bytes * (launches_per_step if strong else 1),
    }
    if parity is not None:
        result["parity_check"] = parity
    # HBM traffic and what limits the kernel: PMC counters of THIS command line, collected by profiles/collect_pmc.sh
    # (rocprofv3 cannot wrap a run from inside it) into profiles/hbm_traffic_<config>[_<kind>].json -- one file per
    # configuration, from the round named in it; profiles/hbm_traffic.json (metric, S-photo) is the older single file
    tname = args.config + ("_" + args.kind if args.kind and args.kind != base_kind else "")
    live = None
    if world == 1 and not args.no_live_pmc and not args.no_extras and result["config"]["scale_kernel"] == "streaming":
        live = live_hbm_traffic(args)
    for traffic_file in (os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % tname),
                         os.path.join(ROOT, "profiles", "hbm_traffic.json")):
        if not os.path.exists(traffic_file):
            continue
        try:
            t = json.load(open(traffic_file))
            if (t.get("workload_frames") == chunk and t.get("kernel") == result["config"]["scale_kernel"]
                    and t.get("config", "metric") == args.config and (t.get("kind") or "") == (tname.split("_", 1) + [""])[1]
                    and t.get("hbm_bytes_per_launch")):
                result["roofline"]["traffic"] = t["hbm_bytes_per_launch"] // launches_per_batch
                result["roofline"]["traffic_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / launches_per_batch / alg_bytes, 3)
                result["roofline"]["traffic_measured_in_this_run"] = False
                # the counters belong to ONE binary: the file names the library it was collected with (collect_pmc.sh)
                import hashlib
                with open(os.path.join(ROOT, "timg_amd", "libtimg_hip.so"), "rb") as lib_f:
                    lib_sha = hashlib.sha256(lib_f.read()).hexdigest()
                stale = t.get("library_sha256") != lib_sha
                result["roofline"]["traffic_stale"] = stale
                if stale:
                    result["roofline"]["traffic_stale_why"] = (
                        "profiles/%s was collected with another build of libtimg_hip.so (%s; this one %s): the ratio is that "
                        "build's, the limiter text is withheld" % (os.path.basename(traffic_file),
                                                                  (t.get("library_sha256") or "unrecorded")[:12], lib_sha[:12]))
                result["roofline"]["traffic_source"] = ("NOT measured in this run: profiles/%s, rocprofv3 --pmc FETCH_SIZE / "
                                                        "WRITE_SIZE passes of this same command (profiles/collect_pmc.sh)"
                                                        % os.path.basename(traffic_file))
                if live:  # (measured a moment ago, by this invocation: the file keeps the limiter text and the issue roof)
                    result["roofline"]["traffic"] = live["hbm_bytes_per_launch"] // launches_per_batch
                    result["roofline"]["traffic_over_algorithmic"] = round(live["hbm_bytes_per_launch"] / launches_per_batch / alg_bytes, 3)
                    result["roofline"]["traffic_measured_in_this_run"] = True
                    result["roofline"]["traffic_source"] = live["source"]
                    result["roofline"]["traffic_counters"] = live["counters"]
                    result["roofline"]["traffic_file_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / launches_per_batch / alg_bytes, 3)
                if t.get("limiter") and not stale:
                    result["roofline"]["limiter"] = t["limiter"]
                # the roof that binds this kernel beside the HBM one: the issue time of its vector instructions (counted by
                # the same collection: SQ_INSTS_VALU x 4 clocks / 1024 SIMDs / the clock the counters saw)
                if t.get("valu_issue") and not stale:
                    vi = dict(t["valu_issue"])
                    vi["frac_of_this_runs_launch"] = round(vi["issue_ms_per_launch"] / launches_per_batch / scale_avg_ms, 3)
                    result["roofline"]["valu_issue"] = vi
                break
        except Exception:
            pass
    if live and not result["roofline"].get("traffic_measured_in_this_run"):  # (no counter file for this configuration)
        result["roofline"]["traffic"] = live["hbm_bytes_per_launch"] // launches_per_batch
        result["roofline"]["traffic_over_algorithmic"] = round(live["hbm_bytes_per_launch"] / launches_per_batch / alg_bytes, 3)
        result["roofline"]["traffic_measured_in_this_run"] = True
        result["roofline"]["traffic_source"] = live["source"]
        result["roofline"]["traffic_counters"] = live["counters"]

    if not args.no_extras and args.config == "metric":
        # -- the same kernel with alpha actually present (S-alpha frames, composed over the background):
        # its own roofline object, measured in this run
        n_alpha = min(8, cfg["frames"])
        src_alpha = make_frames("alpha", list(range(rank * n_alpha, (rank + 1) * n_alpha)))
        src_alpha = src_alpha.repeat((chunk + n_alpha - 1) // n_alpha, 1, 1, 1)[:chunk].contiguous()
        pipe.stream.wait_stream(torch.cuda.current_stream())
        # (same steady clock as the timed region: launches for --prewarm seconds first -- three launches
        # right after the frames are generated read 20 % slow)
        t_end = time.perf_counter() + max(0.1, args.prewarm)
        while time.perf_counter() < t_end:
            for _ in range(8):
                pipe.scale(src_alpha)
            torch.cuda.synchronize()
        evs = []
        for _ in range(max(6, args.steps)):
            e0 = record(pipe.stream)
            pipe.scale(src_alpha)
            evs.append((e0, record(pipe.stream)))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        batch_bytes = pipe.scaler.algorithmic_bytes() * chunk
        result["roofline_alpha"] = {
            "kernel": "scale+alpha-compose on S-alpha frames (radial alpha ramp, transparent border, 10 % special "
                      "alphas), composed over the background",
            "bound": "hbm", "achieved": round(batch_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(batch_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": batch_bytes, "avg_launch_ms": round(ms, 4),
            "measured": "one launch per batch, alone on the chip (HIP events around the scale call)",
        }
        del src_alpha

    if not args.no_extras and not strong and mode == "sixel":
        # -- what the timed region leaves in HBM: the escape bytes.  Ordered emission needs them on the
        # host: the same K steps with every frame's bytes copied into pinned host memory by the encode
        # call itself (out_on_device = 0: one copy of `length` bytes per frame, not overlapped).
        pinned = torch.empty(pipe.cap * chunk, dtype=torch.uint8).pin_memory()
        host_out = pinned.numpy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            pipe.scale(src)
            hips[0].sixel_encode(pipe.scaled.data_ptr(), out_w, out_h, pad_blend=blend, n_frames=chunk,
                                 out=host_out, out_cap=pipe.cap, stream=pipe.stream_ptr())
        torch.cuda.synchronize()
        result["ms_per_step_with_d2h"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
        # ... and the way a host-side consumer would take them: the encode leaves the bytes in one of two
        # device buffers, a copy stream moves the used part of every frame's slot into pinned memory while
        # the next batch is scaled and encoded (22 MB per step over PCIe next to 2.3 ms of kernels).
        copy_stream = torch.cuda.Stream()
        dev_out = [pipe.out, torch.empty_like(pipe.out)]
        host_buf = [pinned, torch.empty(pipe.cap * chunk, dtype=torch.uint8).pin_memory()]
        copied = [None, None]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            i = k & 1
            if copied[i] is not None:
                pipe.stream.wait_event(copied[i])  # (the buffer's previous bytes have left)
            pipe.scale(src)
            lens = hips[0].sixel_encode(pipe.scaled.data_ptr(), out_w, out_h, pad_blend=blend, n_frames=chunk,
                                        out=dev_out[i].data_ptr(), out_cap=pipe.cap, stream=pipe.stream_ptr())
            worst = max(lens)
            with torch.cuda.stream(copy_stream):  # (the encode call returned after its stream was idle)
                host_buf[i][:chunk * worst].view(chunk, worst).copy_(dev_out[i].view(chunk, pipe.cap)[:, :worst],
                                                                    non_blocking=True)
                copied[i] = torch.cuda.Event()
                copied[i].record(copy_stream)
        torch.cuda.synchronize()
        result["ms_per_step_with_d2h_overlapped"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
        del dev_out, host_buf

    if profiled and not args.no_extras and not strong and args.config == "metric":
        result["concurrent_extras"] = ("batched_streams / partitioned_streams skipped: this run is under a profiler, whose per-kernel "
                                       "averages they would mix their concurrent launches into")
    if n_extra > 1:
        # Same K steps on n_extra concurrent batched streams (extra information, outside the
        # timed region above): the serial stages of the sixel canvas keep only part of the chip busy (the
        # median cut one CU per frame, the diffusion four), so consecutive batches overlap on the chip.  Kernel durations are NOT comparable with the
        # roofline object in this mode (concurrent kernels share the chip).
        run_steps(max(args.warmup, n_extra), None, n_extra)
        k_extra = max(args.steps, 24)  # (steady state: with a handful of steps the pipelines' first and last steps overlap nothing)
        dt = timed(k_extra, n_extra)
        result["batched_streams"] = {
            "streams": n_extra, "steps": k_extra, "ms_per_step": round(dt / k_extra * 1e3, 3),
            "value": round(world * cfg["frames"] * in_w * in_h * k_extra / 1e6 / dt, 1), "unit": "Mpixels/s",
            "note": "same workload, batches on independent streams overlap; not the contract's timed region",
        }

    if args.partitioned > 0 and not profiled and world == 1 and not strong and not args.no_extras and mode == "sixel" and args.config == "metric":
        # Same K steps with the chip PARTITIONED between the two calls (extra information, outside the timed region above):
        # the scale call on a stream whose kernels stay off `--partitioned` CUs of every XCD, the sixel chain on an
        # unrestricted stream of the greatest priority (timg_hip_stream_create, PartitionedSixelPipeline): step k + 1's scale
        # kernel runs beside step k's histogram / median cut / table / diffusion, which find the reserved CUs free.  On one
        # stream -- or two plain ones -- the scale kernel holds every CU until its last workgroup retires
        # (profiles/r6/overlap_streams.txt).  Kernel durations are NOT comparable with the roofline object in this mode
        # (the scale kernel has fewer CUs and shares the bus); the last step's frames are compared with the timed region's.
        from timg_amd.pipeline import PartitionedSixelPipeline
        pp = PartitionedSixelPipeline(hips[0], chunk, in_w, in_h, out_w, out_h, blend, reserved_cus_per_xcd=args.partitioned)
        try:
            pp.run(src, max(args.warmup, 2))
            k_part = max(args.steps, 24)  # (a pipeline of two stages: its first and last step overlap nothing)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pp.run(src, k_part)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = None
            if pipe.lengths is not None and pipe.out is not None:
                same = all(pp.frame_bytes(i) == pipe.frame_bytes(i) for i in (0, chunk // 2, chunk - 1))
            result["partitioned_streams"] = {
                "reserved_cus_per_xcd": args.partitioned, "scale_kernel_cus": 256 - 8 * (args.partitioned // 4 * 4),
                "steps": k_part, "ms_per_step": round(dt / k_part * 1e3, 3),
                "value": round(chunk * in_w * in_h * k_part / 1e6 / dt, 1), "unit": "Mpixels/s",
                "bytes_equal_to_the_timed_regions": same,
                "note": "same workload, ONE batch stream: the scale call of step k+1 on a CU-masked stream beside the sixel chain "
                        "of step k; not the contract's timed region",
            }
        except Exception as exc:  # (an extra: the line must come out)
            result["partitioned_streams"] = {"error": repr(exc)}
        finally:
            pp.close()

    if rank == 0 and not args.no_cpu_baseline and mode == "sixel":
        cores = args.cpu_threads or (os.cpu_count() or 1)
        host_frames = src[:min(4, n_mine)].cpu().numpy()
        result["cpu_baseline"] = cpu_baseline(in_w, in_h, out_w, out_h, (BG, PATTERN if cfg.get("checker") else (0, 0, 0, 0), pw, ph),
                                              cores, host_frames, args.cpu_seconds)
        cb = result["cpu_baseline"]
        gpu_sixel_mpx = chunk * in_w * in_h / 1e6 / (sum(encode_ms) / len(encode_ms) * 1e-3)
        cb["gpu_sixel_stage_source_mpx_per_s"] = round(gpu_sixel_mpx, 1)
        cb["gpu_over_host_sixel_encode"] = {k: round(gpu_sixel_mpx / v, 1) for k, v in
                                            cb["sixel_only_source_mpx_per_s_by_threads"].items()}
    if rank == 0 and not args.no_dropin and not args.no_extras and args.config == "metric":
        result["cpp_dropin"] = cpp_dropin()
    # The ONE JSON line is the last line of stdout: libraries that write through C stdio (RCCL prints a version banner on
    # stdout at its first communicator -- fully buffered into a pipe, it came out at process exit, BEHIND the line) are
    # flushed first, on every rank, and the ranks meet before rank 0 prints.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    parity_failed = parity is not None and not parity["ok"]
    for p in pipes:
        p.close()
    if tail_pipe is not None:
        tail_pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if cabi is not None:
        cabi.close()
    for h in hips:
        h.close()
    if parity_failed:
        raise SystemExit("bench.py: the timed output does not match the oracle: " + json.dumps(parity))


if __name__ == "__main__":
    main()
