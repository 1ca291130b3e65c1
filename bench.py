#!/usr/bin/env python3
"""bench.py -- the hot path of hzeller/timg on MI355X, BASELINE.json's metric:

    Mpixels/s scale+sixel-encode, 4K -> 800 px, grid = 8x8
    (+ achieved HBM GB/s of the scale+blend kernel)

One "step" = one pass of the hot path over one batch of 64 synthetic 3840x2160
RGBA frames that are already resident in HBM: scale(+alpha-compose) each to
800x450, sixel-encode each, lengths back on the host; with N>1 ranks every
rank runs its own 64-frame batch (weak scaling, frames are independent) and
the variable-length outputs are gathered to rank 0 over RCCL for ordered
emission -- the only exchange step the path has.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 64] [--kind photo]

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the byte
accounting behind `roofline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(in_w, in_h, out_w, out_h, bg, cores, frames, target_seconds=12.0):
    """The reference's CPU path on the host cores, on a bounded sample of the
    same workload: the REAL reference (oracle/_ref, hzeller/timg sources) for
    scale + alpha-compose where it was built, the oracle's restatement for the
    sixel encode (libsixel is not in the reference tree).  Checker code used as
    a yardstick only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.Oracle()
    ref = oracle_lib.Ref.try_load()
    cores = max(1, cores)
    distinct = [np.ascontiguousarray(f) for f in frames]  # a few of the GPU's own input frames
    scaler = ref if ref is not None else orc

    def work(idx_list):
        for i in idx_list:
            fb = scaler.scale(distinct[i % len(distinct)], out_w, out_h)
            fb, _ = scaler.alpha_compose(fb, bg)
            orc.sixel_encode(fb, bg=bg, lookup_mode=0)

    work([0])  # warm-up
    t1 = time.perf_counter()
    work([0, 1])  # two frames on ONE thread: the per-core rate, for context
    single = 2 * in_w * in_h / 1e6 / (time.perf_counter() - t1)

    def run(n):
        shards = [list(range(t, n, cores)) for t in range(cores)]
        threads = [threading.Thread(target=work, args=(s,)) for s in shards]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return time.perf_counter() - t0

    # bounded sample: one calibration round, then enough frames for ~12 s of wall time
    # on these cores (at most 40 rounds)
    t_round = run(cores)
    rounds = max(1, min(40, int(target_seconds / max(t_round, 1e-3))))
    n_sample_frames = cores * rounds
    dt = run(n_sample_frames)
    mpx = n_sample_frames * in_w * in_h / 1e6 / dt
    return {
        "value": round(mpx, 2), "unit": "Mpixels/s", "cores": cores,
        "kind": "reference" if ref is not None else "port",
        "sample": (f"{n_sample_frames} frames {in_w}x{in_h}->{out_w}x{out_h} on {cores} threads: "
                   f"scale+blend = {'hzeller/timg sources (oracle/_ref)' if ref is not None else 'oracle port'}, "
                   "sixel = oracle restatement of libsixel (parity unpinned)"),
        "seconds": round(dt, 3),
        "single_thread_value": round(single, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pipelines", type=int, default=1,
                    help="independent batched streams per GPU in the timed region (1 = one batch at a time: "
                         "kernel durations are undisturbed, which the roofline object needs)")
    ap.add_argument("--batched-streams", type=int, default=3,
                    help="after the timed region, time the same K steps again on this many concurrent batched "
                         "streams and report it as `batched_streams` (0 = skip)")
    ap.add_argument("--frames", type=int, default=64, help="frames per rank per step (grid 8x8)")
    ap.add_argument("--kind", default="photo", choices=["photo", "noise", "alpha"])
    ap.add_argument("--mode", default="sixel", choices=["sixel", "quarter", "half", "kitty", "iterm2", "png"],
                    help="canvas: sixel (the BASELINE metric), half/quarter blocks, or a graphics protocol at "
                         "--compress=0 (kitty, iterm2; png = png::Encode alone)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 streaming")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the CPU sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import timg_amd
    from timg_amd.gather import gather_frames_to_root
    from timg_amd.pipeline import GridPipeline, run_batched_streams, synth_frames_on_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    in_w, in_h, out_w, out_h = 3840, 2160, 800, 450
    if args.mode in ("quarter", "half"):
        out_w, out_h = 200, 56  # BASELINE config 3: grid cell of an 800-cell canvas
    bg = (0x1E, 0x1E, 0x2E, 0xFF)
    blend = timg_amd.Blend.make(bg)
    # P independent batched streams ("1x MI355X batched streams", BASELINE config 3): every
    # stream owns a context (scratch), a HIP stream and one 64-frame batch in flight; step k
    # runs on stream k % P, driven by its own host thread.  The serial stages of the sixel
    # canvas keep only a few dozen CUs busy, so consecutive batches overlap on the chip.
    n_pipes = max(1, args.pipelines)
    n_extra = max(0, args.batched_streams)
    hips = [timg_amd.TimgHip(local_rank) for _ in range(max(n_pipes, n_extra))]
    pipes = [GridPipeline(h, args.frames, in_w, in_h, out_w, out_h, args.mode, blend) for h in hips]
    if args.kernel:
        for p in pipes:
            p.scaler.set_kernel(args.kernel)
    pipe = pipes[0]
    src = synth_frames_on_device(args.frames, in_w, in_h, args.kind, seed=rank)
    torch.cuda.synchronize()
    for p in pipes:
        p.stream.wait_stream(torch.cuda.current_stream())

    def record(stream):
        # HIP events on the stream the kernels are launched on
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def run_steps(n_steps, timed_events=None, n_pipes=n_pipes):
        """n_steps passes of the hot path, step k on stream k % P; with several ranks the
        outputs are gathered to rank 0 in step order by this (the main) thread."""
        run_batched_streams(pipes, src, n_steps, n_pipes, world, gather_frames_to_root, timed_events, record)

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
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    run_steps(args.warmup)
    events = []
    elapsed = timed(args.steps, n_pipes, events)  # THE timed region: exactly K steps

    scale_ms = [a.elapsed_time(b) for a, b, _ in events]
    encode_ms = [b.elapsed_time(c) for _, b, c in events]
    scale_avg_ms = sum(scale_ms) / len(scale_ms)
    alg_bytes = pipe.scaler.algorithmic_bytes() * args.frames  # per launch (one batch)
    achieved = alg_bytes / (scale_avg_ms * 1e-3) / 1e9
    total_px = world * args.frames * in_w * in_h * args.steps
    value = total_px / 1e6 / elapsed
    info = pipe.scaler.info()
    out_bytes = sum(pipe.lengths)

    result = {
        "metric": "Mpixels/s scale+sixel-encode, 4K->800px grid=8x8" if args.mode == "sixel"
                  else (f"Mpixels/s scale+{args.mode}-encode (--compress=0), 4K->800px grid=8x8"
                        if args.mode in ("kitty", "iterm2", "png")
                        else f"Mpixels/s scale+{args.mode}-block-encode, 4K->200x56 grid=8x8"),
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.frames}x {in_w}x{in_h} RGBA8 S-{args.kind} frames per GPU, resident in HBM "
                         f"-> scale+alpha-compose to {out_w}x{out_h} -> {args.mode} encode "
                         "(BASELINE metric config: 4K->800px grid=8x8)"),
            "frames_per_gpu": args.frames,
            "parallelism": (f"frames sharded {world} way(s), RCCL gather of output bytes to rank 0"
                            if world > 1 else "single GPU, batched launches") +
                           f"; {n_pipes} batch(es) in flight per GPU",
            "pipelines": n_pipes,
            "scale_kernel": "streaming" if (info["streaming_ok"] and args.kernel != 1) else "generic",
            "pass_order": "vertical-first" if info["vertical_first"] else "horizontal-first",
        },
        "roofline": {
            "kernel": "scale+alpha-compose (reads every source byte once)",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(scale_avg_ms, 4),
        },
        "stages_ms": {"scale_blend": round(scale_avg_ms, 3),
                      "encode": round(sum(encode_ms) / len(encode_ms), 3)},
        "output_bytes_per_step": out_bytes,
    }
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file))
            if t.get("workload_frames") == args.frames and t.get("kernel") == result["config"]["scale_kernel"]:
                result["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
        except Exception:
            pass

    if n_extra > 1:
        # Same K steps on n_extra concurrent batched streams (extra information, outside the
        # timed region above): the serial stages of the sixel canvas keep only ~64 CUs busy, so
        # consecutive batches overlap on the chip.  Kernel durations are NOT comparable with the
        # roofline object in this mode (concurrent kernels share the chip).
        run_steps(max(args.warmup, n_extra), None, n_extra)
        k_extra = max(args.steps, 2 * n_extra)
        dt = timed(k_extra, n_extra)
        result["batched_streams"] = {
            "streams": n_extra, "steps": k_extra, "ms_per_step": round(dt / k_extra * 1e3, 3),
            "value": round(world * args.frames * in_w * in_h * k_extra / 1e6 / dt, 1), "unit": "Mpixels/s",
            "note": "same workload, batches on independent streams overlap; not the contract's timed region",
        }

    if rank == 0 and not args.no_cpu_baseline:
        cores = args.cpu_threads or (os.cpu_count() or 1)
        host_frames = src[:min(4, args.frames)].cpu().numpy()
        result["cpu_baseline"] = cpu_baseline(in_w, in_h, out_w, out_h, bg, cores, host_frames,
                                              args.cpu_seconds) if args.mode == "sixel" else None
    if rank == 0:
        print(json.dumps(result), flush=True)
    for p in pipes:
        p.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for h in hips:
        h.close()


if __name__ == "__main__":
    main()
