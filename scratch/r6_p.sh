#!/bin/bash
# scratch/r6_p.sh -- the driver's round-end commands on the final tree: the GPU suite, smoke(), the default bench line (timed)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee "$out/gpu_tests.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$out/smoke.txt"
t0=$(date +%s); python bench.py 2> "$out/bench_default.err" | tail -1 > "$out/bench_default.txt"; echo "bench.py wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/bench_default.txt").read())
r = d["roofline"]
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "vs_baseline")})
print("roofline", {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "traffic_over_algorithmic", "traffic_measured_in_this_run", "avg_launch_ms")})
print("cpu_baseline", d.get("cpu_baseline"))
print("dropin", json.dumps(d.get("cpp_dropin"))[:600])
PY
