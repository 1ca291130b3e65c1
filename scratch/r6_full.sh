#!/bin/bash
# scratch/r6_full.sh -- what the driver runs at round end: the whole GPU suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/gpu_tests.txt" 2>&1; echo "pytest rc=$?" >> "$out/gpu_tests.txt"
tail -4 "$out/gpu_tests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > "$out/bench_default.txt" 2> "$out/bench_default.err"; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6/bench_default.txt") if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")}, d["roofline"], d["cpu_baseline"], d.get("parity_check", {}).get("ok"))
PY
