#!/bin/bash
# scratch/r6_full.sh -- the driver's round-end commands: GPU tests, smoke, the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/gpu_tests.txt" 2>&1; echo "pytest rc=$?" >> "$out/gpu_tests.txt"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; echo "smoke rc=$?" >> "$out/smoke.txt"
timeout 600 python bench.py > "$out/bench_default.txt" 2>&1; echo "bench rc=$?" >> "$out/bench_default.txt"
tail -5 "$out/gpu_tests.txt"; tail -2 "$out/smoke.txt"; tail -3 "$out/bench_default.txt"
