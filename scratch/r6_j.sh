#!/bin/bash
# scratch/r6_j.sh -- after the diffusion's restructuring: random sixel geometries against the oracle, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 400 python scratch/sixel_stress.py 240 2>&1 | tail -6 | tee "$out/sixel_stress.txt"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee "$out/gpu_tests.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$out/smoke.txt"
