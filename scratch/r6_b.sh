#!/bin/bash
# scratch/r6_b.sh -- the tests this round added or tightened: twins / binary against a silent degrade, sixel calls on two streams, pins
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 1700 python -m pytest tests/test_twins.py tests/test_timg_binary.py tests/test_third_party_pins.py tests/test_gpu_parity.py -x -q -m gpu -k "twin or timg or pin or two_streams or async" > "$out/new_tests.txt" 2>&1; echo "pytest rc=$?" >> "$out/new_tests.txt"
tail -25 "$out/new_tests.txt"
