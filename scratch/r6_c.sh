#!/bin/bash
# scratch/r6_c.sh -- twins + binary tests (degrade with the animation in)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 1700 python -m pytest tests/test_twins.py tests/test_timg_binary.py -x -q -m gpu > "$out/twin_tests.txt" 2>&1; echo "pytest rc=$?" >> "$out/twin_tests.txt"
tail -25 "$out/twin_tests.txt"
