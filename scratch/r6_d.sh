#!/bin/bash
# scratch/r6_d.sh -- the kernels of c4 (3 x 200 frames a step) and c5 (1 x 256) under rocprofv3, beside 64-frame launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "c4 0" "c4 64" "c5 0" "c5 64"; do
  set -- $spec
  echo "=== --config $1 --chunk $2"
  bash profiles/prof.sh r6/prof_$1_$2 --config $1 --chunk $2 --no-extras --no-dropin --no-parity --steps 3 --warmup 1 2>&1 | cut -c1-300
done
