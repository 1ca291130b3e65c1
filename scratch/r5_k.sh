#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
timeout 900 python3 -m pytest tests/test_zz_rccl.py -x -q -p no:cacheprovider 2>&1 | tail -6
TIMG_BENCH_FORCE_GATHER=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-dropin --steps 4 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rccl'] if 'rccl' in d else d.get('exchange'))"
