#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
for p in 1 2 3; do
  for parts in 4 2; do
    TIMG_HIP_DITHER_PARTS=$parts timeout 200 python bench.py --pieces $p --no-dropin --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pieces $p parts $parts ms_per_step', d['ms_per_step'], 'parity', (d.get('parity_check') or {}).get('ok'))"
  done
done
