#!/bin/bash
# scratch/r5_s.sh -- the fused scale+sixel call in 1 / 2 / 4 / 8 pieces, with today's encode chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r5; mkdir -p "$out"
: > "$out/fused_pieces.txt"
for p in 0 1 2 4 8; do
  for rep in 1 2; do
  timeout 200 python bench.py --pieces $p --no-cpu-baseline --no-extras --no-dropin --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pieces %s: %.3f ms per step  %.1f Gpx/s  parity %s  scale kernel %.3f ms' % ('$p', d['ms_per_step'], d['value']/1e3 if d['unit'].startswith('M') else d['value'], d.get('parity_check',{}).get('ok'), d['roofline'].get('kernel_ms', -1)))" >> "$out/fused_pieces.txt"
  done
done
cat "$out/fused_pieces.txt"
