#!/bin/bash
# scratch/r6_c5.sh -- BASELINE config 5 with and without the two-column kernel; parity subset first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "horizontal_first or config5 or random_geometries or any_source_width or transparent_pixels or fallback_chain or scale_bit_exact or config1" 2>&1 | tail -3
for h2 in 1 0; do
  echo "== TIMG_HIP_H2=$h2"
  TIMG_HIP_H2=$h2 timeout 600 python bench.py --config c5 --no-cpu-baseline --no-dropin --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print({'ms_per_step':d['ms_per_step'],'value':d['value'],'scale_ms':r['avg_launch_ms'],'frac':r['frac'],'stages':d['stages_ms'],'parity':d.get('parity_check',{}).get('ok')})"
done
} > "$out/c5_h2.txt" 2>&1
cat "$out/c5_h2.txt"
