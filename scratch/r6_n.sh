#!/bin/bash
# scratch/r6_n.sh -- bench.py measuring roofline.traffic itself (rocprofv3 --pmc passes of its own command as subprocesses)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
for c in metric c3; do
  t0=$(date +%s); python bench.py --config $c --no-dropin 2> "$out/live_pmc_$c.err" | tail -1 > "$out/live_pmc_$c.json"
  echo "wall $(( $(date +%s) - t0 )) s"; tail -2 "$out/live_pmc_$c.err" | cut -c1-200
  python3 - "$out/live_pmc_$c.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d["roofline"]
print(d["config"]["workload"][:40], "ms", d["ms_per_step"], {k: r.get(k) for k in ("frac", "traffic", "traffic_over_algorithmic", "traffic_measured_in_this_run", "traffic_file_over_algorithmic", "traffic_stale")})
print("  ", r.get("traffic_source"), r.get("traffic_counters"))
PY
done
