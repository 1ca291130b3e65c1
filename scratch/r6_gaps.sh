#!/bin/bash
# scratch/r6_gaps.sh -- where a step's time goes between kernels: start/end stamps of the default bench's kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6/gaps; rm -rf "$out"; mkdir -p "$out"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python bench.py --steps 6 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1
f=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+Kernel)(<[^>]*>)?", n); return (m.group(1) + (m.group(2) or "")) if m else n[:30]
# the last 3 steps: from a ScaleStreamMKernel<0 to the next
idx = [i for i, r in enumerate(rows) if "ScaleStreamMKernel<0" in r["Kernel_Name"]]
for a, b in zip(idx[-4:-1], idx[-3:]):
    t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
    line = []
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        line.append("%s +%.1f %.1f" % (short(r["Kernel_Name"])[:18], (s - prev_end) / 1e3, (e - s) / 1e3)); prev_end = e
    nxt = int(rows[b]["Start_Timestamp"])
    print("step %.1f us: " % ((nxt - t0) / 1e3) + " | ".join(line) + " | to next step +%.1f" % ((nxt - prev_end) / 1e3))
PY
find "$out" -name '*.csv' -delete
