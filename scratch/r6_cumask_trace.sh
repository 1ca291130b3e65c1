#!/bin/bash
# scratch/r6_cumask_trace.sh <reserved> [form] -- kernel timeline (start / end) of a few steps of scratch/r6_cumask.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6; mkdir -p "$out"
d=$out/prof_cm; rm -rf "$d"; mkdir -p "$d"
timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d "$d" -o prof -- python scratch/r6_cumask.py 8 ${1:-12} 0 ${2:-1} > "$d/log.txt" 2>&1
tail -1 "$d/log.txt"
f=$(find "$d" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ScaleStreamMKernel<0" in r["Kernel_Name"]]
t0 = int(rows[idx[-4]]["Start_Timestamp"])
for r in rows[idx[-4]:]:
    m = re.search(r"(\w+Kernel)", r["Kernel_Name"])
    print("%-22s q%-3s start %8.1f end %8.1f dur %7.1f us" % (m.group(1) if m else r["Kernel_Name"][:22], r.get("Queue_Id", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
