#!/bin/bash
# scratch/dither_rows.sh -- per-dispatch times of the sixel kernels for 64 frames of 800 x H
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
d=gpurun_out/dr; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python scratch/dither_rows.py > $d/log.txt 2>&1 || tail -5 $d/log.txt
python3 - $d <<'PY' | tee $d/result.txt
import csv, glob, sys, collections, re
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(\w+Kernel)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
seq = collections.defaultdict(list)
for s, e, k in rows:
    seq[k].append((e - s) / 1000.0)
hs = [96, 120, 216, 450]
for k, v in seq.items():
    if len(v) == 5 * len(hs):
        print("%-28s" % k, "  ".join("H=%d: %.1f us" % (h, sorted(v[i * 5:(i + 1) * 5])[2]) for i, h in enumerate(hs)))
PY
find $d -name '*.csv' -delete
