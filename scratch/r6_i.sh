#!/bin/bash
# scratch/r6_i.sh -- after a change of the sixel diffusion: its parity tests (stop when red), then the default step's kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 120 python scratch/sixel_one.py 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sixel" 2>&1 | tail -5 | tee "$out/dither_tests.txt"
grep -q "failed\|error" "$out/dither_tests.txt" && exit 1
d=$out/prof_dither; rm -rf "$d"; mkdir -p "$d"
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --steps 20 --warmup 3 --no-dropin --no-cpu-baseline --no-extras > "$d/log.txt" 2>&1
tail -1 "$d/log.txt" | cut -c1-400
f=$(find "$d" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee "$out/dither_kernels.txt"
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].split("::")[-1] if "Kernel" in r["Name"] else r["Name"][:40]
    import re
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", r["Name"])
    print("%-45s calls %5s avg_us %9.1f" % (m.group(1) if m else r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf "$d"
