#!/bin/bash
# scratch/r6_q.sh <tags> -- every sixel kernel's average duration for the main library and variant libraries
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
for tag in main "$@" main; do
  d=$out/prof_q; rm -rf "$d"; mkdir -p "$d"
  lib=X=1; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --steps 8 --warmup 2 --no-dropin --no-cpu-baseline --no-extras > "$d/log.txt" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tag" "$d/log.txt" <<'PY'
import csv, sys, re, json
row = {}
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+Kernel)", r["Name"])
    if m and "ScaleStream" not in m.group(1) and "Synth" not in m.group(1): row[m.group(1)] = float(r["AverageNs"]) / 1e3
ok = None
try:
    ok = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])["parity_check"]["ok"]
except Exception: pass
print("%-22s" % sys.argv[2], " ".join("%s %.1f" % (k.replace("Kernel", ""), v) for k, v in sorted(row.items())), "parity", ok)
PY
  rm -rf "$d"
done
