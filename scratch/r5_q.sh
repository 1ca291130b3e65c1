#!/bin/bash
# scratch/r5_q.sh -- the sixel chain's kernels by frames per batch (1 ... 64): which of them are a chain's latency
# (flat in the batch) and which are throughput (grow with it).  -> gpurun_out/r5/chain_by_frames.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r5; mkdir -p "$out"
: > "$out/chain_by_frames.txt"
for n in 1 4 16 64; do
  d="$out/cbf_$n"; rm -rf "$d"
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --frames $n --no-cpu-baseline --no-extras --no-dropin --no-parity --steps 8 --warmup 3 > "$out/cbf_$n.log" 2>&1
  python3 - "$d" "$n" >> "$out/chain_by_frames.txt" <<'PY'
import csv, glob, sys
d, n = sys.argv[1], sys.argv[2]
fn = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not fn:
    print("frames %s: no stats" % n); sys.exit(0)
print("== %s frame(s) of 3840x2160 -> 800x450 -> sixel per step (us per launch, rocprofv3 --kernel-trace --stats)" % n)
for r in csv.DictReader(open(fn[0])):
    name = r["Name"]
    if "timg_amd" not in name or "Synth" in name: continue
    short = name.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in name else name.split("(")[0]
    if "ScaleStream" in name: short = "ScaleStream" + name.split("ScaleStream")[1].split("(")[0]
    print("  %-40s %5s calls  avg %9.1f  min %9.1f" % (short[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  tail -1 "$out/cbf_$n.log" | cut -c1-200 >> "$out/chain_by_frames.txt"
  rm -rf "$d"
done
cat "$out/chain_by_frames.txt" | cut -c1-220
