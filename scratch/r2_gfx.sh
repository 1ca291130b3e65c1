#!/bin/bash
# scratch/r2_gfx.sh -- first numbers for the graphics-protocol canvases at --compress=0 (kitty, iTerm2, png) and the
# block canvases on the metric geometry: bench lines + rocprofv3 per-kernel times of the kitty run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2gfx; rm -rf $out; mkdir -p $out
for m in kitty iterm2 png quarter half; do
  echo "== python bench.py --mode $m --no-cpu-baseline" >> $out/bench_modes.txt
  timeout 200 python bench.py --mode $m --no-cpu-baseline 2>>$out/err.txt | tail -1 >> $out/bench_modes.txt
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o prof -- python bench.py --mode kitty --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $out/trace.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
fn = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
with open(out + "/kitty_summary.txt", "w") as f:
    f.write("%-40s %6s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in csv.DictReader(open(fn)):
        n = r["Name"]
        if "timg_amd" not in n: continue
        m = re.search(r"(\w+Kernel(<[^>]*>)?)", n)
        short = m.group(1) if m else n[:40]
        f.write("%-40s %6s %12.1f %12.1f %7.2f\n" % (short[:40], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print(open(out + "/kitty_summary.txt").read())
PY
rm -rf $out/trace
cut -c1-420 $out/bench_modes.txt
