#!/bin/bash
# scratch/r4_sixel.sh [variant-lib-tags...] -- sixel parity tests on the main library, then per-kernel times (rocprofv3
# kernel stats of the default bench step) for the main library and each libtimg_hip_<tag>.so, optional env per tag as
# TAG:VAR=VALUE (e.g. main:TIMG_HIP_DITHER_TRIPS=2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
scratch/run_logged.sh sixel_pytest env TIMG_SKIP_CANARY=1 timeout -k 5 240 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "sixel or config or golden"
tail -4 gpurun_out/r4/sixel_pytest.log | cut -c1-300
grep -q " passed" gpurun_out/r4/sixel_pytest.log && ! grep -q " failed" gpurun_out/r4/sixel_pytest.log || { echo "parity is red: no profile"; exit 1; }
for spec in main "$@"; do
  tag=${spec%%:*}; extra=; [ "$spec" != "$tag" ] && extra=${spec#*:}
  name=$(echo "$spec" | tr ':=' '__')
  out=gpurun_out/r4/sixel_prof_$name; rm -rf "$out"; mkdir -p "$out"
  lib=; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib $extra timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python bench.py --steps 8 --no-dropin --no-parity --no-cpu-baseline --no-extras ${BENCH_ARGS} > "$out/log.txt" 2>&1
  echo "== $spec: $(tail -1 $out/log.txt | cut -c1-140)"
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "timg_amd" not in n: continue
    print("  %-50s calls %4s avg_us %9.1f" % (n.split("(anonymous namespace)::", 1)[-1][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find "$out" -name '*kernel_trace.csv' -delete
done
