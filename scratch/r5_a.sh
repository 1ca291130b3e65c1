#!/bin/bash
# scratch/r5_a.sh -- round 5, first GPU call: the vertical mix's issue cost in isolation, the reserved-register ring
# (4 / 6 rows in flight) beside the committed kernel with the wave-time trace, and the drop-in path by queue length.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
echo "== vmix" ; timeout 120 scratch/ubench/vmix.bin 2>&1 | tee $o/vmix.txt
echo "== bench_libs photo"; LIBS=main,r4,r6,t0,t6 timeout 300 python3 scratch/bench_libs.py 2>&1 | grep -v "^$" | tee $o/bench_libs_photo.txt | tail -12
echo "== twin_check timggrid"; timeout 300 tests/twins/build/twin_check timggrid 2>&1 | tail -3
echo "== twin_bench metric gpu"; timeout 300 tests/twins/build/twin_bench --config metric --paths gpu,host --queue 4 --queue 17 --queue 33 --queue 129 > $o/twin_bench_q.txt 2> $o/twin_bench_q.err
echo "== twin_bench metric gpu, cap 64 (old behaviour)"; TIMG_HIP_TWIN_BATCH_CAP=64 timeout 300 tests/twins/build/twin_bench --config metric --paths gpu,host --queue 17 --queue 129 > $o/twin_bench_q_cap64.txt 2>> $o/twin_bench_q.err
python3 - <<'PY'
import json
for fn in ("gpurun_out/r5/twin_bench_q.txt", "gpurun_out/r5/twin_bench_q_cap64.txt"):
    print(fn)
    for l in open(fn):
        if l.startswith("{"):
            d = json.loads(l)
            print("%-7s %-5s q%-4d frames %4d threads %3d  %8.1f Mpx/s  %.3f ms/frame" % (d["config"], d["path"], d["queue_len"], d["frames"], d["loader_threads"], d["mpx_per_s"], d["ms_per_frame"]))
PY
tail -3 $o/twin_bench_q.err
