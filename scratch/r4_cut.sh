#!/bin/bash
# scratch/r4_cut.sh -- r4_sixel.sh, then the median cut's own trace (libtimg_hip_ct.so = sixel_canvas.hip with
# -DTIMG_CUT_TRACE: scratch/build_variant.sh ct sixel_canvas.hip -DTIMG_CUT_TRACE) on the bench's first frame
bash scratch/r4_sixel.sh "$@" || exit 1
TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_ct.so timeout -k 5 120 python3 scratch/cut_trace.py 2>&1 | grep "^cut:" | tail -80 > gpurun_out/r4/cut_trace.txt
grep -c round gpurun_out/r4/cut_trace.txt; grep "setup" gpurun_out/r4/cut_trace.txt; grep "cut: f0 round" gpurun_out/r4/cut_trace.txt | tail -14
