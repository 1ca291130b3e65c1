#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 600 tests/twins/build/twin_check all /tmp/sixel.bin > $O/twin_check.txt 2>&1; echo "twin_check rc=$?"; tail -n 25 $O/twin_check.txt
TIMG_HIP_FILTER=bilinear timeout 300 tests/twins/build/twin_check bilinear 2>&1 | tail -3
timeout 1500 tests/twins/build/twin_bench --repeat 2 > $O/twin_bench.txt 2> $O/twin_bench.err; echo "twin_bench rc=$?"; cat $O/twin_bench.txt | cut -c1-330; tail -3 $O/twin_bench.err
