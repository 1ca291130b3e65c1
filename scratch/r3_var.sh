#!/bin/bash
# scratch/r2_var.sh <variant tags...> -- times scratch/bench_scale.py with experiment builds of the library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r3var; mkdir -p "$out"
for v in "$@"; do
  tag=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  lib=$PWD/timg_amd/libtimg_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/timg_amd/libtimg_hip.so
  echo "== $v" | tee -a "$out/var.txt"
  env TIMG_HIP_LIB=$lib N=${N:-64} KIND=${KIND:-photo} $(echo $envs | tr ',' ' ') timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror" | tee -a "$out/var.txt"
done
