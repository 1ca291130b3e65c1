#!/bin/bash
# scratch/build_variant.sh <tag> <file.hip> [-DFLAG ...] -- experiment build: recompiles ONE kernel file
# with extra defines and links timg_amd/libtimg_hip_<tag>.so (select it with TIMG_HIP_LIB=<path>).
# The kernel file's device assembly is kept as build/var_<tag>/<file>-hip-amdgcn-amd-amdhsa-gfx950.s and, for the
# files with a register ring, goes through check_ring_isa.py exactly as in the real build (a rejected build links nothing).
set -e
tag=$1; file=$2; shift 2
cd "$(dirname "$0")/../timg_amd/csrc"
make -s all >/dev/null
base=${file%.hip}
d=build/var_$tag
rm -rf $d; mkdir -p $d
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wall -Wextra -Wno-unused-parameter "$@" --save-temps=obj -c $file -o $d/$base.o
rm -f $d/*.hipi $d/*.bc $d/*.out* $d/*-host-*.s $d/*.hipfb
asm=$d/$base-hip-amdgcn-amd-amdhsa-gfx950.s
if [ $base = scale_stream ] || [ $base = sixel_canvas ]; then
  python3 check_ring_isa.py $asm
fi
cp $d/$base.o build/${base}_$tag.o
objs=$(for o in capi scale_kernels scale_stream block_canvas sixel_canvas gfx_canvas autocrop synth dev_alloc resample_plan; do
         if [ $o = $base ]; then echo build/${base}_$tag.o; else echo build/$o.o; fi; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtimg_hip_$tag.so $objs
echo built timg_amd/libtimg_hip_$tag.so
