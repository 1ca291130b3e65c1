#!/bin/bash
# scratch/build_variant.sh <tag> <file.hip> [-DFLAG ...] -- experiment build: recompiles ONE kernel file
# with extra defines and links timg_amd/libtimg_hip_<tag>.so (select it with TIMG_HIP_LIB=<path>).
set -e
tag=$1; file=$2; shift 2
cd "$(dirname "$0")/../timg_amd/csrc"
make -s all >/dev/null
base=${file%.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wall -Wextra -Wno-unused-parameter "$@" -c $file -o build/${base}_$tag.o
objs=$(ls build/*.o | grep -v "_[a-z0-9]*\.o$" | grep -v "build/$base.o" || true)
objs=$(for o in capi scale_kernels scale_stream block_canvas sixel_canvas gfx_canvas autocrop synth dev_alloc resample_plan; do
         if [ $o = $base ]; then echo build/${base}_$tag.o; else echo build/$o.o; fi; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtimg_hip_$tag.so $objs
echo built timg_amd/libtimg_hip_$tag.so
