#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG=.. ..."   ->  tools/variants/libogdewarp_NAME.so
# A copy of the library whose K2 / fused kernels (ogd_remap.hip) are compiled with extra -D switches; every other
# object is the shipped one.  Select it at run time with OGDEWARP_LIB=tools/variants/libogdewarp_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants origami_amd/build/variants
python -m origami_amd.build > /dev/null
obj=origami_amd/build/variants/ogd_remap_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-sanitize -Wall -Wno-unused-function $@ -c origami_amd/csrc/ogd_remap.hip -o $obj
others=$(ls origami_amd/build/*.o | grep -v "/ogd_remap.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -o tools/variants/libogdewarp_$name.so $obj $others
echo tools/variants/libogdewarp_$name.so
