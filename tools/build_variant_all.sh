#!/bin/bash
# usage: tools/build_variant_all.sh NAME "-flag ..."   ->  tools/variants/libogdewarp_NAME.so
# Like tools/build_variant.sh, but EVERY source of the library is compiled with the extra flags.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=origami_amd/build/variants/all_$name
mkdir -p tools/variants $d
objs=""
for src in origami_amd/csrc/*.hip; do
  obj=$d/$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-sanitize -Wall -Wno-unused-function $@ -c $src -o $obj &
  objs="$objs $obj"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -o tools/variants/libogdewarp_$name.so $objs
echo tools/variants/libogdewarp_$name.so
