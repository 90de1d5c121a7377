#!/bin/bash
# A/B of the round-3 switches of ogd_remap_lds.h: variant libraries built by tools/build_variant.sh, timed in turn, twice.
# usage: tools/ab_round3_switches.sh "variant names" [case filter]
for round in 1 2; do
for v in $1 shipped; do
  if [ "$v" = shipped ]; then lib=origami_amd/lib/libogdewarp.so; else lib=tools/variants/libogdewarp_$v.so; fi
  echo "== $v (pass $round)"
  OGDEWARP_LIB=$lib timeout -k 10 120 python -u tools/k2_variants_bench.py 50 "${2:-cell 25, linear}" 2>&1 | grep -v amdgpu.ids || exit 1
done
done
