#!/bin/bash
# GPU side of tools/jpeg_atomic_variants.py: every variant library runs the regression test twice (the corruption
# differed from run to run).  usage: tools/jpeg_atomic_probe.sh C D E F I J
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  for k in 1 2; do
    echo "== variant $v run $k"
    OGDEWARP_LIB=$GRAFT_REPO_ROOT/origami_amd/lib/libogd_var$v.so timeout -k 10 120 python -m pytest tests/test_gpu_jpeg_decode.py -q -x -m gpu -k "long_colour_streams or whole_colour" --timeout 100 --timeout-method thread 2>&1 | grep -E "pixels differ|passed|failed" | head -4
  done
done
