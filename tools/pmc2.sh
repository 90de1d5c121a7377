#!/bin/bash
# usage: tools/pmc2.sh <binary> <tag> <KB_ONLY filter> <counters...>   (one rocprofv3 --pmc pass over a kbench binary)
bin=$1; shift; tag=$1; shift; only=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; cd /tmp && export TMPDIR=/tmp
KB_ONLY="$only" timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $out -- $GRAFT_REPO_ROOT/tools/$bin 6000 8000 25 3 > $out.log 2>&1
