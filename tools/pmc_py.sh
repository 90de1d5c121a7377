#!/bin/bash
# usage: tools/pmc_py.sh <tag> <counters...> -- <python script and args>   (one rocprofv3 --pmc pass over a python tool)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout -s KILL 150 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $out -- python3 "$@" > $out.log 2>&1
