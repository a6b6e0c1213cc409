#!/bin/bash
# usage: tools/pmc_any.sh <tag> <script.py> <counters...>   (one --pmc pass over a dev script under tools/)
tag=$1; shift; script=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmca_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/$script > $out.log 2>&1
tail -2 $out.log | cut -c1-150
