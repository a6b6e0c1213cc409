#!/bin/bash
# usage: tools_pmc.sh <tag> <counters...>   (one pass; counters in their own run, no sys/hip trace)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv.py > $out.log 2>&1
ls $out/*/ | head
