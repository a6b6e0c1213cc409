#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ (kernel trace + stats, eager mode)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-full-step "$@" > $out.log 2>&1
grep -h '"metric"' $out.log | head -2
