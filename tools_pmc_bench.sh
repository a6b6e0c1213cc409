#!/bin/bash
# usage: tools_pmc_bench.sh <tag> <counter>   one --pmc pass over 3 eager steps of bench.py
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-full-step --workload ${BENCH_WORKLOAD:-srresnet} > $out.log 2>&1
tail -1 $out.log | cut -c1-120
