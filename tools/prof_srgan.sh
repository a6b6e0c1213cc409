#!/bin/bash
# dev: kernel-trace statistics of the headline step -> gpurun_out/$1_prof_srgan/ ; usage: bash tools/prof_srgan.sh TAG [bench args...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_srgan -- python3 $R/bench.py --workload srgan --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary "$@" > $O/${TAG}_prof_srgan.log 2>&1
echo "prof rc=$?"
