#!/bin/bash
# dev: kernel-trace statistics of the headline step (three discriminator forwards) -> gpurun_out/$1_prof_srgan/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1_prof_srgan -- python3 $R/bench.py --workload srgan --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/$1_prof_srgan.log 2>&1
echo "prof rc=$?"
find $O/$1_prof_srgan -name "*kernel_stats.csv" | head
