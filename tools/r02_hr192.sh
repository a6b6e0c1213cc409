#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-hr192}
timeout -k 10 900 python3 -m pytest $R/tests/test_dp_gpu.py $R/tests/test_hr192_gpu.py -q -x > $R/gpurun_out/${T}_tests.log 2>&1; tail -8 $R/gpurun_out/${T}_tests.log
timeout -k 10 600 python3 $R/bench.py --hr 192 --batch 8 --steps 50 --no-secondary > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err; cut -c1-300 $R/gpurun_out/${T}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 $R/bench.py --hr 192 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/${T}_prof.log 2>&1
ls $R/gpurun_out/${T}_prof/*/ | head -4
