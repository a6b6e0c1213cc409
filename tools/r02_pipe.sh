#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-pipe}
timeout -k 10 300 python3 -m pytest $R/tests/test_conv_pipe_gpu.py -x -q > $R/gpurun_out/${T}_tests.log 2>&1; tail -15 $R/gpurun_out/${T}_tests.log
grep -q "passed" $R/gpurun_out/${T}_tests.log && ! grep -q "failed" $R/gpurun_out/${T}_tests.log || exit 1
cd $R/tools && timeout -k 10 300 python3 time_pipe.py > $R/gpurun_out/${T}_time.log 2>&1; cat $R/gpurun_out/${T}_time.log
