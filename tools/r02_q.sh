#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest $R/tests/test_conv_pipe_gpu.py -q -x > $R/gpurun_out/q_tests.log 2>&1; tail -5 $R/gpurun_out/q_tests.log
grep -q failed $R/gpurun_out/q_tests.log && exit 1
cd $R/tools && timeout -k 10 300 python3 time_s2d.py 2>&1 | grep -v amdgpu.ids
