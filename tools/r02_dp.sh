#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest $R/tests/test_dp_gpu.py -q -x -k "rccl or overlapped" > $R/gpurun_out/dp_tests.log 2>&1; tail -4 $R/gpurun_out/dp_tests.log
cd $R/tools && timeout -k 10 600 python3 time_dp.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200
