#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_kernels_gpu.py tests/test_discriminator_gpu.py tests/test_hr192_gpu.py -x -v -m gpu > $R/gpurun_out/crash.log 2>&1
echo "rc=$?"
grep -n "PASSED\|FAILED" $R/gpurun_out/crash.log | tail -3
grep -n "Fatal\|File \"" $R/gpurun_out/crash.log | head -30
