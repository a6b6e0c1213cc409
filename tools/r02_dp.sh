#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hr192_gpu.py -x -q -s -m gpu 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/hr192.txt
