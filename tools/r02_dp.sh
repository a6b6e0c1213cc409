#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | grep -v amdgpu.ids | tail -5 &&
timeout -k 10 300 python3 tools/time_wgrad_c3.py 2>&1 | grep -v amdgpu.ids
