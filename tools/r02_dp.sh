#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_conv_pipe_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 &&
timeout -k 10 300 python3 tools/time_pipe.py 2>&1 | grep -v amdgpu.ids | cut -c1-175
