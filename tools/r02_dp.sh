#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R/tools && timeout -k 10 600 python3 time_dp.py 2>&1 | grep -v amdgpu.ids | tail -5
