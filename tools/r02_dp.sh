#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_discriminator_gpu.py -x -q -m gpu -k "schedules or capture" 2>&1 | grep -v amdgpu.ids | tail -5 &&
timeout -k 10 600 python3 tools/stamp_step.py 2>&1 | grep -v amdgpu.ids | tail -11 &&
timeout -k 10 300 python3 bench.py --steps 50 --warmup 8 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep -v amdgpu.ids | cut -c1-300 &&
SST_EARLY_D_GT=0 timeout -k 10 300 python3 bench.py --steps 50 --warmup 8 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep -v amdgpu.ids | cut -c1-300
