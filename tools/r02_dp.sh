#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_discriminator_gpu.py -x -q -m gpu -k "schedules or capture" 2>&1 | grep -v amdgpu.ids | tail -3 &&
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | cut -c150-240; done
SST_EARLY_D_PACK=0 python3 bench.py --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | cut -c150-240
