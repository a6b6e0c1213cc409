#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-misc}
timeout -k 10 300 python3 -m pytest $R/tests/test_pixel_loss_gpu.py $R/tests/test_kernels_gpu.py -q -k "pixel or criterion or configs0 or big_tile" > $R/gpurun_out/${T}_tests.log 2>&1; tail -8 $R/gpurun_out/${T}_tests.log
cd $R/tools && timeout -k 10 300 python3 find_copies.py srgan > $R/gpurun_out/${T}_copies.log 2>&1; cat $R/gpurun_out/${T}_copies.log | cut -c1-260
