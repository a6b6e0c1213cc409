#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-pipe3}
cd $R/tools
timeout -k 10 200 python3 time_pipe.py > $R/gpurun_out/${T}_time.log 2>&1; cat $R/gpurun_out/${T}_time.log
timeout -k 10 200 python3 ablate_pipe.py > $R/gpurun_out/${T}_ablate.log 2>&1; cat $R/gpurun_out/${T}_ablate.log
