#!/bin/bash
# dev: GPU tests + per-shape discriminator timings / phase ablation
R=$GRAFT_REPO_ROOT
T=${1:-probe}
python3 -m pytest $R/tests -m gpu -q -x --deselect tests/test_discriminator_gpu.py::test_train_engine_capture_failure_drops_every_graph > $R/gpurun_out/${T}_gputests.log 2>&1; tail -5 $R/gpurun_out/${T}_gputests.log
python3 -m pytest $R/tests -m gpu -q -k capture_failure > $R/gpurun_out/${T}_capfail.log 2>&1; tail -5 $R/gpurun_out/${T}_capfail.log
cd $R/tools && python3 time_d.py > $R/gpurun_out/${T}_time_d.log 2>&1; cat $R/gpurun_out/${T}_time_d.log
python3 ablate_d.py > $R/gpurun_out/${T}_ablate_d.log 2>&1; cat $R/gpurun_out/${T}_ablate_d.log
