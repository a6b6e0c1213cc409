#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-misc2}
timeout -k 10 600 python3 -m pytest $R/tests/test_discriminator_gpu.py $R/tests/test_generator_gpu.py -q -k "linear or run_to_run or full_seed0 or hr192" > $R/gpurun_out/${T}_tests.log 2>&1; tail -6 $R/gpurun_out/${T}_tests.log
cd $R/tools && timeout -k 10 600 python3 grad_errors.py > $R/gpurun_out/${T}_graderr.log 2>&1; cat $R/gpurun_out/${T}_graderr.log
timeout -k 10 300 python3 $R/bench.py --steps 50 --no-cpu-baseline --no-secondary > $R/gpurun_out/${T}_bench.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("$R/gpurun_out/${T}_bench.json"))
print(j["value"], j["ms_per_step"])
for k,v in j["roofline"]["kernels"].items():
    if v["bound"]=="hbm": print(k, round(v["avg_launch_us"],1), round(v["frac"],3))
PY
