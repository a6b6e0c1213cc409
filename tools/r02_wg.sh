#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-wg}
timeout -k 10 600 python3 -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_discriminator_gpu.py $R/tests/test_generator_gpu.py -q -x > $R/gpurun_out/${T}_tests.log 2>&1; tail -4 $R/gpurun_out/${T}_tests.log
timeout -k 10 300 python3 $R/bench.py --steps 50 --no-cpu-baseline --no-secondary > $R/gpurun_out/${T}_bench.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("$R/gpurun_out/${T}_bench.json"))
print(j["value"], j["ms_per_step"])
for k,v in sorted(j["roofline"]["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]):
    if "wgrad" in k: print(k, v["launches_per_step"], round(v["avg_launch_us"],1), round(v["frac"],3))
PY
