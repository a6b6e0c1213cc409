#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > $R/gpurun_out/b3.json 2> $R/gpurun_out/b3.err; echo rc=$?
python3 - <<'PY'
import json, os
j = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/b3.json").read().strip().split("\n")[-1])
print(j["value"], j["ms_per_step"], j["config"]["d_sr_forward"], j["config"]["step_tflops"])
print("shared:", j["srgan_shared_d_sr_step"])
print("srresnet:", j["srresnet_step"]["value"], j["srresnet_step"]["ms_per_step"])
print("roofline:", j["roofline"]["kernel"], j["roofline"]["frac"])
print("cpu:", j["cpu_baseline"]["value"])
PY
