#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for o in "" "--overlap"; do
echo -n "merged $o: "
timeout -k 10 300 python3 -X faulthandler bench.py --steps 100 --no-cpu-baseline --no-roofline --no-secondary $o 2>$R/gpurun_out/ovl.err | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['hip_graph'])" || tail -5 $R/gpurun_out/ovl.err
done
echo -n "srresnet --overlap: "; timeout -k 10 300 python3 bench.py --workload srresnet --steps 100 --no-cpu-baseline --no-roofline --overlap 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['hip_graph'])"
echo -n "srresnet: "; timeout -k 10 300 python3 bench.py --workload srresnet --steps 100 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['hip_graph'])"
