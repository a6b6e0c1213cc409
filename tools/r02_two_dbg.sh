#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for gpc in 3 2 3 2; do
echo -n "merged, SST_PIPE_GPC=$gpc: "
SST_PIPE_GPC=$gpc SST_OVERLAP_GD=1 timeout -k 10 300 python3 bench.py --steps 100 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['hip_graph'])"
done
echo -n "sequential, GPC=2: "; SST_PIPE_GPC=2 timeout -k 10 300 python3 bench.py --steps 100 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
