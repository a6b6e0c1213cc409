#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for k in 8 6 5 4 3 2; do echo "defer=$k"; SST_DEFER_D_WGRAD=$k timeout -k 10 300 python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | cut -c100-230; done
