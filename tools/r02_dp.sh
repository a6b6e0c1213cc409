#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for k in 0 2 3 4 6; do echo "sr defer=$k"; SST_DEFER_D_WGRAD_SR=$k python3 bench.py --steps 60 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | cut -c150-215; done
echo shared; for k in 0 2; do SST_DEFER_D_WGRAD_SR=$k python3 bench.py --share-d-sr --steps 60 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | cut -c150-215; done
