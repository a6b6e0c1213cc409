#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python3 -X faulthandler -m pytest $R/tests/test_discriminator_gpu.py -q -x -k "schedules or golden or capture_failure or graph_equals" > $R/gpurun_out/q_tests.log 2>&1; tail -5 $R/gpurun_out/q_tests.log
cd $R; for i in 1 2; do timeout -k 10 300 python3 bench.py --steps 100 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config']['hip_graph'])"; done
