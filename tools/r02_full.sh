#!/bin/bash
# full GPU validation + default bench + rocprof stats of the default workload
R=$GRAFT_REPO_ROOT
T=${1:-full}
python3 -m pytest $R/tests -m gpu -q > $R/gpurun_out/${T}_gputests.log 2>&1 || { grep -E "^(FAILED|ERROR)" $R/gpurun_out/${T}_gputests.log || true; }
tail -2 $R/gpurun_out/${T}_gputests.log
python3 $R/bench.py > $R/gpurun_out/${T}_bench_srgan.json 2> $R/gpurun_out/${T}_bench_srgan.err; echo "bench rc=$?"
cut -c1-330 $R/gpurun_out/${T}_bench_srgan.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_srgan -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/${T}_prof_srgan.log 2>&1
ls $R/gpurun_out/${T}_prof_srgan/*/ | head -4
