#!/bin/bash
# A/B of two source trees on the same box: build_ab/prev/srgan-st_amd (copy of an older package incl. its built .so) vs the
# working tree; alternating runs, ms/step each.   AB_ARGS / AB_STEPS as in ab_bench.sh
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export SST_PKG_ROOT=$PWD/build_ab/prev/srgan-st_amd; else unset SST_PKG_ROOT; fi
    echo -n "$v: "; python bench.py --no-roofline --no-cpu-baseline --no-full-step --steps ${AB_STEPS:-200} $AB_ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
