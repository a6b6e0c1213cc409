#!/bin/bash
# A/B of environment settings on the same box: tools/ab_env.sh "VAR=a" "VAR=b" ...   (alternating runs, ms/step each)
for i in 1 2 3; do
  for v in "$@"; do
    echo -n "$v: "; env $v python bench.py --no-roofline --no-cpu-baseline --no-full-step --steps ${AB_STEPS:-200} $AB_ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
