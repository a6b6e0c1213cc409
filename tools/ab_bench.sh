#!/bin/bash
# A/B of two builds of libsrganst.so on the same box: alternate runs, report ms/step of each
L=srgan-st_amd/srganst/lib/libsrganst.so
for i in 1 2 3; do
  for v in prev new; do
    cp build_ab/libsrganst_$v.so $L
    echo -n "$v: "; python bench.py --no-roofline --no-cpu-baseline --no-full-step --steps ${AB_STEPS:-200} $AB_ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp build_ab/libsrganst_new.so $L
