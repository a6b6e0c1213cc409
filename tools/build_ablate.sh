#!/bin/bash
# dev: a second build of the library with the ablation switches compiled in (-DSST_PIPE_ABLATE) -> build_ab/libsrganst.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build_ab/obj
cd $R/srgan-st_amd/csrc
for f in *.hip; do
  o=$R/build_ab/obj/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ "$f" = "conv_pipe.hip" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function -ffp-contract=fast -DSST_PIPE_ABLATE -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libsrganst.so $R/build_ab/obj/*.o
echo built $R/build_ab/libsrganst.so
