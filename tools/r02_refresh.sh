#!/bin/bash
# Regenerates the round-2 artefacts kept under profiles/ (run on the GPU box through gpurun; outputs under gpurun_out/r02f_*).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py > $O/r02f_bench_srgan.json 2> $O/r02f_bench_srgan.err; echo "srgan rc=$?"
python3 bench.py --workload srresnet > $O/r02f_bench_srresnet.json 2>/dev/null; echo "srresnet rc=$?"
python3 bench.py --workload srgan_vgg --steps 50 --no-secondary > $O/r02f_bench_srgan_vgg.json 2>/dev/null; echo "vgg rc=$?"
python3 bench.py --hr 192 --batch 8 --steps 50 --no-secondary > $O/r02f_bench_srgan_hr192.json 2>/dev/null; echo "hr192 rc=$?"
cd /tmp && export TMPDIR=/tmp
for wl in srgan srresnet; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_prof_$wl -- python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/r02f_prof_$wl.log 2>&1
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $O/r02f_pmc_${c%%:*}_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $O/r02f_pmc_${c%%:*}_$wl.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/r02f_pmc_sq_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $O/r02f_pmc_sq_$wl.log 2>&1
  echo "$wl profiled"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_prof_hr192 -- python3 $R/bench.py --hr 192 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/r02f_prof_hr192.log 2>&1
ls $O | grep r02f | head -40
