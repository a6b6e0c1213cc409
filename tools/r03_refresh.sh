#!/bin/bash
# Regenerates the round-3 artefacts kept under profiles/ (run on the GPU box through gpurun; outputs under gpurun_out/$TAG_*).
# usage: bash tools/r03_refresh.sh TAG [quick]
TAG=${1:-r03f}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py > $O/${TAG}_bench_srgan.json 2> $O/${TAG}_bench_srgan.err; echo "srgan rc=$?"
if [ "$2" != "quick" ]; then
python3 bench.py --workload srresnet > $O/${TAG}_bench_srresnet.json 2>/dev/null; echo "srresnet rc=$?"
python3 bench.py --workload srgan_vgg --steps 50 --no-secondary > $O/${TAG}_bench_srgan_vgg.json 2>/dev/null; echo "vgg rc=$?"
python3 bench.py --hr 192 --batch 8 --steps 50 --no-secondary > $O/${TAG}_bench_srgan_hr192.json 2>/dev/null; echo "hr192 rc=$?"
fi
cd /tmp && export TMPDIR=/tmp
for wl in srgan; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$wl -- python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/${TAG}_prof_$wl.log 2>&1
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $O/${TAG}_pmc_${c%%:*}_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $O/${TAG}_pmc_${c%%:*}_$wl.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $O/${TAG}_pmc_sq_$wl.log 2>&1
  python3 $R/tools/pmc_summarize.py ${wl}_hr96_b16 $O/${TAG}_pmc_fetch_$wl $O/${TAG}_pmc_write_$wl > $O/${TAG}_pmc_summary_$wl.log 2>&1
  python3 $R/tools/pmc_sq_summary.py $(find $O/${TAG}_pmc_sq_$wl -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_sq_$wl.txt 2>&1
  cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
  echo "$wl profiled"
done
if [ "$2" != "quick" ]; then
  # HBM-side bytes of the 192-px configuration (BASELINE configs[4] on one GPU): its own PMC passes, its own key
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    rocprofv3 --pmc ${c#*:} --kernel-trace --output-format csv -d $O/${TAG}_pmc_${c%%:*}_hr192 -- python3 $R/bench.py --hr 192 --batch 8 --steps 2 --warmup 4 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $O/${TAG}_pmc_${c%%:*}_hr192.log 2>&1
  done
  python3 $R/tools/pmc_summarize.py srgan_hr192_b8 $O/${TAG}_pmc_fetch_hr192 $O/${TAG}_pmc_write_hr192 > $O/${TAG}_pmc_summary_hr192.log 2>&1
  cp $R/profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_hr192 -- python3 $R/bench.py --hr 192 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/${TAG}_prof_hr192.log 2>&1
fi
ls $O | grep ${TAG} | head -60
