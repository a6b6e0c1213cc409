#!/bin/bash
# Regenerates the artefacts kept under profiles/ (run on the GPU box through gpurun; outputs under gpurun_out/)
set -e
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/final_bench_srresnet.json 2> $R/gpurun_out/final_bench_srresnet.err
python3 $R/bench.py --workload srgan > $R/gpurun_out/final_bench_srgan.json 2>/dev/null
python3 $R/bench.py --workload srgan_vgg --steps 50 > $R/gpurun_out/final_bench_srgan_vgg.json 2>/dev/null
echo benches done
$R/tools_prof.sh final
$R/tools_pmc_bench.sh fetch FETCH_SIZE
$R/tools_pmc_bench.sh write WRITE_SIZE
$R/tools_pmc_bench.sh sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
cd $R
ls gpurun_out/prof_final gpurun_out/pmcb_fetch gpurun_out/pmcb_write gpurun_out/pmcb_sq | head -30
