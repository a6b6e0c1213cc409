#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-pipe2}
cd $R/tools
timeout -k 10 200 python3 ablate_pipe.py > $R/gpurun_out/${T}_ablate.log 2>&1; cat $R/gpurun_out/${T}_ablate.log
timeout -k 10 200 python3 mfma_peak.py > $R/gpurun_out/${T}_peak.log 2>&1; cat $R/gpurun_out/${T}_peak.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc -- python3 $R/tools/time_pipe.py > $R/gpurun_out/${T}_pmc.log 2>&1
python3 $R/tools/pmc_sq_summary.py $(ls $R/gpurun_out/${T}_pmc/*/*counter_collection.csv | head -1) | tee $R/gpurun_out/${T}_pmc_summary.txt
