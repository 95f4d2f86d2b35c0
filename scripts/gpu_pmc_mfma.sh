#!/bin/bash
# MFMA utilisation counters of the retrieval coarse kernel (one 262144-query launch), own rocprofv3 run.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_mfma.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmc_mfma coarse
