#!/bin/bash
# SQ counters for the coarse kernel (one 262144-query launch); usage: gpu_pmc_sq.sh <tag>   (env selects library / kernel)
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
T=$1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq_$T -- python3 scripts/pmc_topk.py 262144 > $OUT/sq_$T.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2_$T -- python3 scripts/pmc_topk.py 262144 > $OUT/sq2_$T.log 2>&1
python3 scripts/pmc_summary.py $OUT/sq_$T coarse
python3 scripts/pmc_summary.py $OUT/sq2_$T coarse
