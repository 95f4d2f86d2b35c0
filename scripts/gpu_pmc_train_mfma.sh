#!/bin/bash
# MFMA / memory-pipe counters of the train step's kernels (2 steps); usage: gpu_pmc_train_mfma.sh <pattern> [<pattern> ...]
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmctrain_mfma -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmctrain_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmctrain_ta -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmctrain_ta.log 2>&1 || true
for p in "$@"; do
  python3 scripts/pmc_summary.py $OUT/pmctrain_mfma "$p"
  python3 scripts/pmc_summary.py $OUT/pmctrain_ta "$p" || true
done
