#!/bin/bash
# SQ counters of the train step's kernels (2 steps); usage: gpu_pmc_train_sq.sh <kernel-name-pattern>
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmctrain_sq -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmctrain_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmctrain_sq2 -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmctrain_sq2.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmctrain_sq "$1"
python3 scripts/pmc_summary.py $OUT/pmctrain_sq2 "$1"
