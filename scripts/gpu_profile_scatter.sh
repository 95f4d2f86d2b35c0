#!/bin/bash
# HBM traffic counters of the scatter-add kernels (16,777,216 uniform ids into 4,000,001 x 64), one run per counter group.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcs_fetch -- python3 scripts/pmc_scatter.py > $OUT/pmcs_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcs_write -- python3 scripts/pmc_scatter.py > $OUT/pmcs_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcs_l2 -- python3 scripts/pmc_scatter.py > $OUT/pmcs_l2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmcs_sq -- python3 scripts/pmc_scatter.py > $OUT/pmcs_sq.log 2>&1
for p in fetch write l2 sq; do python3 scripts/pmc_summary.py $OUT/pmcs_$p hmx_; done > $OUT/pmcs_summary.txt
cat $OUT/pmcs_summary.txt
