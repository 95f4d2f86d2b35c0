#!/bin/bash
# HBM traffic counters of the gather kernel (8,388,608 uniform ids over 4,000,001 x 64), one run per counter group.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcg_fetch -- python3 scripts/pmc_gather.py > $OUT/pmcg_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcg_write -- python3 scripts/pmc_gather.py > $OUT/pmcg_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcg_l2 -- python3 scripts/pmc_gather.py > $OUT/pmcg_l2.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmcg_fetch gather_rows > $OUT/pmcg_summary.txt
python3 scripts/pmc_summary.py $OUT/pmcg_write gather_rows >> $OUT/pmcg_summary.txt
python3 scripts/pmc_summary.py $OUT/pmcg_l2 gather_rows >> $OUT/pmcg_summary.txt
cat $OUT/pmcg_summary.txt
