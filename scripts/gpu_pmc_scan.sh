#!/bin/bash
# HBM traffic counters of the backward scan kernel inside the configs[1] train step (2 steps = 12 launches), one run per counter.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcscan_fetch -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcscan_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcscan_write -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmcscan_write.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmcscan_fetch scan_ > $OUT/pmcscan_summary.txt
python3 scripts/pmc_summary.py $OUT/pmcscan_write scan_ >> $OUT/pmcscan_summary.txt
python3 scripts/pmc_summary.py $OUT/pmcscan_fetch block_ >> $OUT/pmcscan_summary.txt
python3 scripts/pmc_summary.py $OUT/pmcscan_write block_ >> $OUT/pmcscan_summary.txt
cat $OUT/pmcscan_summary.txt
