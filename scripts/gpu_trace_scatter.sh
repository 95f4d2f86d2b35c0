#!/bin/bash
# Kernel trace of the scatter-add probe (8,388,608 ids into 4,000,001 x 64), uniform then Zipf(1.05) ids.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
for mode in uniform zipf; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_scatter_$mode -- python3 scripts/probe_scatter.py 8388608 4000001 64 5 $mode > $OUT/trace_scatter_$mode.log 2>&1 || exit 1
  tail -1 $OUT/trace_scatter_$mode.log
  python3 scripts/kernel_groups.py $OUT/trace_scatter_$mode 1 14 | cut -c1-160
done
