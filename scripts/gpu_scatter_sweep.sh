#!/bin/bash
# Scatter-add probe under rocprofv3 for several LDS stage sizes; usage: gpu_scatter_sweep.sh <tag> <kb> [<kb> ...]
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2
mkdir -p $OUT
TAG=$1
shift
cd /tmp
for kb in "$@"; do
  export HMX_SEG_STAGE_KB=$kb
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scatter_${TAG}_$kb -- python3 $ROOT/scripts/probe_scatter.py > $OUT/scatter_${TAG}_$kb.log 2>&1 || exit 1
  echo "stage $kb KB: $(grep scatter_add $OUT/scatter_${TAG}_$kb.log)"
  grep -h "segment_sum_staged" $OUT/prof_scatter_${TAG}_$kb/*/*kernel_stats.csv | awk -F, '{print "   staged kernel avg us:", $4/1000}'
done
