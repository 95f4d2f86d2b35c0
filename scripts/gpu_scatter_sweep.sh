#!/bin/bash
# Scatter-add probe (16.8 M ids into 4 M x 64) under rocprofv3 kernel stats; usage: gpu_scatter_sweep.sh <tag>
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2
mkdir -p $OUT
TAG=$1
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scatter_$TAG -- python3 $ROOT/scripts/probe_scatter.py > $OUT/scatter_$TAG.log 2>&1 || exit 1
grep scatter_add $OUT/scatter_$TAG.log
python3 $ROOT/scripts/kernel_groups.py $OUT/prof_scatter_$TAG 6 12
