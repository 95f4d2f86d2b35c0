#!/bin/bash
# Train-step bench under rocprofv3 kernel stats; usage: gpu_profile_train.sh <tag> [extra bench args]
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=$1
shift
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/prof_train_$TAG.log 2>&1
tail -1 $OUT/prof_train_$TAG.log | cut -c1-300
python3 scripts/kernel_groups.py $OUT/prof_train_$TAG 6 30
