#!/bin/bash
# Train-step bench under rocprofv3 kernel stats; usage: gpu_profile_train.sh <tag>   (HMX_LIBRARY selects the library)
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$1 -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $OUT/prof_train_$1.log 2>&1
tail -1 $OUT/prof_train_$1.log | cut -c1-300
grep -h -E "scan_(fwd|bwd)_kernel" $OUT/prof_train_$1/*/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
