#!/bin/bash
# Kernel trace of the train step at the reference's default width (--dim 60), grouped like the d = 64 one.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_d60 -- python3 bench.py --workload train --dim 60 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/prof_train_d60.log 2>&1 || exit 1
python3 scripts/kernel_groups.py $OUT/prof_train_d60 6 45
