#!/bin/bash
# Round profile set: default bench line, kernel stats of both workloads, retrieval counter passes, configs[4] line and kernel stats.
# usage: gpu_profile_round.sh <part>   (a = default line + retrieval stats + counters, b = train stats + configs[4])
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
set -e
if [ "$1" = "a" ]; then
  timeout -k 10 500 python3 -u bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  tail -n 1 $OUT/bench_default.json | cut -c1-400
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_retr -- python3 bench.py --workload retrieval --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_retr.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_write.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_l2.log 2>&1
  for p in fetch write l2; do python3 scripts/pmc_summary.py $OUT/pmc_$p coarse; done > $OUT/pmc_retr_summary.txt
  cat $OUT/pmc_retr_summary.txt
else
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $OUT/prof_train.log 2>&1
  python3 scripts/kernel_groups.py $OUT/prof_train 6 12
  timeout -k 10 500 python3 -u bench.py --workload train --batch 512 --seq-len 512 --dim 128 --steps 20 --warmup 5 > $OUT/train_config4.json 2> $OUT/train_config4.err
  tail -n 1 $OUT/train_config4.json | cut -c1-300
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_c4 -- python3 bench.py --workload train --batch 512 --seq-len 512 --dim 128 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_train_c4.log 2>&1
  python3 scripts/kernel_groups.py $OUT/prof_train_c4 6 12
fi
