#!/bin/bash
# Copies the newest outputs of scripts/gpu_profile_round.sh a|b (gpurun_out/r3) into profiles/ under the r03_ names.
set -e
R=gpurun_out/r3
newest() { find "$1" -name "$2" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d' ' -f2; }
if [ -s $R/bench_default.json ]; then tail -n 1 $R/bench_default.json > profiles/r03_bench_default.json; fi
if [ -d $R/prof_retr ]; then cp "$(newest $R/prof_retr '*kernel_stats.csv')" profiles/r03_retrieval_bench_kernel_stats.csv; fi
if [ -d $R/prof_train ]; then
  cp "$(newest $R/prof_train '*kernel_stats.csv')" profiles/r03_train_step_kernel_stats.csv
  python3 scripts/kernel_groups.py $R/prof_train 6 40 > profiles/r03_train_step_groups.txt
fi
if [ -d $R/prof_train_c4 ]; then
  cp "$(newest $R/prof_train_c4 '*kernel_stats.csv')" profiles/r03_train_config4_kernel_stats.csv
  python3 scripts/kernel_groups.py $R/prof_train_c4 6 40 > profiles/r03_train_config4_groups.txt
fi
if [ -s $R/train_config4.json ]; then tail -n 1 $R/train_config4.json > profiles/r03_train_config4.json; fi
