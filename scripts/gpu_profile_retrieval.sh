#!/bin/bash
# Retrieval bench + rocprofv3 kernel stats + HBM traffic counter passes (each pass is a run of its own).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout -k 10 400 python3 -u bench.py --steps 3 --warmup 1 > $OUT/bench_retrieval.json 2> $OUT/bench_retrieval.err
tail -1 $OUT/bench_retrieval.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_retr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_retr.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 scripts/pmc_topk.py 262144 > $OUT/pmc_l2.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmc_fetch coarse > $OUT/pmc_summary.txt
python3 scripts/pmc_summary.py $OUT/pmc_write coarse >> $OUT/pmc_summary.txt
python3 scripts/pmc_summary.py $OUT/pmc_l2 coarse >> $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
