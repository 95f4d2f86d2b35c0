#!/bin/bash
# Counter passes over a probe script: usage gpu_pmc_probe.sh <tag> "<counters>" <kernel-name-part> <script> [args...]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
TAG=$1; CTR=$2; KN=$3; shift 3
timeout -s KILL 120 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmcp_$TAG -- python3 "$@" > $OUT/pmcp_$TAG.log 2>&1 || { tail -5 $OUT/pmcp_$TAG.log; exit 1; }
python3 scripts/pmc_summary.py $OUT/pmcp_$TAG $KN
