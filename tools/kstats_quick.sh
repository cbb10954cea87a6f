#!/bin/bash
# quick per-kernel comparison against a committed profile: tools/kstats_quick.sh <out-tag> [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $REPO/gpurun_out/prof_$TAG.log 2>&1
cd $REPO
python3 tools/kstats_compare.py gpurun_out/prof_$TAG profiles/r04_bench_kernel_stats.csv > gpurun_out/kstats_$TAG.txt
tail -3 gpurun_out/prof_$TAG.log | cut -c1-300
