#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py command -> gpurun_out/<tag>_kernel_stats.md (+ the bench line)
# usage (GPU box): bash tools/prof_bench.sh <tag> [bench.py args...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $REPO/bench.py --no-cpu-baseline "$@" > $REPO/gpurun_out/${TAG}_bench_line.json 2> $OUT/bench.err
python $REPO/tools/rocpd_stats.py $(ls $OUT/*trace*_results.db $OUT/*/*trace*_results.db 2>/dev/null | head -1) > $REPO/gpurun_out/${TAG}_kernel_stats.md 2>&1
rm -rf $OUT/*.db $OUT/*/*.db
head -30 $REPO/gpurun_out/${TAG}_kernel_stats.md | cut -c1-160
