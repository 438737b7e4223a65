#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python command of this repo -> gpurun_out/<tag>_kernel_stats.md
# usage (GPU box): bash tools/prof_cmd.sh <tag> tools/kernel_bench.py --only-conv
TAG=$1; shift; SCRIPT=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $REPO/$SCRIPT "$@" > $OUT/cmd.out 2> $OUT/cmd.err
python $REPO/tools/rocpd_stats.py $(ls $OUT/*trace*_results.db $OUT/*/*trace*_results.db 2>/dev/null | head -1) > $REPO/gpurun_out/${TAG}_kernel_stats.md 2>&1
rm -rf $OUT/*.db $OUT/*/*.db
head -24 $REPO/gpurun_out/${TAG}_kernel_stats.md | cut -c1-170
