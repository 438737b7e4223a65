#!/bin/bash
# rocprofv3 passes over the five conv weight-gradient launches of a train step (tools/wgrad_pmc_probe.py).
# usage (on the GPU box): bash tools/prof_wgrad.sh <tag> [layer indices...]   -> gpurun_out/<tag>_*.txt
TAG=${1:-w9}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $REPO/tools/wgrad_pmc_probe.py "$@" > $OUT/trace.log 2>&1
python $REPO/tools/rocpd_stats.py $(ls $OUT/*trace*_results.db $OUT/*/*trace*_results.db 2>/dev/null | head -1) > $REPO/gpurun_out/${TAG}_kernel_stats.md 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $OUT -o pmc$i -- python $REPO/tools/wgrad_pmc_probe.py "$@" > $OUT/pmc$i.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(ls $OUT/*pmc${i}_results.db $OUT/*/*pmc${i}_results.db 2>/dev/null | head -1) wgrad >> $REPO/gpurun_out/${TAG}_pmc.txt 2>&1
  python $REPO/tools/rocpd_pmc.py $(ls $OUT/*pmc${i}_results.db $OUT/*/*pmc${i}_results.db 2>/dev/null | head -1) gemm_tn2 >> $REPO/gpurun_out/${TAG}_pmc.txt 2>&1
done
rm -rf $OUT/*.db $OUT/*/*.db
cat $REPO/gpurun_out/${TAG}_kernel_stats.md
