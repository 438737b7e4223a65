#!/bin/bash
# rocprofv3 --pmc passes over the ten convolution launches of a train step (tools/conv_probe.py) -> profiles-ready JSON.
# usage (GPU box): bash tools/prof_conv_pmc.sh <tag>      -> gpurun_out/<tag>_pmc_conv.json (+ _pmc_conv.txt)
TAG=${1:-rXX}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/${TAG}_pmc_conv.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $OUT -o pmc$i -- python $REPO/tools/conv_probe.py > $OUT/pmc$i.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(ls $OUT/*pmc${i}_results.db $OUT/*/*pmc${i}_results.db 2>/dev/null | head -1) conv_halo >> $REPO/gpurun_out/${TAG}_pmc_conv.txt 2>&1
done
rm -rf $OUT/*.db $OUT/*/*.db
python $REPO/tools/pmc_conv_summary.py $REPO/gpurun_out/${TAG}_pmc_conv.txt > $REPO/gpurun_out/${TAG}_pmc_conv.json
cat $REPO/gpurun_out/${TAG}_pmc_conv.json
