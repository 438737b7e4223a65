#!/bin/bash
# rocprofv3 --pmc passes over the GPU-side captcha synthesis alone (tools/synth_bench.py): HBM bytes, VALU / LDS instruction counts and busy cycles
# of captcha_synth_kernel — the numbers behind "bound by one CU's vector ALU per image".  Counters in their own runs with --kernel-trace only.
# usage (GPU box): bash tools/prof_synth_pmc.sh <tag>      -> gpurun_out/<tag>_synth_pmc.txt
TAG=${1:-rXX}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/${TAG}_synth_pmc.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $OUT -o pmc$i -- python $REPO/tools/synth_bench.py > $OUT/pmc$i.log 2>&1
  echo "== pass $i: $SET" >> $REPO/gpurun_out/${TAG}_synth_pmc.txt
  python $REPO/tools/rocpd_pmc.py $(ls $OUT/*pmc${i}_results.db $OUT/*/*pmc${i}_results.db 2>/dev/null | head -1) 2>&1 | grep -A12 "captcha_synth" >> $REPO/gpurun_out/${TAG}_synth_pmc.txt
done
rm -rf $OUT/*.db $OUT/*/*.db
cat $REPO/gpurun_out/${TAG}_synth_pmc.txt | cut -c1-160
