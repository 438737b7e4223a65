#!/bin/bash
# round 3, call 9: stamps of conv_k2 (LDS-held, no global stores in the pipeline), ablations 3 / 12; setprio off variant.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export OCR_NATIVE_LIB=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
OCR_CONV_K2=1 OCR_K2_CFG=A OCR_K2_ABL=8 timeout 60 python tools/k2_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/r03i_k2_stamps_A.log
OCR_CONV_K2=1 OCR_K2_CFG=D OCR_K2_ABL=8 timeout 60 python tools/k2_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/r03i_k2_stamps_D.log
run() { echo "== $1" >> $O/r03i_conv.log
  env $1 timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03i_conv.log; }
for abl in 0 3 8; do run "OCR_CONV_K2=1 OCR_K2_CFG=A OCR_K2_ABL=$abl"; done
cat $O/r03i_conv.log
