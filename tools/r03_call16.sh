#!/bin/bash
# round 3, call 16: s_memtime stamps of the current conv_k2 tile A (five stages) and its additive ablations (experiments build).
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
X=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
for abl in 8 9 10 12 15; do
  echo "== OCR_K2_ABL=$abl (8 stamps | 1 no DMA | 2 no fragment reads | 4 no MFMA)" >> $O/r03p_k2_stamps.log
  OCR_NATIVE_LIB=$X OCR_CONV_K2=1 OCR_K2_CFG=A OCR_K2_ABL=$abl timeout 120 python tools/k2_stamps.py >> $O/r03p_k2_stamps.log 2>&1
done
cat $O/r03p_k2_stamps.log
