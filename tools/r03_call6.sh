#!/bin/bash
# round 3, call 6: LDS-DMA fill-rate probe (second version: no divisions; 1-16 waves); conv_k2 timing ablations (experiments build).
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 tools/bin/dma_probe > $O/r03f_dma_probe.txt 2>&1; cat $O/r03f_dma_probe.txt
export OCR_NATIVE_LIB=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
run() { echo "== $1" >> $O/r03f_abl.log
  env $1 timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
print(' '.join('%s %.1f' % (json.loads(l)['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), json.loads(l)['us']) for l in sys.stdin))" >> $O/r03f_abl.log; }
for cfg in A C; do
  for abl in 0 1 2 3 4 5 6 7; do run "OCR_CONV_K2=1 OCR_K2_CFG=$cfg OCR_K2_ABL=$abl"; done
done
for abl in 0 1 2 5; do run "OCR_CONV_K2=0 OCR_HALO_ABL=$abl"; done
cat $O/r03f_abl.log
