#!/bin/bash
# round 3, call 15: tile D of conv_k2 with 4 / 5 / 6 weight stages (prefetch distance 2 / 3 / 4), hot and behind a cache scrub (experiments build).
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
X=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
run() { echo "== $1 $2" >> $O/r03o_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03o_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0" "--cold"
  run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=4" "--cold"
  run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=5" "--cold"
  run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=6" "--cold"
  run "OCR_NATIVE_LIB=$X OCR_K2_CFG=A" "--cold"
  run "OCR_K2_CFG=A" "--cold"
done
run "OCR_CONV_K2=0" ""
run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=4" ""
run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=5" ""
run "OCR_NATIVE_LIB=$X OCR_K2_CFG=D OCR_K2_NST=6" ""
run "OCR_NATIVE_LIB=$X OCR_K2_CFG=A" ""
run "OCR_K2_CFG=A" ""
cat $O/r03o_conv.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -3
