#!/bin/bash
# round 3, call 7: conv_k2 third version (taps unrolled, next step's fragment addresses under the MFMAs): parity, timing, ablations.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 700 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv_kernel_generations" 2>&1 | tail -30 ) > $O/r03h_k2_tests.log
tail -3 $O/r03h_k2_tests.log
run() { echo "== $1" >> $O/r03h_conv.log
  env $1 timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03h_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0 OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_K2_CFG=A OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_K2_CFG=C OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_HALO_PRIO=1"
done
export OCR_NATIVE_LIB=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
for cfg in A C; do for abl in 0 1 2 4 7; do run "OCR_CONV_K2=1 OCR_K2_CFG=$cfg OCR_K2_ABL=$abl"; done; done
cat $O/r03h_conv.log
