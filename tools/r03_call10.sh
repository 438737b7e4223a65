#!/bin/bash
# round 3, call 10: conv_k2 with the fragment addresses generated in the load segment (OCR_K2_AIL=1) against beside the MFMAs (0).
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $1" >> $O/r03j_conv.log
  env $1 timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03j_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0 OCR_HALO_PRIO=1"
  for ail in 0 1; do
    run "OCR_CONV_K2=1 OCR_K2_CFG=A OCR_K2_AIL=$ail OCR_HALO_PRIO=1"
    run "OCR_CONV_K2=1 OCR_K2_CFG=D OCR_K2_AIL=$ail OCR_HALO_PRIO=1"
  done
done
cat $O/r03j_conv.log
( OCR_K2_AIL=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv_kernel_generations" 2>&1 | tail -3 )
