#!/bin/bash
# round 3, call 5: conv_k2 with 64-channel tiles (C 512 x 64, D 256 x 64): parity for every tile; per-layer timing of each forced tile, of
# the automatic choice and of conv_halo (with s_setprio), interleaved; stamps.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 700 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_kernel_generations" 2>&1 | tail -30 ) > $O/r03e_k2_tests.log
tail -3 $O/r03e_k2_tests.log
run() { echo "== $1" >> $O/r03e_conv.log
  env $1 timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
tot=0
for l in sys.stdin:
    d=json.loads(l); tot+=d['us']; print('%-14s %6.1f us %5.0f TF' % (d['kernel'], d['us'], d['tflops']))
print('sum %.1f us  -> %.0f TF avg (386.5 GF)' % (tot, 386.5e3/tot))" >> $O/r03e_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0 OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_K2_CFG=C OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_K2_CFG=A OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_K2_CFG=D OCR_HALO_PRIO=1"
  run "OCR_CONV_K2=1 OCR_HALO_PRIO=1"
done
cat $O/r03e_conv.log
OCR_CONV_K2=1 OCR_K2_CFG=C timeout 60 python tools/k2_stamps.py 2>&1 | grep -v amdgpu.ids > $O/r03e_k2_stamps_C.log; cat $O/r03e_k2_stamps_C.log | head -12
