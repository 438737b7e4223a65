#!/bin/bash
# round 3, call 24: conv_k3 staged write-out with the fused max-pool: parity; conv2 forward (9 K steps, fused 2 x 2 pool) on conv_k3 or conv_halo in the step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or conv_kernel_generations" 2>&1 | tail -3 | tee $O/r03ae_tests.log
X=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so
for rep in 1 2 3; do
  timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
  for t in 18 9; do
  OCR_NATIVE_LIB=$X OCR_K3_MINSTEPS=$t timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp-lib k3 minsteps $t', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
  done
done | tee $O/r03ae_step.log
