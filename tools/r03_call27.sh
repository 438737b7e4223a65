#!/bin/bash
# round 3, call 27: gemm_tn2 plain products with descriptor DMA (OCR_TN2_SRD=1) against flat pointers: parity + step A/B; conv tests after the conv_k3 clean-up.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_tn or lstm or conv3x3_relu_pool or conv3x3_accumulate" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/r03ah_tests.log
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $O/r03ah_tests.log
for rep in 1 2 3; do for c in 0 1; do
  OCR_TN2_SRD=$c timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn2_srd$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"
done; done | tee $O/r03ah_step.log
