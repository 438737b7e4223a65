#!/bin/bash
# round 3, call 23: fused batch-norm passes (one launch, tensor kept in registers, two grid barriers): parity + A/B.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "batchnorm" 2>&1 | tail -4 | tee $O/r03z_tests.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -m gpu 2>&1 | tail -4 | tee -a $O/r03z_tests.log
for rep in 1 2; do
  for c in 0 1; do
  OCR_BN_FUSED=$c timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn_fused$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"
  done
done | tee $O/r03z_step.log
for c in 0 1; do
  OCR_BN_FUSED=$c timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep bn_fused$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"
done | tee -a $O/r03z_step.log
