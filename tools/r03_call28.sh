#!/bin/bash
# round 3, call 28: conv_k3 general-width form (tiles crossing image boundaries; H = 2): parity through the conv tests with forced tiles, the
# engine / golden / trained-fixture tests (ragged batches), and A/B on the variable-width and deep workloads.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or conv_kernel_generations" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/r03ai_tests.log
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_golden.py tests/test_trained_fixture.py -q -x -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $O/r03ai_tests.log
for rep in 1 2; do for c in 0 1; do
  OCR_K3_GENW=$c timeout 120 python bench.py --workload varwidth --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('varwidth genw$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"
done; done | tee $O/r03ai_step.log
for c in 0 1; do
  OCR_K3_GENW=$c timeout 120 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep genw$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"
done | tee -a $O/r03ai_step.log
