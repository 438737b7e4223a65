#!/bin/bash
# round 5, call 9: the plain weight-gradient launch (gemm_tn3) on a side stream beside the main chain's short kernels (OCR_TN_SIDE=1): headline
# gradient parity with it on, then the step A/B (single GPU one-graph step, and the emulated two-rank schedule)
O=gpurun_out; mkdir -p $O; T=${1:-r05j}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OCR_TN_SIDE=1 timeout 600 python -m pytest tests/test_golden.py -q -m gpu -x -k "headline_fixture_gradients" 2>&1 | tail -2 | tee $O/${T}_tests.log
OCR_TN_SIDE=1 timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "train_step_parity or round4_fusions" 2>&1 | tail -2 | tee -a $O/${T}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; }
for v in 0 1 0 1 0 1; do
  OCR_TN_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "TN_SIDE=$v" | tee -a $O/${T}_ab.log
done
for v in 0 1; do
  OCR_FAKE_WORLD=2 OCR_TN_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "FAKE_WORLD=2 TN_SIDE=$v" | tee -a $O/${T}_ab.log
done
for v in 0 1; do
  OCR_TN_SIDE=$v timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | line "deep TN_SIDE=$v" | tee -a $O/${T}_ab.log
done
