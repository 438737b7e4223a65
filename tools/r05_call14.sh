#!/bin/bash
O=gpurun_out; mkdir -p $O; T=${1:-r05o}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "timeout_drops or train_step_parity or round4_fusions" 2>&1 | tail -4 | tee $O/${T}_tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "optimizer or rmsprop" 2>&1 | tail -2 | tee -a $O/${T}_tests.log
timeout 600 python -m pytest tests/test_golden.py -q -m gpu -x -k "headline" 2>&1 | tail -2 | tee -a $O/${T}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; }
for v in 1 0 1 0; do
  OCR_LSTM_TIMEOUT_GUARD=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "GUARD=$v" | tee -a $O/${T}_ab.log
done
