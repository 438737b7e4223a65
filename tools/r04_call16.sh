#!/bin/bash
# round 4, call 16: four-wave persistent LSTM kernels, second version (results back through LDS into the wide store layout, operands loaded
# wide a step ahead) against the one-wave kernels (OCR_LSTM_KSPLIT=1)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=${1:-r04p}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -3
for K in 1 4; do
  OCR_LSTM_KSPLIT=$K timeout 120 python tools/lstm_bench.py 2>&1 | tail -1 | tee -a $O/${T}_lstm_bench.jsonl
  OCR_LSTM_KSPLIT=$K timeout 120 python tools/lstm_bench.py --nb 32 --u 512 2>&1 | tail -1 | tee -a $O/${T}_lstm_bench.jsonl
done
for cfg in 1 4 1 4; do
  OCR_LSTM_KSPLIT=$cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KSPLIT=$cfg', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/${T}_ab.log
done
