#!/bin/bash
# round 4, call 15: four-wave persistent LSTM kernels (OCR_LSTM_KSPLIT=4, default) against the one-wave kernels (=1); slab reduction with four
# float4 columns per thread (OCR_W9R_CPT=4, default) against one (=1).  Parity first, then the recurrence alone, then the whole step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3
for K in 1 4; do
  OCR_LSTM_KSPLIT=$K timeout 120 python tools/lstm_bench.py 2>&1 | tail -1 | tee -a $O/r04o_lstm_bench.jsonl
  OCR_LSTM_KSPLIT=$K timeout 120 python tools/lstm_bench.py --nb 32 --u 512 2>&1 | tail -1 | tee -a $O/r04o_lstm_bench.jsonl
done
for rep in 1 2; do
for cfg in "1 1" "4 1" "1 4" "4 4"; do
  set -- $cfg
  OCR_LSTM_KSPLIT=$1 OCR_W9R_CPT=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KSPLIT=$1 CPT=$2', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04o_ab.log
done
done
bash tools/prof_bench.sh r04o --no-roofline > /dev/null 2>&1; grep -E "lstm|reduce" $O/r04o_kernel_stats.md | cut -c1-150; tail -1 $O/r04o_kernel_stats.md
