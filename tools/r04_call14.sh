#!/bin/bash
# round 4, call 14: conv1 forward with packed fp32 FMAs; rows per thread of the batch-norm apply passes (OCR_BN_ROWS)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv1 or batchnorm" 2>&1 | tail -2
for V in 4 8 16 4 8 16; do
  OCR_BN_ROWS=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCR_BN_ROWS=$V', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04n_bn_rows_ab.log
done
bash tools/prof_bench.sh r04n --no-roofline > /dev/null 2>&1; grep -E "conv1_pool|bn_" $O/r04n_kernel_stats.md | cut -c1-140
OCR_BN_ROWS=8 bash tools/prof_bench.sh r04n_8 --no-roofline --steps 50 > /dev/null 2>&1; grep -E "bn_" $O/r04n_8_kernel_stats.md | cut -c1-140
