#!/bin/bash
# round 4, call 12: conv1 + pool backward without atomics (per-block slab rows + two jobs of the merged reduction) — OCR_CONV1_SLAB
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv1" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py tests/test_gpu_dsl.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python -m pytest tests/test_golden.py -q -m gpu -x -k "gradients or headline" 2>&1 | tail -2
for V in 1 0 1 0; do
  OCR_CONV1_SLAB=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCR_CONV1_SLAB=$V', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04l_conv1_slab_ab.log
done
bash tools/prof_bench.sh r04l --no-roofline > /dev/null 2>&1; grep -E "conv1_pool|reduce_jobs" $O/r04l_kernel_stats.md | cut -c1-140; tail -1 $O/r04l_kernel_stats.md
