#!/bin/bash
# round 4, call 10: conv1 + pool backward: routed patch values from a private LDS copy (was 3 v_cndmask per tap and channel), code form on its own
# template instance (164 registers: 3 waves per SIMD instead of 2); forward code form at 3 or 4 waves per SIMD (OCR_CONV1_OCC4)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv1" 2>&1 | tail -2
OCR_CONV1_CODES=0 timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "parity or oracle_update" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py -q -m gpu -x 2>&1 | tail -2
for V in "1 1" "1 0" "0 1" "1 1" "1 0" "0 1"; do
  set -- $V
  OCR_CONV1_CODES=$1 OCR_CONV1_OCC4=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCR_CONV1_CODES=$1 OCR_CONV1_OCC4=$2', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04j_conv1_ab.log
done
bash tools/prof_bench.sh r04j --no-roofline > /dev/null 2>&1; grep -E "conv1_pool" $O/r04j_kernel_stats.md | cut -c1-140
OCR_CONV1_OCC4=0 bash tools/prof_bench.sh r04j_occ3 --no-roofline --steps 50 > /dev/null 2>&1; grep -E "conv1_pool" $O/r04j_occ3_kernel_stats.md | cut -c1-140
