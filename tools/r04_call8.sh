#!/bin/bash
# round 4, call 8: gradient-buffer clear on the conv1 + pool forward launch (OCR_FUSE_ZERO)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py tests/test_gpu_dsl.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python -m pytest tests/test_golden.py -q -m gpu -x -k "gradients or headline" 2>&1 | tail -2
for V in 1 0 1 0; do
  OCR_FUSE_ZERO=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCR_FUSE_ZERO=$V', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04h_fuse_zero_ab.log
done
