#!/bin/bash
# round 4, call 5: L2 term folded into the update kernels (the norm pass stores nothing); slab reduction flushed while hot (OCR_W9_FLUSH_MB)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "optimizer or gemm_tn_jobs" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dp_two_ranks.py -q -m gpu -x 2>&1 | tail -2
for V in 0 80 120 0 80; do
  OCR_W9_FLUSH_MB=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/r04e_bench_flush$V.json 2>/dev/null
  python - $V <<'P'
import json, sys
d = json.loads(open('gpurun_out/r04e_bench_flush%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('OCR_W9_FLUSH_MB=%s' % sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
P
done
