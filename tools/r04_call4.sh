#!/bin/bash
# round 4, call 4: optimiser tick reverted; gemm_tn3 with 5 stages against 4; would the slab reduction hide behind conv1 + pool backward?
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm_tn or optimizer" 2>&1 | tail -5 > $O/r04d_kernels.log; tail -2 $O/r04d_kernels.log
OCR_TN3_NST=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm_tn_jobs" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/overlap_probe2.py 2>&1 | tail -3 | tee $O/r04d_overlap_probe.log
for V in 5 4 5 4; do
  OCR_TN3_NST=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/r04d_bench_nst$V.json 2>/dev/null
  python - $V <<'P'
import json, sys
d = json.loads(open('gpurun_out/r04d_bench_nst%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('OCR_TN3_NST=%s' % sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
P
done
bash tools/prof_bench.sh r04d --no-roofline > /dev/null 2>&1; grep -E "gemm_tn3|optim_prep|optim_tick|reduce_jobs|conv1_pool_bwd" $O/r04d_kernel_stats.md | cut -c1-150
