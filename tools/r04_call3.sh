#!/bin/bash
# round 4, call 3: gemm_tn3 (two plain weight-gradient products in one ping-pong launch): kernel test, whole-graph parity, A/B bench, kernel stats
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm_tn or statistics" 2>&1 | tail -25 > $O/r04c_kernels.log; tail -4 $O/r04c_kernels.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py tests/test_gpu_dsl.py -q -m gpu -x 2>&1 | tail -25 > $O/r04c_engine.log; tail -4 $O/r04c_engine.log
timeout 600 python -m pytest tests/test_golden.py -q -m gpu -x -k "gradients or headline or deep" 2>&1 | tail -25 > $O/r04c_golden.log; tail -4 $O/r04c_golden.log
for V in 1 0 1 0; do
  OCR_TN_JOBS=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/r04c_bench_tn$V.json 2>/dev/null
  python - $V <<'P'
import json, sys
d = json.loads(open('gpurun_out/r04c_bench_tn%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('OCR_TN_JOBS=%s' % sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
P
done
bash tools/prof_bench.sh r04c --no-roofline > /dev/null 2>&1; head -45 $O/r04c_kernel_stats.md | cut -c1-150
