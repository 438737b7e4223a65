#!/bin/bash
# round 4, call 6: batch-norm BACKWARD sums from the consumer's data-gradient epilogue (conv_k3b); slab-reduction flush threshold sweep
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "batchnorm or statistics or backward_sums" 2>&1 | tail -6 > $O/r04f_kernels.log; tail -3 $O/r04f_kernels.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py tests/test_gpu_dsl.py -q -m gpu -x 2>&1 | tail -4 > $O/r04f_engine.log; tail -2 $O/r04f_engine.log
timeout 900 python -m pytest tests/test_golden.py tests/test_trained_fixture.py -q -m gpu -x 2>&1 | tail -4 > $O/r04f_golden.log; tail -2 $O/r04f_golden.log
for V in "1 80" "0 80" "1 60" "1 40" "1 80" "0 80"; do
  set -- $V
  OCR_FUSE_BN_BWD=$1 OCR_W9_FLUSH_MB=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/r04f_bench_$1_$2.json 2>/dev/null
  python - $1 $2 <<'P'
import json, sys
d = json.loads(open('gpurun_out/r04f_bench_%s_%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print('OCR_FUSE_BN_BWD=%s OCR_W9_FLUSH_MB=%s' % (sys.argv[1], sys.argv[2]), round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
P
done
timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"
OCR_FUSE_BN_STATS=0 OCR_FUSE_BN_POOL=0 timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep without the BN fusions', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"
timeout 300 python bench.py --workload varwidth --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('varwidth', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"
