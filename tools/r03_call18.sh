#!/bin/bash
# round 3, call 18: wgrad9p (plane layout) parity + A/B against wgrad9_kernel, hot / cold / in-step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3_fwd_dgrad_wgrad" 2>&1 | tail -8 > $O/r03r_tests.log
cat $O/r03r_tests.log
run() { echo "== $1 $2" >> $O/r03r_wgrad.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "wgrad_slab" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.wgrad_slab','w'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03r_wgrad.log; }
for rep in 1 2; do
  run "OCR_W9_PLANES=0" "--cold"
  run "OCR_W9_PLANES=1" "--cold"
done
run "OCR_W9_PLANES=0" ""
run "OCR_W9_PLANES=1" ""
cat $O/r03r_wgrad.log
line() { python - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
except Exception as e:
    print(sys.argv[1], 'no line', e)
P
}
for rep in 1 2 3; do
  OCR_W9_PLANES=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03r_w0_$rep.json 2>/dev/null; line planes0 $O/r03r_w0_$rep.json
  OCR_W9_PLANES=1 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03r_w1_$rep.json 2>/dev/null; line planes1 $O/r03r_w1_$rep.json
done | tee $O/r03r_step_ab.log
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_golden.py -q -x -m gpu 2>&1 | tail -4
