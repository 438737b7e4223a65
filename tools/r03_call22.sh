#!/bin/bash
# round 3, call 22: wgrad9p continuous read stream (OCR_W9P_CONT = 1: look-ahead 2, 2: look-ahead 5) against the per-step barrier form.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 1 2; do
  OCR_W9P_CONT=$c timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3_fwd_dgrad_wgrad" 2>&1 | tail -3 | sed "s/^/CONT=$c: /" >> $O/r03x_tests.log
done
cat $O/r03x_tests.log
run() { echo "== $1 $2" >> $O/r03x_wgrad.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "wgrad_slab" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.wgrad_slab','w'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03x_wgrad.log; }
for rep in 1 2; do
  run "OCR_W9P_CONT=0" ""
  run "OCR_W9P_CONT=1" ""
  run "OCR_W9P_CONT=2" ""
done
run "OCR_W9P_CONT=0" "--cold"
run "OCR_W9P_CONT=1" "--cold"
run "OCR_W9P_CONT=2" "--cold"
cat $O/r03x_wgrad.log
for rep in 1 2 3; do
  for c in 0 1 2; do
  OCR_W9P_CONT=$c timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cont$c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
  done
done | tee $O/r03x_step.log
