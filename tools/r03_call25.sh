#!/bin/bash
# round 3, call 25: conv_k3 MFMA priority of the wave group that multiplies first (OCR_K3_PRIO 1 = equal, 2, 3): hot / cold / in-step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $1 $2" >> $O/r03af_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03af_conv.log; }
for rep in 1 2; do for p in 1 2 3; do run "OCR_K3_PRIO=$p" ""; done; done
for p in 1 2 3; do run "OCR_K3_PRIO=$p" "--cold"; done
cat $O/r03af_conv.log
for rep in 1 2 3; do for p in 1 3; do
  OCR_K3_PRIO=$p timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio$p', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
done; done | tee $O/r03af_step.log
OCR_K3_PRIO=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3" 2>&1 | tail -2
