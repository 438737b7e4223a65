#!/bin/bash
# round 3, call 17 / 20: conv_k3 (plane layout; call 20: with H = 16) parity + hot / cold / in-step A/B against conv_k2 and conv_halo.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or conv_kernel_generations" 2>&1 | tail -15 > $O/r03t_tests.log
cat $O/r03t_tests.log
run() { echo "== $1 $2" >> $O/r03t_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03t_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0" "--cold"
  run "OCR_CONV_K3=0" "--cold"
  run "OCR_CONV_K3=1" "--cold"
  run "OCR_CONV_K3=1 OCR_K3_MINSTEPS=18" "--cold"
done
run "OCR_CONV_K2=0" ""
run "OCR_CONV_K3=0" ""
run "OCR_CONV_K3=1" ""
run "OCR_CONV_K3=1 OCR_K3_MINSTEPS=18" ""
run "OCR_CONV_K3=1 OCR_K2_CFG=A" ""
run "OCR_CONV_K3=1 OCR_K2_CFG=D" ""
cat $O/r03t_conv.log
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
  OCR_CONV_K3=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03t_k2_$rep.json 2>/dev/null; line k2 $O/r03t_k2_$rep.json
  OCR_CONV_K3=1 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03t_k3_$rep.json 2>/dev/null; line k3 $O/r03t_k3_$rep.json
done | tee $O/r03t_step_ab.log
