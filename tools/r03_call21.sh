#!/bin/bash
# round 3, call 21: conv_k3 staged write-out: parity, phase stamps, hot / cold, in-step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or conv_kernel_generations" 2>&1 | tail -5 > $O/r03v_tests.log
cat $O/r03v_tests.log
OCR_NATIVE_LIB=$PWD/lstm_ctc_ocr_amd/libocrhip_exp.so timeout 200 python tools/k3_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/r03v_k3_phases.log
run() { echo "== $1 $2" >> $O/r03v_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03v_conv.log; }
run "OCR_CONV_K3=1" "--cold"
run "OCR_CONV_K3=1" "--cold"
run "OCR_CONV_K3=1" ""
run "OCR_CONV_K3=1" ""
cat $O/r03v_conv.log
for rep in 1 2 3; do
  timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
done | tee $O/r03v_step.log
