#!/bin/bash
# round 3, call 14: tile A of conv_k2 with prefetch distance 3 (five stages) against distance 2, hot and behind a cache scrub; in-step A/B.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -3 > $O/r03n_tests.log
run() { echo "== $1 $2" >> $O/r03n_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03n_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0" "--cold"
  run "OCR_CONV_K2=1 OCR_K2_TILES=AD OCR_K2_NST=5" "--cold"
  run "OCR_CONV_K2=1 OCR_K2_TILES=AD OCR_K2_NST=4" "--cold"
  run "OCR_CONV_K2=1 OCR_K2_TILES=D" "--cold"
done
run "OCR_CONV_K2=0" ""
run "OCR_CONV_K2=1 OCR_K2_TILES=AD OCR_K2_NST=5" ""
run "OCR_CONV_K2=1 OCR_K2_TILES=AD OCR_K2_NST=4" ""
cat $O/r03n_conv.log
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
  OCR_CONV_K2=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03n_h_$rep.json 2>/dev/null; line halo $O/r03n_h_$rep.json
  OCR_CONV_K2=1 OCR_K2_TILES=D timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03n_d_$rep.json 2>/dev/null; line k2-D $O/r03n_d_$rep.json
  OCR_CONV_K2=1 OCR_K2_TILES=AD OCR_K2_NST=5 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03n_a_$rep.json 2>/dev/null; line k2-AD5 $O/r03n_a_$rep.json
done | tee $O/r03n_step_ab.log
cat $O/r03n_tests.log
