#!/bin/bash
# round 3, call 13: convolution launches behind a cache scrub (operands from HBM, as inside the step): conv_halo vs conv_k2 tiles; s_setprio in-step A/B.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $1 $2" >> $O/r03m_conv.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.fwd','f').replace('.dgrad','d'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03m_conv.log; }
for rep in 1 2; do
  run "OCR_CONV_K2=0" "--cold"
  run "OCR_CONV_K2=1 OCR_K2_TILES=AD" "--cold"
  run "OCR_CONV_K2=0 OCR_HALO_PRIO=0" "--cold"
done
run "OCR_CONV_K2=0" ""
run "OCR_CONV_K2=1 OCR_K2_TILES=AD" ""
cat $O/r03m_conv.log
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
  OCR_HALO_PRIO=1 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03m_p1_$rep.json 2>/dev/null; line prio1 $O/r03m_p1_$rep.json
  OCR_HALO_PRIO=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03m_p0_$rep.json 2>/dev/null; line prio0 $O/r03m_p0_$rep.json
done | tee $O/r03m_prio_ab.log
