#!/bin/bash
# round 3, call 26: wave-priority skew between the wave sets that share a SIMD, one kernel family at a time, inside the train step.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() { env $1 timeout 150 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', d.get('final_loss'))"; }
for rep in 1 2; do
  one "OCR_NONE=0"
  one "OCR_HALO_PRIO=2"
  one "OCR_W9P_PRIO=1"
  one "OCR_IG_PRIO=1"
  one "OCR_TN2_PRIO=1"
  one "OCR_HALO_PRIO=2 OCR_W9P_PRIO=1 OCR_IG_PRIO=1 OCR_TN2_PRIO=1"
done | tee $O/r03ag_prio_step.log
for c in "OCR_NONE=0" "OCR_HALO_PRIO=2 OCR_W9P_PRIO=1 OCR_IG_PRIO=1 OCR_TN2_PRIO=1"; do
  env $c timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep $c', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')"
done | tee -a $O/r03ag_prio_step.log
