#!/bin/bash
# round 6, call 2: the hand-off stress cases with the skew at ONE iteration (a delay in every iteration is absorbed by the other workgroups' adaptive
# pre-poll sleep: call 1 showed the round-4 rule surviving it), on the product library and on the round-4 rule (experiments library: must FAIL).
T=${1:-r06b}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
( time timeout 1200 python -m pytest tests/test_gpu_stress.py -q -s 2>&1 | grep -E "HANDOFF|OLD-RULE|NEW-RULE|passed|failed|Error|error|assert" | tail -40 ) 2>&1 | tee $O/${T}_stress.log
# where does the old rule break?  (a wider sweep, for the log)
E=lstm_ctc_ocr_amd/libocrhip_exp.so
for sk in "32:last,64:last,96:last,128:last,160:last,192:last,256:last,384:last,512:last"; do
  for short in 1 4; do
    OCR_NATIVE_LIB=$E OCR_LSTM_RING_RULE=always timeout 300 python tools/lstm_tail_race_probe.py --width 88 --short $short --reps 0 --skews $sk --skew-reps 10 --json 2>&1 | grep RESULT | tee -a $O/${T}_old_rule_sweep.log
  done
done
OCR_NATIVE_LIB=$E OCR_LSTM_RING_RULE=always timeout 300 python tools/lstm_tail_race_probe.py --width 88 --short 6 --train --reps 0 --skews "16:0,24:0,32:0,40:0,48:0,56:0,64:0,80:0,96:0,128:0,160:0,224:0" --skew-reps 10 --json 2>&1 | grep RESULT | tee -a $O/${T}_old_rule_sweep.log
timeout 200 python tools/lstm_bench.py 2>&1 | tail -6 | tee $O/${T}_lstm_bench.log
