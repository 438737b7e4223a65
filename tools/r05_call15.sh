#!/bin/bash
# round 5, call 15: the forward-recurrence ring race (free-running workgroups refilled / overwrote slots a slower workgroup was reading): reproducer on
# the previous build (lstm_ctc_ocr_amd/libocrhip_prev.so = HEAD's sources) and on the fixed one, LSTM parity tests, step A/B
O=gpurun_out; mkdir -p $O; T=${1:-r05p}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PREV=$(pwd)/lstm_ctc_ocr_amd/libocrhip_prev.so
for short in 1 4; do
  OCR_NATIVE_LIB=$PREV timeout 300 python tools/lstm_tail_race_probe.py --reps 20000 --short $short 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/prev /" | tee -a $O/${T}_race.log
  timeout 300 python tools/lstm_tail_race_probe.py --reps 20000 --short $short 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/new  /" | tee -a $O/${T}_race.log
done
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -2 | tee $O/${T}_tests.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/${T}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; }
for lib in prev new prev new; do
  if [ $lib = prev ]; then export OCR_NATIVE_LIB=$PREV; else unset OCR_NATIVE_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "$lib" | tee -a $O/${T}_ab.log
done
unset OCR_NATIVE_LIB
