#!/bin/bash
# round 5, call 16: the backward recurrence with the same rule (workgroups none of whose rows is active leave the ring alone): LSTM / engine / stress / ragged
# golden tests first; only if they pass: reproducer (training mode, sequences 6 steps shorter than T) on the previous and the new build, then the validation
O=gpurun_out; mkdir -p $O; T=${1:-r05q}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PREV=$(pwd)/lstm_ctc_ocr_amd/libocrhip_prev.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm" 2>&1 | tail -2 | tee $O/${T}_tests.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/${T}_tests.log
timeout 900 python -m pytest tests/test_golden.py -q -m gpu -x -k "ragged or headline_fixture_gradients" 2>&1 | tail -2 | tee -a $O/${T}_tests.log
if grep -q "failed\|error" $O/${T}_tests.log; then echo "TESTS FAILED: stopping"; exit 1; fi
OCR_NATIVE_LIB=$PREV timeout 300 python tools/lstm_tail_race_probe.py --reps 10000 --short 6 --train 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/prev /" | tee -a $O/${T}_race.log
timeout 300 python tools/lstm_tail_race_probe.py --reps 10000 --short 6 --train 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/new  /" | tee -a $O/${T}_race.log
rm -f $PREV
bash tools/r05_validate.sh r05_final2
