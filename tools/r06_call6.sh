#!/bin/bash
# round 6, call 6: the four-wave LSTM kernels' LDS exchange arrays element-major (conflict-free) — parity, stress, stand-alone timings and an A/B of the
# whole step against the previous build (lstm_ctc_ocr_amd/libocrhip_prev.so = the validated build 2c1dbcbee66f3622) in ONE call
T=${1:-r06f}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "lstm" 2>&1 | tail -3 | tee $O/${T}_lstm_kernels.log
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_golden.py tests/test_gpu_engine.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/${T}_tests.log
for i in 1 2; do
  timeout 200 python tools/lstm_bench.py 2>&1 | tail -1 | cut -c1-400
  OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so timeout 200 python tools/lstm_bench.py 2>&1 | tail -1 | cut -c1-400
done | tee $O/${T}_lstm_bench.log
timeout 900 python tools/ab_bench.py --tag ${T} --rounds 3 new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -6 | tee $O/${T}_ab.log
timeout 600 python tools/ab_bench.py --tag ${T}_deep --rounds 2 --bench-args "--workload deep" new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_deep.log
