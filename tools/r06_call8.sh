#!/bin/bash
# round 6, call 8: wgrad9p's general-width (zero-row) instances — parity through the convolution tests, whole-graph golden / engine tests, and an A/B of the
# variable-width workload against the previous build (libocrhip_prev.so = 58d884a7a4c71cda) and against the knob (OCR_W9P_GENW=0) in ONE call
T=${1:-r06h}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "test_conv3x3_fwd_dgrad_wgrad" 2>&1 | tail -4 | tee $O/${T}_conv_tests.log
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_engine.py tests/test_trained_fixture.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/${T}_graph_tests.log
timeout 900 python tools/ab_bench.py --tag ${T}_varwidth --rounds 3 --bench-args "--workload varwidth" new: knob_off:OCR_W9P_GENW=0 prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -5 | tee $O/${T}_ab_varwidth.log
timeout 600 python tools/ab_bench.py --tag ${T}_fixed --rounds 2 new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_fixed.log
bash tools/prof_bench.sh ${T}_varwidth --no-roofline --workload varwidth --steps 100 > /dev/null 2>&1; head -14 $O/${T}_varwidth_kernel_stats.md | cut -c1-120
