#!/bin/bash
# round 6, call 15: the optimiser's norm pass on a grid that covers the buffer in one unrolled iteration, against the previous build as a second library
T=${1:-r06p}; O=gpurun_out; mkdir -p $O
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" 2>&1 | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -q -k "optim or adam or momentum or rmsprop or clip or guard or step" 2>&1 | tail -3 | tee $O/${T}_optim_tests.log
timeout 600 python tools/ab_bench.py --tag ${T}_fixed --rounds 4 new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_fixed.log
timeout 900 python tools/ab_bench.py --tag ${T}_deep --rounds 3 --bench-args "--workload deep" new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_deep.log
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; grep -i "optim_prep\|adam_update" $O/${T}_kernel_stats.md | cut -c1-130
