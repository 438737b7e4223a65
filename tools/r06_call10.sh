#!/bin/bash
# round 6, call 10: the job-table kernels (pack_jobs, wgrad9_reduce_jobs) look their job up with all threads at once instead of a serial scan —
# whole-graph tests and an A/B against the previous build (libocrhip_prev.so = 0459b1dd46710fba) on all three workloads, one call
T=${1:-r06j}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_engine.py tests/test_gpu_dsl.py tests/test_gpu_drivers.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/${T}_graph_tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3_fwd_dgrad_wgrad or optimizer" 2>&1 | tail -2 | tee $O/${T}_kernel_tests.log
timeout 900 python tools/ab_bench.py --tag ${T}_deep --rounds 3 --bench-args "--workload deep" new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_deep.log
timeout 600 python tools/ab_bench.py --tag ${T}_fixed --rounds 3 new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_fixed.log
timeout 600 python tools/ab_bench.py --tag ${T}_varwidth --rounds 2 --bench-args "--workload varwidth" new: prev:OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_prev.so 2>&1 | tail -4 | tee $O/${T}_ab_varwidth.log
bash tools/prof_bench.sh ${T}_deep --no-roofline --workload deep --steps 60 > /dev/null 2>&1; grep -E "pack_jobs|reduce_jobs|adam" $O/${T}_deep_kernel_stats.md | cut -c1-120
