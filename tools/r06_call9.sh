#!/bin/bash
# round 6, call 9: batch-norm finalize inside the apply launches (small layers) — bit-identity against the three-launch form, the BN / DSL / deep-golden tests,
# and an A/B by the policy knob (OCR_BN_FIN_KB: 0 = never, 128 = default, 512 = the headline's layers too) on all three workloads, one call
T=${1:-r06i}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "batchnorm" 2>&1 | tail -3 | tee $O/${T}_bn_tests.log
timeout 1200 python -m pytest tests/test_gpu_dsl.py tests/test_golden.py tests/test_gpu_engine.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/${T}_graph_tests.log
timeout 900 python tools/ab_bench.py --tag ${T}_deep --rounds 3 --bench-args "--workload deep" kb128: kb0:OCR_BN_FIN_KB=0 kb64:OCR_BN_FIN_KB=64 kb512:OCR_BN_FIN_KB=512 2>&1 | tail -6 | tee $O/${T}_ab_deep.log
timeout 600 python tools/ab_bench.py --tag ${T}_fixed --rounds 2 kb128: kb0:OCR_BN_FIN_KB=0 kb512:OCR_BN_FIN_KB=512 2>&1 | tail -5 | tee $O/${T}_ab_fixed.log
timeout 600 python tools/ab_bench.py --tag ${T}_varwidth --rounds 2 --bench-args "--workload varwidth" kb128: kb0:OCR_BN_FIN_KB=0 2>&1 | tail -4 | tee $O/${T}_ab_varwidth.log
