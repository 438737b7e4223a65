#!/bin/bash
# round 6, call 5: stability of the timing-dependent tests — the stress file three more times, then the whole GPU suite once more at HEAD
T=${1:-r06e}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_stress.py -q 2>&1 | grep -E "passed|failed" | tail -1; done | tee $O/${T}_stress_x3.log
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -8 ) 2>&1 | tee $O/${T}_gpu_suite.log
