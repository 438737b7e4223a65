#!/bin/bash
# round 5, call 3: conv_ws diagnostics — phase stamps per tile, run-to-run determinism, and the headline golden test with the kernel off / on
O=gpurun_out; mkdir -p $O; T=${1:-r05c}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OCR_CONV_WS=2 timeout 300 python tools/ws_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ws_phases.log
OCR_CONV_WS=0 timeout 300 python -m pytest tests/test_golden.py -q -m gpu -x -k "test_device_matches_headline_fixture and not gradients" 2>&1 | tail -3 | tee $O/${T}_golden_ws0.log
OCR_CONV_WS=1 timeout 300 python -m pytest tests/test_golden.py -q -m gpu -x -k "test_device_matches_headline_fixture and not gradients" 2>&1 | grep -E "^E|passed|failed" | cut -c1-300 | head -30 | tee $O/${T}_golden_ws1.log
