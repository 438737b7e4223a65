#!/bin/bash
# round 5, call 5: conv_ws after the prologue / write-out rework — parity forced, phase stamps, per-layer times, forward-repeat probe
O=gpurun_out; mkdir -p $O; T=${1:-r05e}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OCR_CONV_WS=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3_fwd_dgrad_wgrad or test_conv3x3_relu_pool" 2>&1 | tail -4 | tee $O/${T}_forced.log
OCR_CONV_WS=2 timeout 300 python tools/ws_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ws_phases.log
for ws in 0 1 0 1; do
  OCR_CONV_WS=$ws timeout 200 python tools/ws_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/WS=$ws /" | tee -a $O/${T}_ws_bench.log
done
OCR_CONV_WS=1 timeout 400 python tools/fwd_repeat_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/${T}_fwd_repeat_ws1.log
OCR_CONV_WS=0 timeout 400 python tools/fwd_repeat_probe.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/${T}_fwd_repeat_ws0.log
