#!/bin/bash
# round 5, call 6: conv_ws as adopted (halo pieces at the top of the tile again; default policy = the single-chunk instance): forced parity,
# phase stamps, per-layer times and the step A/B, the headline golden tests on the default policy
O=gpurun_out; mkdir -p $O; T=${1:-r05f}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OCR_CONV_WS=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3_fwd_dgrad_wgrad or test_conv3x3_relu_pool" 2>&1 | tail -3 | tee $O/${T}_forced.log
OCR_CONV_WS=2 timeout 300 python tools/ws_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ws_phases.log
for ws in 0 1 0 1; do
  OCR_CONV_WS=$ws timeout 200 python tools/ws_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/WS=$ws /" | tee -a $O/${T}_ws_bench.log
done
for ws in 0 1; do
  OCR_CONV_WS=$ws timeout 200 python tools/ws_bench.py --cold 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/WS=$ws /" | tee -a $O/${T}_ws_bench.log
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms frac', r.get('frac'), 'avg_us', r.get('avg_launch_us'))"; }
for ws in 0 1 0 1; do
  OCR_CONV_WS=$ws timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line "WS=$ws" | tee -a $O/${T}_ab.log
done
timeout 900 python -m pytest tests/test_golden.py -q -m gpu -x -k "headline" 2>&1 | tail -4 | tee $O/${T}_golden.log
