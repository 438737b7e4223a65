#!/bin/bash
# round 5, call 2: conv_ws.hip (weight-stationary persistent 3x3 convolution) — parity forced on every covered shape, then the default policy,
# per-layer times with the kernel off / on (hot and cold), the step A/B and the engine / golden tests on the default policy.
T=${1:-r05b}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
OCR_CONV_WS=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3_fwd_dgrad_wgrad or test_conv3x3_relu_pool" 2>&1 | tail -25 | tee $O/${T}_forced.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3_fwd_dgrad_wgrad or test_conv3x3_relu_pool" 2>&1 | tail -5 | tee $O/${T}_default.log
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
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_engine.py -q -m gpu -x 2>&1 | tail -4 | tee $O/${T}_engine.log
