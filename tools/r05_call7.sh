#!/bin/bash
# round 5, call 7: conv_k3 with the pixel fragments kept per plane (one read per dw group instead of one per tap): parity of every generation
# that runs conv_k3, then A/B against the previous build (lstm_ctc_ocr_amd/libocrhip_prev.so, built from HEAD~'s sources) in one call
O=gpurun_out; mkdir -p $O; T=${1:-r05h}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3" 2>&1 | tail -3 | tee $O/${T}_default.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv_kernel_generations and (env0 or env1)" 2>&1 | tail -3 | tee $O/${T}_gens.log
PREV=$(pwd)/lstm_ctc_ocr_amd/libocrhip_prev.so
for lib in prev new prev new; do
  if [ $lib = prev ]; then export OCR_NATIVE_LIB=$PREV; else unset OCR_NATIVE_LIB; fi
  timeout 200 python tools/ws_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$lib /" | tee -a $O/${T}_ws_bench.log
done
for lib in prev new; do
  if [ $lib = prev ]; then export OCR_NATIVE_LIB=$PREV; else unset OCR_NATIVE_LIB; fi
  timeout 200 python tools/ws_bench.py --cold 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$lib /" | tee -a $O/${T}_ws_bench.log
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms frac', r.get('frac'), 'avg_us', r.get('avg_launch_us'))"; }
for lib in prev new prev new; do
  if [ $lib = prev ]; then export OCR_NATIVE_LIB=$PREV; else unset OCR_NATIVE_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line "$lib" | tee -a $O/${T}_ab.log
done
unset OCR_NATIVE_LIB
