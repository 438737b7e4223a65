#!/bin/bash
# round 5, call 8: merged slab reduction with non-temporal slab loads and the running gradient fetched before the barrier — the deferred-reduction
# bit-identity tests, then the step A/B against the previous build (lstm_ctc_ocr_amd/libocrhip_prev.so) and the kernel's time under rocprofv3
O=gpurun_out; mkdir -p $O; T=${1:-r05i}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "deferred_weight_gradient or round4_fusions" 2>&1 | tail -3 | tee $O/${T}_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_conv3x3_fwd_dgrad_wgrad" 2>&1 | tail -2 | tee -a $O/${T}_tests.log
PREV=$(pwd)/lstm_ctc_ocr_amd/libocrhip_prev.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; }
for lib in prev new prev new prev new; do
  if [ $lib = prev ]; then export OCR_NATIVE_LIB=$PREV; else unset OCR_NATIVE_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "$lib" | tee -a $O/${T}_ab.log
done
unset OCR_NATIVE_LIB
bash tools/prof_bench.sh ${T} --no-roofline --steps 100 > /dev/null 2>&1; grep -E "reduce_jobs|wgrad9p|adam" $O/${T}_kernel_stats.md | cut -c1-120
