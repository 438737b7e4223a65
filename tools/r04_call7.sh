#!/bin/bash
# round 4, call 7: igemm tile variants on the four large plain GEMMs (experiments flavour knobs), engine tests after the test fix, kernel stats
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd tools
for V in 0 8 4; do OCR_NATIVE_LIB=$(pwd)/../lstm_ctc_ocr_amd/libocrhip_exp.so OCR_IG_NW=$V timeout 300 python gemm_nt_probe.py 2>&1 | grep -v amdgpu.ids | tee -a ../$O/r04g_gemm_nt_probe.log; done
cd ..
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -x 2>&1 | tail -2
bash tools/prof_bench.sh r04g --no-roofline > /dev/null 2>&1; tail -1 $O/r04g_kernel_stats.md; python - <<'P'
import json
d = json.loads(open('gpurun_out/r04g_bench_line.json').read().strip().splitlines()[-1]); print('profiled bench', round(d['value']), round(d['ms_per_step'], 4))
P
