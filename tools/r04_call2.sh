#!/bin/bash
# round 4, call 2: BN statistics from the conv_k3 epilogue, BN apply + pool, pooled gradient routing, optimiser tick merged into the norm pass:
# kernel tests, whole-graph parity, stress; then the plain weight-gradient GEMM sweep (splits x pipeline depth, experiments flavour) and a bench line.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "batchnorm or statistics or optimizer or small_ops or second_generation or maxpool" 2>&1 | tail -25 > $O/r04b_kernels.log; tail -4 $O/r04b_kernels.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stress.py -q -m gpu -x 2>&1 | tail -25 > $O/r04b_engine.log; tail -4 $O/r04b_engine.log
timeout 900 python -m pytest tests/test_golden.py tests/test_trained_fixture.py -q -m gpu -x 2>&1 | tail -25 > $O/r04b_golden.log; tail -4 $O/r04b_golden.log
for P in 0 3 4; do
  echo "== OCR_TN2_PIPE=$P" >> $O/r04b_tn_sweep.log
  OCR_NATIVE_LIB=$(pwd)/lstm_ctc_ocr_amd/libocrhip_exp.so OCR_TN2_PIPE=$P timeout 300 python tools/tn_split_sweep.py >> $O/r04b_tn_sweep.log 2>&1
done
cat $O/r04b_tn_sweep.log
timeout 400 python bench.py --no-cpu-baseline > $O/r04b_bench.json 2> $O/r04b_bench.err; python - <<'P'
import json
d = json.loads(open('gpurun_out/r04b_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print('bench', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms; conv frac', round(r['frac'], 4), 'clock', r['shader_clock_mhz'], 'exec frac at clock', r.get('executed_frac_of_peak_at_that_clock'))
P
OCR_FUSE_BN_STATS=0 OCR_FUSE_BN_POOL=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/r04b_bench_nofuse.json 2>/dev/null; python - <<'P'
import json
d = json.loads(open('gpurun_out/r04b_bench_nofuse.json').read().strip().splitlines()[-1])
print('bench without the BN fusions', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
P
