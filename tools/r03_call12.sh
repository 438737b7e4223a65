#!/bin/bash
# round 3, call 12: in-step A/B of the convolution dispatch (conv_halo only / + tile D / + tiles A and D) and of the LSTM's counted
# top-of-step wait, interleaved in one call; LSTM parity tests; LSTM kernel timing.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "lstm" 2>&1 | tail -4 ) > $O/r03l_lstm_tests.log; tail -2 $O/r03l_lstm_tests.log
for cfg in "64 256" "32 512"; do set -- $cfg; timeout 60 python tools/lstm_bench.py --nb $1 --u $2 2>&1 | tail -1 | cut -c1-330; done | tee $O/r03l_lstm_bench.jsonl
line() { python - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
except Exception as e:
    print(sys.argv[1], 'no line', e)
P
}
for rep in 1 2 3; do
  OCR_CONV_K2=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03l_b_halo_$rep.json 2>/dev/null; line halo $O/r03l_b_halo_$rep.json
  timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03l_b_D_$rep.json 2>/dev/null; line D $O/r03l_b_D_$rep.json
  OCR_K2_TILES=AD timeout 150 python bench.py --no-cpu-baseline --no-roofline > $O/r03l_b_AD_$rep.json 2>/dev/null; line AD $O/r03l_b_AD_$rep.json
done | tee $O/r03l_bench_ab.log
