#!/bin/bash
# round 3, call 2: persistent LSTM at U = 256 / 512 (ring hand-off, 8-row tiles), parity + timing; the whole GPU suite; bench lines.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lstm" 2>&1 | tail -30 ) > $O/r03b_lstm_tests.log
tail -3 $O/r03b_lstm_tests.log
for cfg in "64 256" "32 512" "64 512"; do
  set -- $cfg
  for env in "OCR_LSTM_PROTO=2" "OCR_LSTM_PROTO=4" "OCR_LSTM_PROTO=4 OCR_LSTM_ROWS=8" "OCR_LSTM_PROTO=2 OCR_LSTM_ROWS=8" "OCR_LSTM_PROTO=0"; do
    env $env timeout 60 python tools/lstm_bench.py --nb $1 --u $2 2>&1 | tail -1 >> $O/r03b_lstm_bench.jsonl
  done
done
cut -c1-420 $O/r03b_lstm_bench.jsonl
( timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/r03b_pytest.log
tail -4 $O/r03b_pytest.log
for env in "OCR_LSTM_PROTO=2" "OCR_LSTM_PROTO=4" "OCR_LSTM_PROTO=2" "OCR_LSTM_PROTO=4"; do
  env $env timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done | tee $O/r03b_bench_ab.log
timeout 150 python bench.py --workload deep --no-cpu-baseline > $O/r03b_deep.json 2> $O/r03b_deep.err; tail -c 600 $O/r03b_deep.json
