#!/bin/bash
# round 3, call 3: conv_k2 (128 x 64 wave tiles, in-workgroup K split, ping-pong K halves): parity, per-layer timing against
# conv_halo, segment stamps; s_setprio on conv_halo; stamps of the round-2 ping-pong kernel for comparison.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_kernel_generations" 2>&1 | tail -30 ) > $O/r03c_k2_tests.log
tail -3 $O/r03c_k2_tests.log
for env in "OCR_CONV_K2=0" "OCR_CONV_K2=1" "OCR_CONV_K2=0 OCR_HALO_PRIO=1" "OCR_CONV_K2=1 OCR_K2_FM=4" "OCR_CONV_K2=0" "OCR_CONV_K2=1"; do
  echo "== $env" >> $O/r03c_conv.log
  env $env timeout 120 python tools/kernel_bench.py --only-conv 2>&1 | grep -E "fwd|dgrad" | python -c "
import sys, json
tot=0
for l in sys.stdin:
    d=json.loads(l); tot+=d['us']; print('%-14s %6.1f us %5.0f TF' % (d['kernel'], d['us'], d['tflops']))
print('sum %.1f us  -> %.0f TF avg (386.5 GF)' % (tot, 386.5e3/tot))" >> $O/r03c_conv.log
done
cat $O/r03c_conv.log
OCR_CONV_K2=1 timeout 60 python tools/k2_stamps.py > $O/r03c_k2_stamps.log 2>&1; cat $O/r03c_k2_stamps.log | tail -20
OCR_GEMM_ENGINE=4 timeout 60 python tools/pp_stamps.py > $O/r03c_pp_stamps.log 2>&1; cat $O/r03c_pp_stamps.log | tail -12
( timeout 300 python -m pytest tests/test_golden.py tests/test_gpu_kernels.py -m gpu -q -k "deep_fixture or lstm" -s 2>&1 | grep -E "grad |deep fixture|passed|failed|Error" | tail -30 ) > $O/r03c_deep_lstm.log; cat $O/r03c_deep_lstm.log
