#!/bin/bash
# round 5, call 17: the tool that exposed the ring race (cli_throughput.py, W = 88 live loop; 3 reported time-outs in 21 runs before the fix) 16 more times at the final build
O=gpurun_out; mkdir -p $O; T=${1:-r05r}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id())" | tee $O/${T}_cli.log
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  timeout 120 python tools/cli_throughput.py --iters 1500 --only W88 2>&1 | grep -E "TIMEOUT-DIAG|dropped|WARNING: persistent|^W88" | cut -c1-200 | sed "s/^/#$i /" | tee -a $O/${T}_cli.log
done
