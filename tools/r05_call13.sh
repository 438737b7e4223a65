#!/bin/bash
# round 5, call 13: ten runs of cli_throughput.py's W = 88 loop with diagnostics on a reported time-out (finite parameters / activations? lengths?)
O=gpurun_out; mkdir -p $O; T=${1:-r05n}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 200 python tools/cli_throughput.py --iters 1500 --only W88 2>&1 | grep -E "TIMEOUT-DIAG|^W88" | cut -c1-700 | sed "s/^/#$i /" | tee -a $O/${T}_cli.log
done
