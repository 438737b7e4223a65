#!/bin/bash
# round 5, call 11: does tools/cli_throughput.py's W = 88 loop reproduce the persistent-LSTM time-out?  (five runs, projection fusion on / off)
O=gpurun_out; mkdir -p $O; T=${1:-r05l}
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { timeout 300 python tools/cli_throughput.py --iters 1500 --only W88 2>&1 | grep -E "NativeError|^W88" | cut -c1-260 | sed "s/^/$1 /" | tee -a $O/${T}_cli.log; }
run "FUSE_X=1 #1"; run "FUSE_X=1 #2"; run "FUSE_X=1 #3"
export OCR_LSTM_FUSE_X=0
run "FUSE_X=0 #1"; run "FUSE_X=0 #2"
unset OCR_LSTM_FUSE_X
timeout 300 python tools/cli_throughput.py --iters 1500 2>&1 | grep -E "NativeError|^W88|^W256" | cut -c1-260 | tee -a $O/${T}_cli.log
