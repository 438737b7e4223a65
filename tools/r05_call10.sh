#!/bin/bash
# round 5, call 10: the persistent-LSTM time-out seen once in cli_throughput.py (W = 88, every sequence one step shorter than T)
O=gpurun_out; mkdir -p $O; T=${1:-r05k}
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { timeout 300 python tools/lstm_timeout_probe.py "$@" 2>&1 | grep -v "amdgpu.ids\|WARNING" | tail -2 | tee -a $O/${T}_probe.log; }
run --iters 4000
run --iters 4000 --short 0
OCR_LSTM_FUSE_X=0 run --iters 4000
run --iters 1500 --live
run --iters 1500 --live
OCR_LSTM_FUSE_X=0 run --iters 1500 --live
OCR_FUSE_RINGFILL=0 run --iters 1500 --live
