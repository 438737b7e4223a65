#!/bin/bash
O=gpurun_out; mkdir -p $O; T=${1:-r05m}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/xcc_concurrency_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_xcc_concurrency.log
