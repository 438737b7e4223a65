#!/bin/bash
# round 3, call 4: LDS-DMA fill-rate probe; the GPU tests added since call 2 (two-rank bench line, lagged loss reports, constant dropout,
# converted checkpoint, LSTM protocol 0 / 32-row tiles).
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 tools/bin/dma_probe > $O/r03d_dma_probe.txt 2>&1; cat $O/r03d_dma_probe.txt
( timeout 900 python -m pytest tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine.py tests/test_gpu_dsl.py tests/test_tf_bundle.py tests/test_gpu_kernels.py -m gpu -q -k "two_ranks or lagged or bias_job or constant_keep or converted or other_protocols" 2>&1 | tail -30 ) > $O/r03d_tests.log
tail -12 $O/r03d_tests.log
