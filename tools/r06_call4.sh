#!/bin/bash
# round 6, call 4: the renderer speed-up (bit-identical images) through the live training loop, and the GPU tests that touch the data path
T=${1:-r06d}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_drivers.py tests/test_trained_fixture.py -q 2>&1 | tail -4 | tee $O/${T}_tests.log
timeout 600 python tools/cli_throughput.py --iters 1500 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/${T}_cli_throughput_live.log
