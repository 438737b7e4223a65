#!/bin/bash
# round 6, call 13: variable-width batches through the live loop (GPU synthesis and PIL ring), the synth tests incl. the variable-width stream
T=${1:-r06n}; O=gpurun_out; mkdir -p $O
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" 2>&1 | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -8 | tee $O/${T}_synth_tests.log
OCR_WIDTH_BUCKET=0 timeout 600 python tools/cli_throughput.py --iters 1500 --synth --only var 2>&1 | grep -v amdgpu.ids | grep "^varwidth" | cut -c1-330 | tee $O/${T}_cli_throughput_varwidth.log
timeout 600 python tools/cli_throughput.py --iters 1000 --only var 2>&1 | grep -v amdgpu.ids | grep "^varwidth" | cut -c1-330 | tee -a $O/${T}_cli_throughput_varwidth.log
