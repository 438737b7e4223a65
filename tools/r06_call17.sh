#!/bin/bash
# round 6, call 17: parameter workers as fresh interpreters (no fork of the training process) — synth tests with durations, throughput, train.sh
T=${1:-r06u}; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_synth.py -q --durations=6 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/${T}_synth_tests.log
timeout 600 python tools/cli_throughput.py --iters 1500 --synth --var 2>&1 | grep "^W88\|^W256\|^varwidth" | cut -c1-230 | tee $O/${T}_cli_throughput_synth.log
( time timeout 900 ./train.sh --iters 40000 2>&1 | grep -E "^iter: *[0-9]*0000 |accuracy|done solving|Error|Traceback" | tail -8 ) 2>&1 | tail -12 | tee $O/${T}_train_cli_40k.log
ps aux | grep "lstm_ctc_ocr_amd.utils.synth" | grep -v grep | wc -l | tee $O/${T}_leftover_workers.log
