#!/bin/bash
# round 6, call 12: GPU-side captcha synthesis after the stage-A changes — parity tests, live throughput, kernel time
T=${1:-r06l}; O=gpurun_out; mkdir -p $O
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" 2>&1 | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -15 | tee $O/${T}_synth_tests.log
timeout 600 python tools/cli_throughput.py --iters 1500 --synth 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330 | tee $O/${T}_cli_throughput_synth.log
bash tools/prof_cmd.sh ${T}_synth tools/cli_throughput.py --iters 300 --synth > /dev/null 2>&1
grep -i "captcha\|bind_batch\|^| kernel\|---" $O/${T}_synth_kernel_stats.md | cut -c1-170 | head -6
