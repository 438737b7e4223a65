#!/bin/bash
# round 6, call 11: GPU-side captcha synthesis — parity tests, kernel time, live throughput through the training loop
T=${1:-r06k}; O=gpurun_out; mkdir -p $O
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" 2>&1 | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -15 | tee $O/${T}_synth_tests.log
timeout 600 python tools/cli_throughput.py --iters 1500 --synth 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/${T}_cli_throughput_synth.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/${T}_prof -o synth -- python $GRAFT_REPO_ROOT/tools/cli_throughput.py --iters 300 --synth > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 $f; grep -i "captcha\|bind_batch" $f) | tee $O/${T}_synth_kernel_stats.txt
