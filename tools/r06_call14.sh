#!/bin/bash
# round 6, call 14 (no source change: evidence at the validated build): counters of the synthesis kernel, a 100 000-iteration training soak on
# synthesised batches, the whole GPU suite a second time
T=${1:-r06o}; O=gpurun_out; mkdir -p $O
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" 2>&1 | tee $O/${T}_build_id.txt
bash tools/prof_synth_pmc.sh $T 2>&1 | tail -60
( time timeout 600 ./train.sh --iters 100000 2>&1 | grep -E "^iter: *[0-9]*0000 |accuracy|done solving|Error|error|Traceback" | tail -30 ) > $O/${T}_train_cli_100k_synth.log 2>&1; tail -8 $O/${T}_train_cli_100k_synth.log
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -12 ) 2>&1 | tee $O/${T}_gpu_suite_second_run.log
