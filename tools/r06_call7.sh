#!/bin/bash
# round 6, call 7: the whole GPU suite + smoke once more at HEAD (build 58d884a7a4c71cda; the only change since the validation call is the step-report
# test's block size), plus the driver-form bench line
T=${1:-r06g}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -8 ) 2>&1 | tee $O/${T}_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/${T}_smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_form.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/${T}_bench_driver_form.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('pmc_error'), d['dropped_steps'])"
