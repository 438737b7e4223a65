#!/bin/bash
# round 5: validation of one build (ocr_build_id): full GPU suite, smoke, the three bench lines, kernel summary and the whole-step counter passes
# of ALL THREE workloads (bench.py refuses counters of another workload or build; tests/test_bench_host.py checks the committed ones on the CPU).
#   usage (GPU box): bash tools/r05_validate.sh <tag>
T=${1:-r05v}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -12 | tee $O/${T}_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/${T}_smoke.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms frac', r.get('frac'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy_frac'), 'err', r.get('pmc_error'))"; }
bash tools/prof_step_pmc.sh ${T} 2>&1 | tail -14
bash tools/prof_step_pmc.sh ${T}_varwidth --workload varwidth 2>&1 | tail -3
bash tools/prof_step_pmc.sh ${T}_deep --workload deep 2>&1 | tail -3
# the lines pick the counter summaries up from profiles/: copy this call's there first (the library is the same build)
cp $O/${T}_pmc_step_fixed.json $O/${T}_varwidth_pmc_step_varwidth.json $O/${T}_deep_pmc_step_deep.json profiles/ 2>/dev/null
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; line headline < $O/${T}_bench_full.json
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null; line varwidth < $O/${T}_varwidth.json
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null; line deep < $O/${T}_deep.json
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -16 $O/${T}_kernel_stats.md | cut -c1-130; tail -1 $O/${T}_kernel_stats.md
ls -la $O/${T}*pmc_step_*.json
