#!/bin/bash
# round 4, final validation at one build (ocr_build_id): full GPU suite, the bench lines (headline with cpu_baseline + roofline, varwidth, deep,
# two emulated ranks), rocprof kernel summaries and the whole-step PMC passes of ALL THREE workloads (bench.py refuses counters of another
# workload or another build).     usage (GPU box): bash tools/r04_final.sh [tag]
T=${1:-r04_final}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -8 | tee $O/${T}_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; tail -c 2800 $O/${T}_bench_full.json; echo
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null
OCR_FAKE_WORLD=2 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2.json 2>/dev/null
for f in varwidth deep fake_world2; do python - $O/${T}_$f.json $f <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[2], 'no line', e)
P
done
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -12 $O/${T}_kernel_stats.md | cut -c1-130; tail -1 $O/${T}_kernel_stats.md
bash tools/prof_bench.sh ${T}_varwidth --no-roofline --workload varwidth --steps 100 > /dev/null 2>&1; tail -1 $O/${T}_varwidth_kernel_stats.md
bash tools/prof_bench.sh ${T}_deep --no-roofline --workload deep --steps 100 > /dev/null 2>&1; tail -1 $O/${T}_deep_kernel_stats.md
bash tools/prof_step_pmc.sh ${T} 2>&1 | tail -12
bash tools/prof_step_pmc.sh ${T}_varwidth --workload varwidth 2>&1 | tail -6
bash tools/prof_step_pmc.sh ${T}_deep --workload deep 2>&1 | tail -6
ls -la $O/${T}*pmc_step_*.json
# the real training loop through the live generator (one-channel renderer of round 4) and the shared-memory input ring, next to the device rate
timeout 600 python tools/cli_throughput.py --iters 1500 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/${T}_cli_throughput_live.log
# ... and the reference's own entry point end to end
timeout 900 ./train.sh --iters 40000 2>&1 | grep -E "^iter: *[0-9]*000 |accuracy|done solving|speed" | tail -60 > $O/${T}_train_cli_40k.log; tail -4 $O/${T}_train_cli_40k.log
