#!/bin/bash
# round 3, final validation at one commit (.build_commit): full GPU suite, the bench lines (headline with cpu_baseline + roofline, deep, varwidth,
# two emulated ranks), the rocprof kernel summary and the whole-step PMC passes.   usage (GPU box): bash tools/r03_final.sh [tag]
T=${1:-r03_final}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 | tee $O/${T}_gpu_suite.log
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; tail -c 2500 $O/${T}_bench_full.json; echo
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null
OCR_FAKE_WORLD=2 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2.json 2>/dev/null
for f in deep varwidth fake_world2; do python - $O/${T}_$f.json $f <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[2], 'no line', e)
P
done
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -30 $O/${T}_kernel_stats.md | cut -c1-130
bash tools/prof_step_pmc.sh ${T} 2>&1 | tail -18
