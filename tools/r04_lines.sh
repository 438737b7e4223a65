#!/bin/bash
# the bench lines of a finished validation run, re-taken once its counter summaries lie in profiles/ (bench.py reads the summary of the loaded build)
T=${1:-r04_final2}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; tail -c 1500 $O/${T}_bench_full.json; echo
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null
for f in bench_full varwidth deep; do python - $O/${T}_$f.json $f <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms frac', r.get('frac'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy_frac'), 'err', r.get('pmc_error'))
P
done
