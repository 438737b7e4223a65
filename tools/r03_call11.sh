#!/bin/bash
# round 3, call 11: the whole GPU suite at the new defaults, bench lines (headline, deep, varwidth, FAKE_WORLD=2), kernel stats, whole-step PMC.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/r03k_pytest.log
tail -5 $O/r03k_pytest.log | cut -c1-200
line() { python - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', 'roofline', round(r.get('frac', 0), 3), 'pmc', r.get('pmc_commit'), r.get('pmc_error'))
except Exception as e:
    print(sys.argv[1], 'no line', e)
P
}
timeout 150 python bench.py --no-cpu-baseline > $O/r03k_bench_a.json 2> $O/r03k_bench_a.err; line new $O/r03k_bench_a.json
OCR_CONV_K2=0 OCR_HALO_PRIO=0 timeout 150 python bench.py --no-cpu-baseline > $O/r03k_bench_oldconv.json 2>/dev/null; line oldconv $O/r03k_bench_oldconv.json
timeout 150 python bench.py --no-cpu-baseline > $O/r03k_bench_b.json 2>/dev/null; line new2 $O/r03k_bench_b.json
timeout 200 python bench.py --workload deep --no-cpu-baseline > $O/r03k_deep.json 2>/dev/null; line deep $O/r03k_deep.json
timeout 200 python bench.py --workload varwidth --no-cpu-baseline > $O/r03k_varwidth.json 2>/dev/null; line varwidth $O/r03k_varwidth.json
OCR_FAKE_WORLD=2 timeout 150 python bench.py --no-cpu-baseline > $O/r03k_fake_world2.json 2>/dev/null; line fakeworld2 $O/r03k_fake_world2.json
timeout 200 bash tools/prof_bench.sh r03k > $O/r03k_prof.log 2>&1; head -34 $O/r03k_kernel_stats.md | cut -c1-150
timeout 600 bash tools/prof_step_pmc.sh r03k > $O/r03k_pmc.log 2>&1; tail -18 $O/r03k_pmc.log | cut -c1-170
