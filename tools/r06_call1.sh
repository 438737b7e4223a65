#!/bin/bash
# round 6, call 1: the hand-off closure on the GPU — stress cases beside an HBM-copy stream and with the skew hook, the same cases on the
# round-4 ring rule (experiments library: must FAIL), multi-rank drop with fault injection on one rank, per-step drop reports, the direct oracle
# assertion on ocr_lstm_fwd_seq_x; then a bench line (dropped_steps / lstm_timeouts; did the hooks cost anything?).
T=${1:-r06a}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
( time timeout 900 python -m pytest tests/test_gpu_stress.py -q -x -s 2>&1 | grep -E "HANDOFF|OLD-RULE|passed|failed|Error|error|assert" | tail -40 ) 2>&1 | tee $O/${T}_stress.log
timeout 900 python -m pytest tests/test_gpu_dp_two_ranks.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine.py tests/test_gpu_drivers.py -q -x 2>&1 | tail -15 | tee $O/${T}_dp_engine.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "lstm" 2>&1 | tail -8 | tee $O/${T}_lstm_kernels.log
timeout 300 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<P
import json
d = json.loads(open('$O/${T}_bench.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('bench', d['value'], d['ms_per_step'], 'dropped', d.get('dropped_steps'), 'timeouts', d.get('lstm_timeouts'), 'frac', r.get('frac'), 'err', r.get('pmc_error', '')[:60] if r.get('pmc_error') else None)
P
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -30 $O/${T}_kernel_stats.md | cut -c1-150
