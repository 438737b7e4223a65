#!/bin/bash
# round 6, final validation at one build (ocr_build_id): full GPU suite (incl. the hand-off stress cases and the round-4-rule detector), smoke, counter
# passes of all three workloads, the bench lines (headline with cpu_baseline + roofline incl. wgrad / per-launch decomposition, varwidth, deep), rocprof
# kernel summaries, the emulated data-parallel schedules (three graphs / one graph, without / with held CUs), a bs=128 side line, a long hand-off soak,
# the live-generator training loop and the reference's own entry point end to end.
#   usage (GPU box): bash tools/r06_final.sh [tag]
T=${1:-r06_final}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -12 ) 2>&1 | tee $O/${T}_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/${T}_smoke.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; w=r.get('wgrad') or {}; a=r.get('all_conv_mfma_work') or {}; print('$1', d['value'] and round(d['value']), 'img/s', d['ms_per_step'] and round(d['ms_per_step'],4), 'ms frac', r.get('frac'), 'wgrad', w.get('frac'), 'all', a.get('frac'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy_frac'), 'dropped', d.get('dropped_steps'), 'sched', (d.get('dp_schedule') or '')[:12], 'err', r.get('pmc_error'))"; }
bash tools/prof_step_pmc.sh ${T} 2>&1 | tail -14
bash tools/prof_step_pmc.sh ${T}_varwidth --workload varwidth 2>&1 | tail -3
bash tools/prof_step_pmc.sh ${T}_deep --workload deep 2>&1 | tail -3
cp $O/${T}_pmc_step_fixed.json $O/${T}_varwidth_pmc_step_varwidth.json $O/${T}_deep_pmc_step_deep.json profiles/ 2>/dev/null
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; line headline < $O/${T}_bench_full.json
timeout 200 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_driver_form.json 2>/dev/null; line "driver form (20 steps)" < $O/${T}_bench_driver_form.json
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null; line varwidth < $O/${T}_varwidth.json
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null; line deep < $O/${T}_deep.json
timeout 300 python bench.py --batch 128 --no-cpu-baseline > $O/${T}_batch128_side.json 2>/dev/null; line "SIDE bs=128" < $O/${T}_batch128_side.json
OCR_FAKE_WORLD=2 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2.json 2>/dev/null; line "FAKE_WORLD=2 three graphs" < $O/${T}_fake_world2.json
OCR_FAKE_WORLD=2 OCR_DP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2_onegraph.json 2>/dev/null; line "FAKE_WORLD=2 ONE graph" < $O/${T}_fake_world2_onegraph.json
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=16 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2_cus16.json 2>/dev/null; line "FAKE_WORLD=2 COMM_CUS=16 three" < $O/${T}_fake_world2_cus16.json
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=16 OCR_DP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_world2_cus16_onegraph.json 2>/dev/null; line "FAKE_WORLD=2 COMM_CUS=16 ONE" < $O/${T}_fake_world2_cus16_onegraph.json
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -16 $O/${T}_kernel_stats.md | cut -c1-130; tail -1 $O/${T}_kernel_stats.md
bash tools/prof_bench.sh ${T}_varwidth --no-roofline --workload varwidth --steps 100 > /dev/null 2>&1; tail -1 $O/${T}_varwidth_kernel_stats.md
bash tools/prof_bench.sh ${T}_deep --no-roofline --workload deep --steps 100 > /dev/null 2>&1; tail -1 $O/${T}_deep_kernel_stats.md
# hand-off soak on the product library: 100 000 graph launches per case beside the HBM-copy stream + the one-iteration skews
timeout 600 python tools/lstm_tail_race_probe.py --width 88 --short 1 --reps 100000 --skews "64:last,128:last,256:last,96:0,224:0" --skew-reps 300 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/${T}_handoff_soak.log
timeout 600 python tools/lstm_tail_race_probe.py --width 88 --short 4 --train --reps 50000 --skews "64:last,128:last,256:last,96:0,224:0" --skew-reps 300 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/${T}_handoff_soak.log
timeout 600 python tools/lstm_tail_race_probe.py --width 320 --ragged --train --reps 20000 --skews "64:-1,128:70,224:0" --skew-reps 100 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/${T}_handoff_soak.log
timeout 600 python tools/cli_throughput.py --iters 1500 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/${T}_cli_throughput_live.log
timeout 600 python tools/cli_throughput.py --iters 1500 --synth 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/${T}_cli_throughput_synth.log
timeout 300 python tools/synth_bench.py 2>&1 | grep -v "amdgpu.ids\|WARNING" | tee $O/${T}_synth_bench.log
bash tools/prof_synth_pmc.sh ${T} > /dev/null 2>&1; head -30 $O/${T}_synth_pmc.txt | cut -c1-120
timeout 300 python tools/lstm_timeout_probe.py --live --iters 20000 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee $O/${T}_live_soak_20k.log
( time timeout 900 ./train.sh --iters 40000 2>&1 | grep -E "^iter: *[0-9]*000 |accuracy|done solving|speed" | tail -60 ) > $O/${T}_train_cli_40k.log 2>&1; tail -6 $O/${T}_train_cli_40k.log
( time OCR_PIPELINE=ring timeout 900 ./train.sh --iters 40000 2>&1 | grep -E "^iter: *[0-9]*000 |accuracy|done solving|speed" | tail -60 ) > $O/${T}_train_cli_40k_pil_ring.log 2>&1; tail -6 $O/${T}_train_cli_40k_pil_ring.log
ls -la $O/${T}*pmc_step_*.json
