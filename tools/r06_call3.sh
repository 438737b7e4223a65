#!/bin/bash
# round 6, call 3: the data-parallel schedule as ONE graph with captured collectives (OCR_DP_GRAPH=1) — RCCL capture on a 1-rank group, the emulated
# two-rank trajectory, and what the emulation (OCR_FAKE_WORLD=2, with and without held CUs) says about its cost against the three-graph schedule.
T=${1:-r06c}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 900 python -m pytest tests/test_gpu_drivers.py -q -x -s -k "rccl or emulated or residual" 2>&1 | grep -E "RCCL collectives|passed|failed|Error|error|assert" | tail -15 | tee $O/${T}_dp_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms', d.get('dp_schedule'), 'dropped', d.get('dropped_steps'))"; }
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_one_gpu_$rep.json 2>/dev/null; line "one graph, one GPU      " < $O/${T}_one_gpu_$rep.json
OCR_FAKE_WORLD=2 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_3g_$rep.json 2>/dev/null; line "FAKE_WORLD=2 three graphs" < $O/${T}_fake_3g_$rep.json
OCR_FAKE_WORLD=2 OCR_DP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_1g_$rep.json 2>/dev/null; line "FAKE_WORLD=2 ONE graph   " < $O/${T}_fake_1g_$rep.json
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=16 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_3g_cus16_$rep.json 2>/dev/null; line "FAKE 2 + 16 CUs, three   " < $O/${T}_fake_3g_cus16_$rep.json
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=16 OCR_DP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_1g_cus16_$rep.json 2>/dev/null; line "FAKE 2 + 16 CUs, ONE     " < $O/${T}_fake_1g_cus16_$rep.json
done 2>&1 | tee $O/${T}_fake_comm.log
