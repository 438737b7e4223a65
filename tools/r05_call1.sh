#!/bin/bash
# round 5, call 1: the tree with round 4's parked LSTM patch adopted (input projection inside the persistent forward kernel), the accumulated
# clock stamps of the convolution kernels (bench.py --stamp-clock -> counter summaries over the kernels' OWN cycles), the CU-holding
# stand-in for RCCL (OCR_FAKE_COMM_CUS), tightened headline gradient bars and the torch-anchored conv1 + pool test:
# full GPU suite, A/B of the projection fusion, bench line, kernel summary, counter passes of all three workloads, fake-comm sweep.
T=${1:-r05a}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from lstm_ctc_ocr_amd import _native as n; print('build_id', n.build_id(), 'source', n.source_build_id())" | tee $O/${T}_build_id.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|Error" | tail -12 | tee $O/${T}_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/${T}_smoke.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; }
for cfg in 0 1 0 1; do
  OCR_LSTM_FUSE_X=$cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "FUSE_X=$cfg" | tee -a $O/${T}_ab.log
done
timeout 400 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench.err; tail -c 3000 $O/${T}_bench_full.json; echo
bash tools/prof_bench.sh ${T} --no-roofline > /dev/null 2>&1; head -14 $O/${T}_kernel_stats.md | cut -c1-130; tail -1 $O/${T}_kernel_stats.md
# one-GPU data-parallel emulation with CUs held on the communication stream (n workgroups x 250 us per 25 MB)
for cus in 0 8 16 32; do
  OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=$cus timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/${T}_fake_comm_${cus}.json 2>/dev/null
  line "FAKE_WORLD=2 COMM_CUS=$cus" < $O/${T}_fake_comm_${cus}.json | tee -a $O/${T}_fake_comm.log
done
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=0 bash tools/prof_bench.sh ${T}_fake0 --no-roofline --steps 100 > /dev/null 2>&1
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=16 bash tools/prof_bench.sh ${T}_fake16 --no-roofline --steps 100 > /dev/null 2>&1
OCR_FAKE_WORLD=2 OCR_FAKE_COMM_CUS=32 bash tools/prof_bench.sh ${T}_fake32 --no-roofline --steps 100 > /dev/null 2>&1
bash tools/prof_step_pmc.sh ${T} 2>&1 | tail -12
bash tools/prof_step_pmc.sh ${T}_varwidth --workload varwidth 2>&1 | tail -4
bash tools/prof_step_pmc.sh ${T}_deep --workload deep 2>&1 | tail -4
timeout 300 python bench.py --workload varwidth --no-cpu-baseline > $O/${T}_varwidth.json 2>/dev/null; line varwidth < $O/${T}_varwidth.json
timeout 300 python bench.py --workload deep --no-cpu-baseline > $O/${T}_deep.json 2>/dev/null; line deep < $O/${T}_deep.json
ls -la $O/${T}*pmc_step_*.json
