#!/bin/bash
# Last GPU minutes of round 2: the whole GPU suite at the FINAL defaults (merged slab reduction on one stream), bench A/B against the
# previous schedule and against the overlap variant, then the experimental in-kernel-fill LSTM protocol (OCR_LSTM_PROTO=3): its
# kernel tests, the kernel timings and a bench line.   usage (GPU box): bash tools/r02_last_call.sh <tag>
TAG=${1:-r02k}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
OLD="OCR_W9_DEFER=0 OCR_FUSE_PACK_BIAS=0 OCR_W9_OVERLAP=0 OCR_COL2IM_V1=1"
date +%s > $O/${TAG}_t0
line() { python - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', 'roofline', round(d['roofline']['frac'], 3))
except Exception as e:
    print(sys.argv[1], 'no line', e)
P
}
( timeout 330 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/${TAG}_pytest.log
tail -2 $O/${TAG}_pytest.log
timeout 100 python bench.py --no-cpu-baseline > $O/${TAG}_bench_new.json 2> $O/${TAG}_bench_new.err;              line new $O/${TAG}_bench_new.json
env $OLD timeout 100 python bench.py --no-cpu-baseline > $O/${TAG}_bench_old.json 2> /dev/null;                  line old $O/${TAG}_bench_old.json
env OCR_W9_OVERLAP=1 timeout 100 python bench.py --no-cpu-baseline > $O/${TAG}_bench_overlap.json 2> /dev/null;  line overlap $O/${TAG}_bench_overlap.json
timeout 100 python bench.py --no-cpu-baseline > $O/${TAG}_bench_new2.json 2> /dev/null;                          line new2 $O/${TAG}_bench_new2.json
( env OCR_LSTM_PROTO=3 timeout 150 python -m pytest tests/test_gpu_kernels.py -k "lstm_fwd_bwd" -q 2>&1 | tail -15 ) > $O/${TAG}_pytest_proto3.log
tail -1 $O/${TAG}_pytest_proto3.log
env OCR_LSTM_PROTO=3 timeout 100 python bench.py --no-cpu-baseline > $O/${TAG}_bench_proto3.json 2> $O/${TAG}_bench_proto3.err; line proto3 $O/${TAG}_bench_proto3.json
timeout 100 python bench.py --workload deep --no-cpu-baseline > $O/${TAG}_deep_new.json 2> /dev/null;            line deep_new $O/${TAG}_deep_new.json
env $OLD timeout 100 python bench.py --workload deep --no-cpu-baseline > $O/${TAG}_deep_old.json 2> /dev/null;   line deep_old $O/${TAG}_deep_old.json
env OCR_LSTM_PROTO=3 timeout 100 python bench.py --workload deep --no-cpu-baseline > $O/${TAG}_deep_proto3.json 2> /dev/null; line deep_proto3 $O/${TAG}_deep_proto3.json
timeout 120 bash tools/prof_bench.sh $TAG > $O/${TAG}_prof.log 2>&1
echo done $(( $(date +%s) - $(cat $O/${TAG}_t0) )) s
