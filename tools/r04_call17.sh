#!/bin/bash
# round 4, call 17: the persistent LSTM launches' ring fills on the conv1 + pool forward launch (OCR_FUSE_RINGFILL=1, default) against one fill
# launch per LSTM launch (=0)
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=${1:-r04r}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm or conv1" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -x 2>&1 | tail -3
for cfg in 0 1 0 1; do
  OCR_FUSE_RINGFILL=$cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RINGFILL=$cfg', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/${T}_ab.log
done
OCR_FUSE_RINGFILL=0 timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep RINGFILL=0', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/${T}_ab.log
timeout 300 python bench.py --workload deep --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deep RINGFILL=1', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/${T}_ab.log
