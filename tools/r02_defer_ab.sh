#!/bin/bash
# bench-only A/B (about 7 s per run once torch is paged in): merged slab reduction forced on for the headline plan (its slabs are
# 162 MB, 2 MB above the default threshold) with and without the overlap branch, against the defaults and the previous schedule
O=gpurun_out; mkdir -p $O; TAG=${1:-r02l}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; torch.zeros(1, device='cuda')" > /dev/null 2>&1
run() { env "$@" timeout 60 python bench.py --no-cpu-baseline --steps 300 > $O/${TAG}_$NAME.json 2> /dev/null
        python -c "import json,sys; d=json.loads(open('$O/${TAG}_$NAME.json').read().strip().splitlines()[-1]); print('$NAME', round(d['value']), round(d['ms_per_step'],4))" 2>/dev/null || echo $NAME no line; }
for r in 1 2; do
NAME=default_$r run A=1
NAME=defer_$r run OCR_W9_DEFER_MAX_MB=400
NAME=defer_overlap_$r run OCR_W9_DEFER_MAX_MB=400 OCR_W9_OVERLAP=1
NAME=old_$r run OCR_W9_DEFER=0 OCR_FUSE_PACK_BIAS=0 OCR_COL2IM_V1=1
done
