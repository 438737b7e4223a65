#!/bin/bash
# bench-only probe of OCR_HALO_STAGGER=units,bit (conv_halo: start offset for every other workgroup of a CU), ~3.5 s per run
O=gpurun_out; mkdir -p $O; TAG=${1:-r02n}
run() { env "$@" timeout 40 python bench.py --no-cpu-baseline --steps 300 > $O/${TAG}_$NAME.json 2> /dev/null
        python -c "import json; d=json.loads(open('$O/${TAG}_$NAME.json').read().strip().splitlines()[-1]); print('$NAME', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['achieved']))" 2>/dev/null || echo $NAME no line; }
NAME=base_1 run A=1
for v in 16,32 64,32 128,32 16,1 64,1 128,1; do NAME=stag_${v/,/_} run OCR_HALO_STAGGER=$v; done
NAME=base_2 run A=1
