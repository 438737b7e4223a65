#!/bin/bash
# ONE bounded gpurun call (round 2, last GPU minutes): the whole GPU test-suite at the new defaults, then bench.py at the new defaults
# against the previous launch schedule (the four launch-merging knobs of this commit series off), per-knob ablations, a kernel profile.
# Most important outputs first: the call may be cut by the remaining GPU budget.
#   usage (GPU box): bash tools/r02_final_check.sh <tag>
TAG=${1:-r02i}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
OLD="OCR_W9_DEFER=0 OCR_FUSE_FILLS=0 OCR_FUSE_PACK_BIAS=0 OCR_LSTM_AUX=0 OCR_W9_OVERLAP=0 OCR_COL2IM_V1=1"
date +%s > $O/${TAG}_t0
( timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -120 ) > $O/${TAG}_pytest.log
tail -3 $O/${TAG}_pytest.log
FAILED=$(grep -E "^(FAILED|ERROR) " $O/${TAG}_pytest.log | awk '{print $2}' | sort -u | head -12 | tr '\n' ' ')
if [ -n "$FAILED" ]; then          # which knob is it?  the failing tests again with each new default switched off, then with all of them off
    for k in OCR_W9_DEFER OCR_FUSE_PACK_BIAS OCR_W9_OVERLAP OCR_COL2IM_V1 ALL; do
        v=0; [ $k = OCR_COL2IM_V1 ] && v=1
        E="$k=$v"; [ $k = ALL ] && E="$OLD"
        ( env $E timeout 200 python -m pytest $FAILED -q 2>&1 | tail -6 ) > $O/${TAG}_pytest_no_$k.log
        echo "$k off: $(tail -1 $O/${TAG}_pytest_no_$k.log)"
    done
fi
timeout 150 python bench.py --no-cpu-baseline > $O/${TAG}_bench_new.json 2> $O/${TAG}_bench_new.err
env $OLD timeout 150 python bench.py --no-cpu-baseline > $O/${TAG}_bench_old.json 2> $O/${TAG}_bench_old.err
timeout 150 python bench.py --no-cpu-baseline > $O/${TAG}_bench_new2.json 2>> $O/${TAG}_bench_new.err
for f in new old new2; do python - <<E
import json
try:
    d = json.loads(open('$O/${TAG}_bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms', 'roofline', round(d['roofline']['frac'], 3))
except Exception as e:
    print('$f', 'no line', e)
E
done
timeout 200 bash tools/prof_bench.sh $TAG > $O/${TAG}_prof.log 2>&1
for k in OCR_W9_DEFER OCR_FUSE_PACK_BIAS OCR_W9_OVERLAP OCR_COL2IM_V1; do
    [ -n "$SHORT" ] && break
    v=0; [ $k = OCR_COL2IM_V1 ] && v=1
    env $k=$v timeout 150 python bench.py --no-cpu-baseline --steps 200 > $O/${TAG}_bench_no_$k.json 2> /dev/null
    python - <<E
import json
try:
    d = json.loads(open('$O/${TAG}_bench_no_$k.json').read().strip().splitlines()[-1])
    print('$k=$v', round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
except Exception as e:
    print('$k=$v', 'no line', e)
E
done
timeout 200 python bench.py --workload deep --no-cpu-baseline > $O/${TAG}_deep_new.json 2> $O/${TAG}_deep_new.err
env $OLD timeout 200 python bench.py --workload deep --no-cpu-baseline > $O/${TAG}_deep_old.json 2> $O/${TAG}_deep_old.err
timeout 200 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err
python - <<E
import json
for f in ('deep_new', 'deep_old', 'bench_line'):
    try:
        d = json.loads(open('$O/${TAG}_%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(d['value']), 'img/s', round(d['ms_per_step'], 4), 'ms')
    except Exception as e:
        print(f, 'no line', e)
E
echo done $(( $(date +%s) - $(cat $O/${TAG}_t0) )) s
