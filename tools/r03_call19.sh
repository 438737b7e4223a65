#!/bin/bash
# round 3, call 19: wgrad9p look-ahead 2 / 3 / 4, conv2 with 128 splits, and a kernel trace separating wgrad kernels from their reductions.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $1 $2" >> $O/r03s_wgrad.log
  env $1 timeout 150 python tools/kernel_bench.py --only-conv $2 2>&1 | grep -E "wgrad_slab" | python -c "
import sys, json
ls=[json.loads(l) for l in sys.stdin]
print(' '.join('%s %.1f' % (d['kernel'].replace('conv','').replace('.wgrad_slab','w'), d['us']) for d in ls), ' sum %.1f' % sum(d['us'] for d in ls))" >> $O/r03s_wgrad.log; }
for rep in 1 2; do
  run "OCR_W9P_LA=2" ""
  run "OCR_W9P_LA=3" ""
  run "OCR_W9P_LA=4" ""
  run "OCR_W9_SMAX=128" ""
done
run "OCR_W9P_LA=2" "--cold"
run "OCR_W9P_LA=3" "--cold"
run "OCR_W9P_LA=4" "--cold"
run "OCR_W9_SMAX=128" "--cold"
cat $O/r03s_wgrad.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kb -o kb -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --only-conv > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_kb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P' | tee $O/r03s_kernel_bench_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r.get('Name', '')[:90], r.get('Calls'), r.get('AverageNs'), r.get('MinNs'), r.get('MaxNs'))
P
