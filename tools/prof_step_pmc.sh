#!/bin/bash
# rocprofv3 --pmc passes over bench.py itself (eager launches: --no-graphs, a few steps) -> per-kernel HBM bytes, matrix-pipe occupancy,
# L2 hit rate and LDS conflicts for EVERY kernel of the train step, pinned to the commit and to the kernel symbols of the library
# that ran.  Counters are collected in their own runs with --kernel-trace only (never with the trace domains gpurun refuses).
# usage (GPU box): bash tools/prof_step_pmc.sh <tag> [bench.py args, e.g. --workload deep]      -> gpurun_out/<tag>_pmc_step_<workload>.json (+ _pmc_step.txt)
TAG=${1:-rXX}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/${TAG}_pmc_step.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $OUT -o pmc$i -- python $REPO/bench.py --no-cpu-baseline --no-graphs --no-roofline --stamp-clock --steps 3 --warmup 2 "$@" > $OUT/pmc$i.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(ls $OUT/*pmc${i}_results.db $OUT/*/*pmc${i}_results.db 2>/dev/null | head -1) >> $REPO/gpurun_out/${TAG}_pmc_step.txt 2>&1
done
rm -rf $OUT/*.db $OUT/*/*.db
WL=fixed; for a in "$@"; do case "$prev" in --workload) WL=$a;; esac; prev=$a; done
python $REPO/tools/pmc_step_summary.py $REPO/gpurun_out/${TAG}_pmc_step.txt $REPO $WL $OUT/pmc3.log > $REPO/gpurun_out/${TAG}_pmc_step_$WL.json
python - <<P
import json
d = json.load(open('$REPO/gpurun_out/${TAG}_pmc_step_$WL.json'))
print('commit', d['commit'], 'kernels', len(d['kernels']))
for k in d['kernels'][:14]:
    print('%-46s n=%3d %8.1f us  rd %7.2f MB wr %7.2f MB  mfma %s  l2hit %s' % (k['short'][:46], k['launches'], k['avg_us'], k.get('read_mb') or 0, k.get('write_mb') or 0,
          None if k.get('mfma_busy_frac_own_cycles', k.get('mfma_busy_frac_grbm_window')) is None else round(k.get('mfma_busy_frac_own_cycles', k.get('mfma_busy_frac_grbm_window')), 3), None if k.get('l2_hit_rate') is None else round(k['l2_hit_rate'], 3)))
P
