#!/bin/bash
# round 4, call 1: the new headline-size gradient parity tests + exact-shape conv tests, the two-rank bench (launcher and bare form),
# the MFMA-busy counter calibration, and a baseline bench line at the round's starting code.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$(pwd)
timeout 900 python -m pytest tests/test_golden.py -q -m gpu -s -k "gradients or headline" 2>&1 | tail -80 > $O/r04a_grad_parity.log; tail -5 $O/r04a_grad_parity.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_conv3x3_fwd_dgrad_wgrad or test_misc or second_generation" 2>&1 | tail -15 > $O/r04a_conv_shapes.log; tail -3 $O/r04a_conv_shapes.log
timeout 600 python -m pytest tests/test_gpu_bench_two_ranks.py tests/test_native_abi.py -q 2>&1 | tail -15 > $O/r04a_two_ranks.log; tail -3 $O/r04a_two_ranks.log
# --- counter calibration (own run, --kernel-trace only)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace -d $R/$O/prof_mfma -o cal -- python $R/tools/mfma_busy_probe.py launch $R/$O/r04a_mfma_probe_launches.json > $R/$O/r04a_mfma_probe_launch.log 2>&1 )
DB=$(ls $O/prof_mfma/*cal*_results.db $O/prof_mfma/*/*cal*_results.db 2>/dev/null | head -1)
python tools/rocpd_pmc.py $DB mfma_busy_probe --each > $O/r04a_mfma_probe_each.jsonl 2>&1
python tools/mfma_busy_probe.py report $O/r04a_mfma_probe_launches.json $O/r04a_mfma_probe_each.jsonl > $O/r04a_mfma_busy_calibration.md 2>&1; cat $O/r04a_mfma_busy_calibration.md
rm -rf $O/prof_mfma
timeout 400 python bench.py > $O/r04a_bench_full.json 2> $O/r04a_bench.err; tail -c 1800 $O/r04a_bench_full.json; echo
