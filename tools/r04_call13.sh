#!/bin/bash
# round 4, call 13: conv1 + pool backward slab form: pooled pixels per block (OCR_CONV1_PPB); skip reasons of the conv1 tests
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -rs -k "conv1" 2>&1 | tail -4
for V in 256 128 64 512 256 128 64 512; do
  OCR_CONV1_PPB=$V timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCR_CONV1_PPB=$V', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')" | tee -a $O/r04m_conv1_slab_ppb_ab.log
done
for V in 128 64; do OCR_CONV1_PPB=$V bash tools/prof_bench.sh r04m_$V --no-roofline --steps 50 > /dev/null 2>&1; echo PPB=$V; grep -E "conv1_pool_bwd" $O/r04m_${V}_kernel_stats.md | cut -c1-140; done
