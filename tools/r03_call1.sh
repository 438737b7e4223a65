#!/bin/bash
# round 3, call 1: the three never-executed gated tests, then conv timings for the 16x16x32 and the 32x32x16 halo kernel.
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( OCR_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_kernels.py::test_ctc_tensorflow_known_answers tests/test_tf_bundle.py tests/test_halo_m32_model.py -m gpu -q -x 2>&1 | tail -60 ) > $O/r03a_gated.log
( OCR_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_tf_bundle.py tests/test_halo_m32_model.py -m gpu -q 2>&1 | tail -80 ) > $O/r03a_gated_all.log
tail -5 $O/r03a_gated_all.log
timeout 120 python tools/kernel_bench.py --only-conv > $O/r03a_conv16.log 2>&1
OCR_HALO_MFMA32=1 timeout 120 python tools/kernel_bench.py --only-conv > $O/r03a_conv32.log 2>&1
grep -E "fwd|dgrad" $O/r03a_conv16.log | cut -c1-120
echo ---- m32
grep -E "fwd|dgrad" $O/r03a_conv32.log | cut -c1-120
