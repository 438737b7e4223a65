"""The four large plain GEMMs of the headline step (M = 4032 rows) on the igemm engine, hot and behind a cache scrub — run with the
experiments flavour and OCR_IG_NW = 8 / 4 to force the 256-row (8 waves, three stages) or the 128-row (4 waves, two workgroups per CU)
tiles:   OCR_NATIVE_LIB=lstm_ctc_ocr_amd/libocrhip_exp.so OCR_IG_NW=8 python tools/gemm_nt_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402
from kernel_bench import timeit, timeit_cold  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16
Nb, T = 64, 63
R = Nb * T
x5 = torch.randn(Nb, 64, 1024, device=dev).to(BF); w5 = (torch.randn(512, 2048, device=dev) * 0.05).to(BF)
xl = torch.randn(R, 512, device=dev).to(BF); wx = (torch.randn(2048, 512, device=dev) * 0.05).to(BF)
dz = torch.randn(R, 2048, device=dev).to(BF); wcat = (torch.randn(512, 2048, device=dev) * 0.05).to(BF)
d5 = torch.randn(R, 512, device=dev).to(BF); w5s = (torch.randn(2048, 512, device=dev) * 0.05).to(BF)
o512 = torch.empty(R, 512, dtype=BF, device=dev); o2048f = torch.empty(R, 2048, device=dev); o2048 = torch.empty(R, 2048, dtype=BF, device=dev)
cases = [
    ('conv5 forward   N=512  K=2048', lambda: ops.gemm_nt(x5, w5, out=o512, M=R, N=512, K=2048, ldp=1024, row_group=T, row_skip=1)),
    ('lstm x-proj     N=2048 K=512 ', lambda: ops.gemm_nt(xl, wx, out=o2048f)),
    ('lstm dX         N=512  K=2048', lambda: ops.gemm_nt(dz, wcat, out=o512)),
    ('conv5 dX (col)  N=2048 K=512 ', lambda: ops.gemm_nt(d5, w5s, out=o2048)),
]
print('OCR_IG_NW =', os.environ.get('OCR_IG_NW'))
for name, fn in cases:
    h, c = timeit(fn) * 1e6, timeit_cold(fn) * 1e6
    print('%s   hot %.1f us (%.0f TFLOP/s)   cold %.1f us' % (name, h, 2.0 * R * 512 * 2048 / h / 1e6, c), flush=True)
