"""Split-count sweep of the two plain weight-gradient GEMMs of the headline step (gemm_tn2, atomics): conv5 (Mk 4032, I 2048, J 512)
and both BiLSTM directions' [x | h]^T dz (Mk 4032, I 768, J 1024, batch 2).   python tools/tn_split_sweep.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


R = 4032
# conv5: A = pooled activations [N, 64, 2*512] read as overlapping rows (row_group 63, skip 1), B = dY [R, 512]
x5 = torch.randn(64, 64, 1024, device=dev).to(BF); dy5 = torch.randn(R, 512, device=dev).to(BF)
dw5 = torch.zeros(2048, 512, device=dev); db5 = torch.zeros(512, device=dev)
xh = torch.randn(2, R, 768, device=dev).to(BF); dz = torch.randn(R, 2048, device=dev).to(BF)
dwl = torch.zeros(2, 768, 1024, device=dev); dbl = torch.zeros(2, 1024, device=dev)
for S in (0, 1, 2, 3, 4, 6, 8):
    t5 = timeit(lambda: ops.gemm_tn(x5, dy5, dw5, Mk=R, I=2048, J=512, lda=1024, ldb=512, ldo=512, row_group=63, row_skip=1,
                                    colsum=db5, splits=S))
    tl = timeit(lambda: ops.gemm_tn_batched(xh, 768, R * 768, dz, 2048, 1024, dwl, 1024, 768 * 1024, R, 768, 1024, 2, colsum=dbl,
                                            strideColsum=1024, splits=S))
    print('splits %d (0 = planner): conv5 wgrad %.1f us (%.0f TF)   lstm dW %.1f us (%.0f TF)' % (
        S, t5, 2.0 * R * 2048 * 512 / t5 / 1e6, tl, 2.0 * 2 * R * 768 * 1024 / tl / 1e6), flush=True)
