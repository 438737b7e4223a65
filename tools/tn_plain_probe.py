"""Split sweep for the short-K plain weight gradients (LSTM dWx / dWh, conv5) on gemm_tn2 (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = 4032
for name, I, J in (("dWx", 512, 1024), ("dWh", 256, 1024), ("conv5", 2048, 512), ("dWx both dirs", 512, 2048)):
    A = torch.randn(R, I, device=dev).to(BF); B = torch.randn(R, J, device=dev).to(BF)
    out = torch.zeros(I, J, device=dev)
    row = []
    for S in (0, 1, 2, 3, 4, 6, 8, 12):
        us = timeit(lambda: ops.gemm_tn(A, B, out, splits=S))
        row.append("S%d %.1fus" % (S, us))
    print(name, (I, J), "tiles", I // 128 * (J // 128), " ".join(row), " (%.1f GF)" % (2.0 * R * I * J / 1e9), flush=True)
