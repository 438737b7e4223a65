"""Conv weight-gradient timing sweep over split-M factors on both wgrad engines (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (Nb, W, H, Ci, Co) in [(64, 64, 4, 512, 512), (64, 64, 8, 256, 256), (64, 64, 8, 128, 256)]:
    x = torch.randn(Nb, W, H, Ci, device=dev).to(BF); y = torch.randn(Nb, W, H, Co, device=dev).to(BF)
    dw = torch.zeros(3, 3, Ci, Co, device=dev)
    fl = 2.0 * Nb * W * H * 9 * Ci * Co
    for eng in (0, 1):
        nat.call("ocr_set_wgrad_engine", eng)
        row = []
        for sp in (1, 2, 4, 8, 16, 32):
            us = timeit(lambda: ops.conv3x3_wgrad(x, y, dw, splits=sp))
            row.append("s%d:%.0fus(%.0fTF)" % (sp, us, fl / us / 1e6))
        print((Ci, Co, H), "engine", eng, " ".join(row), flush=True)
