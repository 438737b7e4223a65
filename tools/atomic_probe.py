"""How expensive is the fp32 atomic epilogue of the weight-gradient kernel?  (measured: ~0.8 us per MB of partial sums;
a plain-store variant of the epilogue, tried once as a diagnostic, cost ~0.3 us per MB)  Tiny K (4 steps per workgroup) so that the
launch is dominated by `splits x output bytes` of atomics (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (W, H, Ci, Co) in [(64, 4, 512, 512), (64, 8, 256, 256)]:
    for s in (1, 2, 4, 8, 16):
        Nb = s * 256 // (W * H) if W * H <= 256 else s
        Nb = max(Nb, 1)
        x = torch.randn(Nb, W, H, Ci, device=dev).to(BF); y = torch.randn(Nb, W, H, Co, device=dev).to(BF)
        dw = torch.zeros(3, 3, Ci, Co, device=dev)
        us = timeit(lambda: ops.conv3x3_wgrad(x, y, dw, splits=s))
        mb = s * dw.numel() * 4 / 1e6
        print("Ci %d Co %d  pixels %d splits %d  atomics %.1f MB  %.1f us" % (Ci, Co, Nb * W * H, s, mb, us), flush=True)
