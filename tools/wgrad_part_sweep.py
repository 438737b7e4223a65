"""Sweep of the weight-gradient kernel's XCD partition (XS, XJ, XI) and split count per layer shape (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(128, 16, 64, 128), (64, 8, 128, 256), (64, 8, 256, 256), (64, 4, 256, 512), (64, 4, 512, 512)]
for (W, H, Ci, Co) in shapes:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); y = torch.randn(64, W, H, Co, device=dev).to(BF)
    dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
    pair = Ci == 64
    IT, JT = (1 if pair else Ci // 128), Co // 128
    tiles = (5 if pair else 9) * IT * JT
    res = []
    for XS in (8, 4, 2, 1):
        for XJ in (1, 2, 4, 8):
            if 8 % (XS * XJ): continue
            XI = 8 // XS // XJ
            if IT % XI or JT % XJ: continue
            for S in sorted(set(max(XS, (t // XS) * XS) for t in range(max(1, 200 // tiles), 620 // tiles + 2))):
                if S * tiles > 640 or S * tiles < 200: continue
                os.environ["OCR_TN2_PART"] = "%d,%d,%d,%d" % (XS, XJ, XI, S)
                res.append((timeit(lambda: ops.conv3x3_wgrad(x, y, dw, dbias=db)), XS, XJ, XI, S, S * tiles))
    res.sort()
    print((Ci, Co, H), "tiles", tiles, " | ".join("%.0fus xs%d xj%d xi%d S%d wg%d" % r for r in res[:6]), " ... worst %.0fus" % res[-1][0], flush=True)
