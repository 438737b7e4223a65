"""Runs the 10 implicit-GEMM convolution launches of one train step (5 forward + 5 data gradient, batch 64, W=256) a few
times — the workload bench.py's roofline block times — so rocprofv3 --pmc can attribute HBM traffic to them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
shapes = [(128, 16, 64, 128), (64, 8, 128, 256), (64, 8, 256, 256), (64, 4, 256, 512), (64, 4, 512, 512)]
bufs = []
for (W, H, Ci, Co) in shapes:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); y = torch.randn(64, W, H, Co, device=dev).to(BF)
    wf = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF); wd = (torch.randn(Ci, 3, 3, Co, device=dev) * 0.05).to(BF)
    bufs.append((x, y, wf, wd, torch.zeros(Co, device=dev), torch.empty_like(y), torch.empty_like(x)))
for it in range(4):
    for (x, y, wf, wd, b, oy, ox) in bufs:
        ops.conv3x3(x, wf, out=oy, bias=b, relu=True)
        ops.conv3x3(y, wd, out=ox, mask=x)
torch.cuda.synchronize()
print("done")
