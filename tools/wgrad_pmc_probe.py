"""The five 3x3 weight-gradient launches of one train step (batch 64, W=256), repeated, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
shapes = [(128, 16, 64, 128), (64, 8, 128, 256), (64, 8, 256, 256), (64, 4, 256, 512), (64, 4, 512, 512)]
if len(sys.argv) > 1:
    shapes = [shapes[int(a)] for a in sys.argv[1:]]
bufs = []
slab = os.environ.get("WGRAD_SLAB", "1") != "0"          # 1: nine-tap slab kernel (workspace form), 0: atomics kernels
for (W, H, Ci, Co) in shapes:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); y = torch.randn(64, W, H, Co, device=dev).to(BF)
    n = ops.conv3x3_wgrad_workspace_bytes(64, W, H, Ci, Co)
    ws = torch.empty(n, dtype=torch.uint8, device=dev) if (slab and n) else None
    bufs.append((x, y, torch.zeros(3, 3, Ci, Co, device=dev), torch.zeros(Co, device=dev), ws))
for it in range(4):
    for (x, y, dw, db, ws) in bufs:
        ops.conv3x3_wgrad(x, y, dw, dbias=db, workspace=ws)
torch.cuda.synchronize()
print("done")
