"""conv4_2 / conv3_2 forward on the halo kernel, repeated, for rocprofv3 --pmc passes (where do the cycles go?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
for (W, H, Ci, Co) in [(64, 4, 512, 512), (64, 8, 256, 256)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF)
    wf = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF); b = torch.zeros(Co, device=dev)
    y = torch.empty(64, W, H, Co, device=dev, dtype=BF)
    for _ in range(5):
        ops.conv3x3(x, wf, out=y, bias=b, relu=True)
torch.cuda.synchronize()
print("done")
