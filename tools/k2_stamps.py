"""s_memtime stamps of conv_k2_kernel (workgroup 0) per step and wave: LOAD = addresses + reads issue | DMA issue | waits + barrier,
COMP = MFMA | wait + barrier.  (The stamped workgroup's own stores sit in its vmcnt queue: its waits are longer than the others'.)
OCR_CONV_K2=1 [OCR_K2_CFG=C] python tools/k2_stamps.py   (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
for name, W, H, Ci, Co in [("conv4_2", 64, 4, 512, 512), ("conv3_2", 64, 8, 256, 256), ("conv2", 128, 16, 64, 128)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); wp = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    for _ in range(3): ops.conv3x3(x, wp, out=y, bias=b, relu=True)
    dbg = torch.zeros(8 * 80 * 6, dtype=torch.int64, device=dev)
    nat.call("ocr_conv_k2_debug", dbg.data_ptr())
    ops.conv3x3(x, wp, out=y, bias=b, relu=True); torch.cuda.synchronize()
    nat.call("ocr_conv_k2_debug", None)
    d = dbg.cpu().numpy().reshape(8, 80, 6).astype(np.float64)
    n = min(9 * Ci // 64, 80)
    print(name, 'steps', n)
    md = np.median
    for w in (0, 4, 1, 5):
        s = d[w, 2:n - 1]
        if not s[:, 0].any():
            print('  (no stamps: kernel not taken for this shape?)'); break
        print('  wave %d: addr+reads issue %.0f | DMA issue %.0f | waits+barrier %.0f | MFMA %.0f | vmwait+barrier %.0f | step %.0f' % (
            w, md(s[:, 1] - s[:, 0]), md(s[:, 2] - s[:, 1]), md(s[:, 3] - s[:, 2]), md(s[:, 4] - s[:, 3]), md(s[:, 5] - s[:, 4]),
            md(s[1:, 0] - s[:-1, 0])))
