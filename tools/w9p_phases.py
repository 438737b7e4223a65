"""Wall-clock phase stamps of every workgroup of wgrad9p_kernel (experiments build): entry -> K loop done -> column sums done -> slab
stores issued -> acknowledged, against the event-timed launch (wgrad kernel + its slab reduction).
    OCR_NATIVE_LIB=.../libocrhip_exp.so python tools/w9p_phases.py      (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
for name, W, H, Ci, Co in [("conv4_2", 64, 4, 512, 512), ("conv4_1", 64, 4, 256, 512), ("conv3_2", 64, 8, 256, 256), ("conv3_1", 64, 8, 128, 256)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); dy = torch.randn(64, W, H, Co, device=dev).to(BF)
    dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
    ws = torch.empty(ops.conv3x3_wgrad_workspace_bytes(64, W, H, Ci, Co), dtype=torch.uint8, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, dy, dw, dbias=db, workspace=ws)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    dbg = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
    nat.call("ocr_wgrad9_debug", dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    nat.call("ocr_wgrad9_debug", None)
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    ph = (d[:, :5] - t0) / 100.0
    med = np.median(ph, axis=0); mx = ph.max(axis=0)
    print('%-8s %4d workgroups  wgrad + reduce %.1f us (events) | median entry %.1f loop done %.1f column sums %.1f stores issued %.1f acked %.1f | last acked %.1f'
          % (name, len(d), us, med[0], med[1], med[2], med[3], med[4], mx[4]))
