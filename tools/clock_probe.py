"""Shader clock the halo convolution kernel really runs at: workgroup 0 stamps the shader-clock counter (s_memtime) and the 100 MHz
wall clock (s_memrealtime) at entry and exit (ocr_conv_halo_clock_debug); MHz = d(shader) / d(wall) * 100.
    python tools/clock_probe.py            (GPU box; OCR_HALO_ABL=5 in the environment = MFMA-only ablation of the same kernel)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402
from lstm_ctc_ocr_amd import ops  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
nat.call("ocr_conv_halo_clock_debug", dbg.data_ptr())
for name, W, H, Ci, Co in (("conv2", 128, 16, 64, 128), ("conv3_2", 64, 8, 256, 256), ("conv4_2", 64, 4, 512, 512)):
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); wp = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    mhz, us = [], []
    for it in range(60):                       # back to back, like the layers of a step: the clock settles under sustained load
        ops.conv3x3(x, wp, out=y, bias=b, relu=True)
        if it >= 40:
            torch.cuda.synchronize()
            d = dbg.cpu().numpy()
            mhz.append((d[2] - d[0]) / max(1, d[3] - d[1]) * 100.0); us.append((d[3] - d[1]) / 100.0)
    print('%-8s ABL=%s: workgroup 0 lived %.1f us, shader clock %.0f MHz (min %.0f max %.0f)' % (
        name, os.environ.get('OCR_HALO_ABL', '0'), np.median(us), np.median(mhz), min(mhz), max(mhz)), flush=True)
nat.call("ocr_conv_halo_clock_debug", None)
