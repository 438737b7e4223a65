"""Where a conv_ws launch spends its time: every workgroup's first thread stamps the 100 MHz wall clock per phase and tile
(ocr_conv_ws_debug); printed: medians over the workgroups, in microseconds since the workgroup's entry; and a run-to-run determinism check.
    OCR_CONV_WS=2 python tools/ws_phases.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402
from lstm_ctc_ocr_amd import ops  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16
LAYERS = [('2f', 128, 16, 64, 128, (2, 2), False), ('2d', 128, 16, 128, 64, None, True), ('3_1f', 64, 8, 128, 256, None, False)]
NAMES = ['barrier', 'dma issued', 'k loop', 'landed', 'exchanged', 'written']
for name, W, H, Ci, Co, pool, dgrad in LAYERS:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    b = torch.randn(Co, device=dev)
    if dgrad:
        m = torch.randn(64, W, H, Co, device=dev).to(BF)
        fn = lambda: ops.conv3x3(x, w, out=y, mask=m)
        kn = ops.conv3x3_kernel_choice(64, W, H, Ci, Co, bias=False, relu=False, mask=True)
    elif pool:
        p = torch.empty(64, W // pool[0], H // pool[1], Co, dtype=BF, device=dev)
        fn = lambda: ops.conv3x3_relu_pool(x, w, y, p, b, pool[0], pool[1])
        kn = ops.conv3x3_kernel_choice(64, W, H, Ci, Co, pool=pool)
    else:
        fn = lambda: ops.conv3x3(x, w, out=y, bias=b, relu=True)
        kn = ops.conv3x3_kernel_choice(64, W, H, Ci, Co)
    # determinism: the same launch 30 times, every output bit-identical to the first
    fn(); torch.cuda.synchronize()
    first = y.clone(); bad = 0
    for _ in range(30):
        y.fill_(7.0)
        fn(); torch.cuda.synchronize()
        bad += int(not torch.equal(y, first))
    nblk = 256
    dbg = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    nat.call("ocr_conv_ws_debug", dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    nat.call("ocr_conv_ws_debug", None)
    d = dbg.cpu().numpy().reshape(nblk, 64).astype(np.float64)
    live = d[:, 0] > 0
    t0 = d[live, 0:1]
    rel = (d[live] - t0) / 100.0
    first_entry = (d[live, 0].min())
    print('%-5s %s: launch %.1f us, %d workgroups stamped, non-deterministic repeats %d/30; entry spread %.1f us; weights + first halo landed at %.1f us'
          % (name, kn, e0.elapsed_time(e1) * 1e3, int(live.sum()), bad, (d[live, 0].max() - first_entry) / 100.0, np.median(rel[:, 1])), flush=True)
    for i in range(10):
        base = 2 + 6 * i
        if base + 5 >= 64 or not (d[live, base] > 0).any():
            break
        ok = d[live, base] > 0
        vals = [np.median(rel[ok, base + k]) for k in range(6)]
        print('      tile %d: ' % i + '  '.join('%s %.2f' % (n, v) for n, v in zip(NAMES, vals)), flush=True)
    print('      last stamp (median over workgroups) %.1f us, slowest workgroup %.1f us' % (np.median(rel.max(axis=1)), rel.max()), flush=True)
