"""Per-launch time of the convolution launches conv_ws.hip targets, in the form the train step runs them (conv2 forward with the fused
2 x 2 pool, conv2 data gradient with the ReLU mask, conv3_1 forward) plus the seven other 3 x 3 launches of the step, hot (back to back)
and cold (a 512 MB scrub between launches).  The kernel family is whatever the dispatcher picks under the environment (OCR_CONV_WS=0/1/2).
    python tools/ws_bench.py [--cold]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16
cold = '--cold' in sys.argv
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if cold else None
LAYERS = [('2f', 128, 16, 64, 128, (2, 2)), ('2d', 128, 16, 128, 64, None), ('3_1f', 64, 8, 128, 256, None), ('3_1d', 64, 8, 256, 128, None),
          ('3_2f', 64, 8, 256, 256, (1, 2)), ('3_2d', 64, 8, 256, 256, None), ('4_1f', 64, 4, 256, 512, None), ('4_1d', 64, 4, 512, 256, None),
          ('4_2f', 64, 4, 512, 512, None), ('4_2d', 64, 4, 512, 512, None)]
out, tot = [], 0.0
for name, W, H, Ci, Co, pool in LAYERS:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    b = torch.zeros(Co, device=dev)
    if name.endswith('d'):
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
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5 if cold else 1):
        if cold:
            scrub.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 1 if cold else 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    t = sorted(ts)[len(ts) // 2]
    tot += t
    out.append('%s %.1f (%s)' % (name, t, kn))
print(('cold ' if cold else 'hot  ') + ' '.join(out) + '  sum %.1f' % tot, flush=True)
