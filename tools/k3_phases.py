"""Wall-clock phase stamps of every workgroup of conv_k3_kernel (experiments build: make EXPERIMENTS=1 TARGET=../libocrhip_exp.so):
entry -> prologue landed -> K loop done -> K halves exchanged -> stores issued -> stores acknowledged, against the event-timed launch.
    OCR_NATIVE_LIB=.../libocrhip_exp.so python tools/k3_phases.py      (GPU box)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops, _native as nat
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = nat.lib(); lib.ocr_conv_k3_debug.argtypes = [ctypes.c_void_p]; lib.ocr_conv_k3_debug.restype = ctypes.c_int
for name, W, H, Ci, Co, mask in [("conv4_2.fwd", 64, 4, 512, 512, False), ("conv4_2.dgrad", 64, 4, 512, 512, True), ("conv4_1.fwd", 64, 4, 256, 512, False),
                                 ("conv3_2.fwd", 64, 8, 256, 256, False), ("conv3_1.fwd", 64, 8, 128, 256, False), ("conv3_1.dgrad", 64, 8, 256, 128, True)]:
    x = torch.randn(64, W, H, Ci, device=dev).to(BF); wp = (torch.randn(Co, 3, 3, Ci, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev); y = torch.empty(64, W, H, Co, dtype=BF, device=dev)
    mk = torch.randn(64, W, H, Co, device=dev).to(BF) if mask else None
    fn = (lambda: ops.conv3x3(x, wp, out=y, mask=mk)) if mask else (lambda: ops.conv3x3(x, wp, out=y, bias=b, relu=True))
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    nblk = 64 * W * H // 256 * (Co // 64)
    dbg = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    lib.ocr_conv_k3_debug(dbg.data_ptr())
    fn(); fn(); torch.cuda.synchronize()
    lib.ocr_conv_k3_debug(None)
    d = dbg.cpu().numpy().reshape(nblk, 8)
    d = d[d[:, 0] > 0]
    if not len(d):
        print(name, 'not on conv_k3'); continue
    t0 = d[:, 0].min()
    ph = (d[:, :6] - t0) / 100.0                       # us since the first workgroup's entry
    med = np.median(ph, axis=0); mx = ph.max(axis=0)
    print('%-14s %4d workgroups  launch %.1f us (events, back to back) | median entry %.1f prologue %.1f loop %.1f exchange %.1f stores issued %.1f acked %.1f | '
          'last workgroup acked %.1f' % (name, len(d), us, med[0], med[1], med[2], med[3], med[4], med[5], mx[5]))
