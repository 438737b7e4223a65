"""Prints the XCD (XCC_ID register) every workgroup of a 1-D grid lands on."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat
for n, thr in ((128, 64), (256, 64), (512, 256), (1000, 512)):
    out = torch.zeros(2 * n, dtype=torch.int32, device="cuda:0")
    nat.call("ocr_probe_xcc", out.data_ptr(), n, thr, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    x = out[:n].cpu().tolist()
    ok = sum(1 for i, v in enumerate(x) if v == i % 8)
    print("grid %d x %d threads: xcc of first 24 ids %s ; id %% 8 == xcc for %d / %d" % (n, thr, x[:24], ok, n))
