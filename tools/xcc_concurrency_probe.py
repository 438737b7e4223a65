"""Does "workgroups with equal (id & 7) share an XCD" — what the persistent LSTM's in-L2 hand-off relies on — survive OTHER queues dispatching at the
same time?  Stream A launches the XCC probe with the LSTM launch's grid (128 workgroups x 256 threads) many times; stream B meanwhile issues what
the input pipeline issues: small pinned H2D copies (blit kernels) and one-workgroup kernels.  Every launch whose groups are split is counted.
    python tools/xcc_concurrency_probe.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402

dev = torch.device('cuda:0')
N, THR, REPS = 128, 256, 4000
out = torch.zeros(REPS, 2 * N, dtype=torch.int32, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode):
    out.zero_()
    torch.cuda.synchronize()
    stop = threading.Event()
    small_h = torch.zeros(256, dtype=torch.int32).pin_memory()
    big_h = torch.zeros(64 * 88 * 32, dtype=torch.uint8).pin_memory()
    small_d = torch.zeros(256, dtype=torch.int32, device=dev)
    big_d = torch.zeros(64 * 88 * 32, dtype=torch.uint8, device=dev)
    one = torch.zeros(2, dtype=torch.int32, device=dev)
    count = [0]

    def noise():
        with torch.cuda.stream(sb):
            while not stop.is_set():
                if mode in ('copies', 'both'):
                    small_d.copy_(small_h, non_blocking=True); big_d.copy_(big_h, non_blocking=True)
                if mode in ('kernels', 'both'):
                    nat.call("ocr_probe_xcc", one.data_ptr(), 1, 64, sb.cuda_stream)
                count[0] += 1
                if count[0] % 64 == 0:
                    sb.synchronize()
    th = None
    if mode != 'none':
        th = threading.Thread(target=noise); th.start()
    with torch.cuda.stream(sa):
        for i in range(REPS):
            nat.call("ocr_probe_xcc", out[i].data_ptr(), N, THR, sa.cuda_stream)
            if i % 256 == 255:
                sa.synchronize()
    sa.synchronize()
    stop.set()
    if th is not None:
        th.join()
    torch.cuda.synchronize()
    x = out[:, :N].cpu().numpy()
    split = sum(1 for r in x if any(len(set(r[k::8].tolist())) != 1 for k in range(8)))
    ident = sum(1 for r in x if all(int(v) == i % 8 for i, v in enumerate(r)))
    shifted = sum(1 for r in x if all(len(set(r[k::8].tolist())) == 1 for k in range(8)) and not all(int(v) == i % 8 for i, v in enumerate(r)))
    ex = next((r.tolist()[:24] for r in x if any(len(set(r[k::8].tolist())) != 1 for k in range(8))), None)
    print('noise=%-8s %d launches of %d x %d: groups split across XCDs in %d, id %% 8 == xcc in %d, uniformly shifted in %d; noise ops %d; first split example %s'
          % (mode, REPS, N, THR, split, ident, shifted, count[0], ex), flush=True)


for m in ('none', 'copies', 'kernels', 'both'):
    run(m)
