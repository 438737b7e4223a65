"""Do two independent kernels overlap when issued on two streams (eagerly, and as parallel branches of one hipGraph)?
A = conv2's weight gradient (wgrad9: 128 workgroups of 147 KiB LDS, half the chip), B = conv2's data gradient (conv_halo, 1024
workgroups), C = a long latency-bound kernel (persistent LSTM forward, 128 one-wave workgroups).
    python tools/side_stream_probe.py      (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402

dev = torch.device('cuda:0'); BF = torch.bfloat16
Nb, W, H, Ci, Co = 64, 128, 16, 64, 128
x = torch.randn(Nb, W, H, Ci, device=dev).to(BF); dy = torch.randn(Nb, W, H, Co, device=dev).to(BF)
dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
ws = torch.empty(ops.conv3x3_wgrad_workspace_bytes(Nb, W, H, Ci, Co), dtype=torch.uint8, device=dev)
wd = (torch.randn(Ci, 3, 3, Co, device=dev) * 0.05).to(BF); dx = torch.empty_like(x)
T, U = 63, 256
whT = (torch.randn(2, 4 * U, U, device=dev) * 0.05).to(BF); xproj = torch.randn(Nb * T, 8 * U, device=dev)
sl = torch.full((Nb,), T, dtype=torch.int32, device=dev); hout = torch.zeros(Nb * T, 2 * U, dtype=BF, device=dev)
gates = torch.zeros(2, Nb * T, 4 * U, device=dev); cell = torch.zeros(2, Nb * T, U, device=dev)
syn = torch.zeros(ops.lstm_seq_sync_words(Nb), dtype=torch.int32, device=dev)

A = lambda: ops.conv3x3_wgrad(x, dy, dw, dbias=db, workspace=ws)
B = lambda: ops.conv3x3(dy, wd, out=dx, mask=x)
C = lambda: ops.lstm_fwd_seq(xproj, whT, sl, hout, gates, cell, Nb, T, U, syn)
side = torch.cuda.Stream(dev)


def seq(f, g):
    f(); g()


def par(f, g):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        f()
    g()
    main.wait_stream(side)


def timeit(body, iters=20, graph=False):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    run = body
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                body()
        run = g.replay
        run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if graph:
        run()
    else:
        for _ in range(iters):
            body()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, f, g in (('wgrad2 + dgrad2', A, B), ('wgrad2 + lstm_fwd', A, C), ('dgrad2 + lstm_fwd', B, C)):
    for graph in (False, True):
        ta, tb = timeit(f, graph=graph), timeit(g, graph=graph)
        ts, tp = timeit(lambda: seq(f, g), graph=graph), timeit(lambda: par(f, g), graph=graph)
        print('%-20s %s: alone %.1f + %.1f us, sequential %.1f us, two streams %.1f us' % (name, 'graph' if graph else 'eager', ta, tb, ts, tp), flush=True)
