"""Persistent BiLSTM recurrence timings (forward / backward, one launch each) with the phase stamps of workgroup 0.
    python tools/lstm_bench.py [--nb 64 --u 256 --t 63]      env: OCR_LSTM_PROTO (0 / 4), OCR_LSTM_ROWS (16 / 32 / 64)
One JSON line: us per launch, us per recurrent step, median phase times."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402
from lstm_ctc_ocr_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=64)
    ap.add_argument("--u", type=int, default=256)
    ap.add_argument("--t", type=int, default=63)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    Nb, U, T = a.nb, a.u, a.t
    assert ops.lstm_seq_supported(Nb, U)
    whT = (torch.randn(2, 4 * U, U, device=dev) * 0.05).to(BF)
    xproj = torch.randn(Nb * T, 8 * U, device=dev)
    sl = torch.full((Nb,), T, dtype=torch.int32, device=dev)
    hout = torch.zeros(Nb * T, 2 * U, dtype=BF, device=dev)
    gates = torch.zeros(2, Nb * T, 4 * U, device=dev); cell = torch.zeros(2, Nb * T, U, device=dev)
    whb = (torch.randn(2, 3 * U, 4 * U, device=dev) * 0.05).to(BF)
    dzb = torch.zeros(Nb * T, 8 * U, dtype=BF, device=dev)
    dh = (torch.randn(Nb * T, 2 * U, device=dev) * 0.01).to(BF)
    syn = torch.zeros(ops.lstm_seq_sync_words(Nb, U), dtype=torch.int32, device=dev)
    fwd = lambda: ops.lstm_fwd_seq(xproj, whT, sl, hout, gates, cell, Nb, T, U, syn)
    bwd = lambda: ops.lstm_bwd_seq(whb[:, 2 * U:], 4 * U, 3 * U * 4 * U, sl, dh, gates, cell, dzb, Nb, T, U, syn)
    res = {"nb": Nb, "u": U, "t": T, "proto": os.environ.get("OCR_LSTM_PROTO", "default"), "rows": os.environ.get("OCR_LSTM_ROWS", "default")}
    for nm, fn in (("fwd", fwd), ("bwd", bwd)):
        us = timeit(fn)
        res[nm + "_us"] = round(us, 1); res[nm + "_us_per_step"] = round(us / T, 3)
        assert int(syn[-1]) == 0, "spin time-out"
    dbg = torch.zeros(4 * T, dtype=torch.int64, device=dev)
    nat.call("ocr_lstm_seq_debug", dbg.data_ptr())
    for nm, fn in (("fwd", fwd), ("bwd", bwd)):
        fn(); torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(T, 4).astype(float) * 10.0      # wall_clock64 ticks are 10 ns
        st = d[5:-2]
        res[nm + "_phase_ns"] = {"wait": float(np.median(st[:, 1] - st[:, 0])), "mfma": float(np.median(st[:, 2] - st[:, 1])),
                                 "tail": float(np.median(st[:, 3] - st[:, 2])), "step": float(np.median(st[1:, 0] - st[:-1, 0]))}
    nat.call("ocr_lstm_seq_debug", None)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
