"""End-to-end check of the device-resident input stream on VARIABLE-WIDTH data (BASELINE configs[3]): trains LSTM_train on
captchas of 3..8 characters rendered at 50 px per character (W = 80 .. 220, one engine plan + hipGraphs per padded width), reading
every loss one iteration behind like the training driver, and reports loss and greedy sequence accuracy on fresh batches.
    python tools/train_stream_probe.py [--iters 8000]      (GPU box)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd.config import cfg  # noqa: E402
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402
from lstm_ctc_ocr_amd.utils.pipeline import DeviceBatchStream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=8000)
    a = ap.parse_args()
    cfg.TRAIN.SOLVER, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.WEIGHT_DECAY = 'Adam', 1e-4, 1e-5
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    eng.setup_optimizer()
    stream = DeviceBatchStream('cuda:0', 64, min_len=3, max_len=8, px_per_char=50)
    it = iter(stream)
    pending, losses, widths = None, [], set()
    t0 = time.time()
    for i in range(a.iters):
        pix, lab, ll, st = next(it)
        widths.add(int(pix.shape[1]))
        eng.train_step(pix, lab, ll, st, fetch_loss=False)
        h = eng.report_async()
        if pending is not None:
            losses.append(eng.report_wait(pending))
        pending = h
        if (i + 1) % 1000 == 0:
            print('iter %5d  loss (mean of last 200) %.4f  max of last 200 %.3f  plans %d  %.1f s' % (
                i + 1, float(np.mean(losses[-200:])), float(np.max(losses[-200:])), len(eng.plans), time.time() - t0), flush=True)
    losses.append(eng.report_wait(pending))
    ok = n = 0
    for _ in range(8):                                   # fresh batches, greedy decode (blank 0), whole-sequence accuracy
        pix, lab, ll, st = next(it)
        x = pix.float() / 255.0
        dec = eng.decode(x, st, method='greedy')
        lab, ll = lab.cpu().numpy(), ll.cpu().numpy()
        pos = 0
        for k, L in enumerate(ll):
            ok += int(list(lab[pos:pos + L]) == dec[k]); n += 1
            pos += L
    print('widths seen %d (%d..%d), plans %d, final loss %.4f, greedy sequence accuracy %d/%d = %.3f' % (
        len(widths), min(widths), max(widths), len(eng.plans), float(np.mean(losses[-200:])), ok, n, ok / n))
    stream.close()


if __name__ == '__main__':
    main()
