"""Diagnostic for a persistent-LSTM time-out seen once in tools/cli_throughput.py (W = 88 batches: T = 21, every sequence 20 steps long): the same
training loop (loss read one step behind) on (a) synthetic device-resident batches of that shape, (b) the live pipeline; reports the iteration of a
time-out, the losses before it and which hand-off block carried the error word.
    python tools/lstm_timeout_probe.py [--iters 3000] [--live]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402
from lstm_ctc_ocr_amd.config import cfg, cfg_from_file  # noqa: E402
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=3000)
ap.add_argument('--live', action='store_true')
ap.add_argument('--width', type=int, default=88)
ap.add_argument('--short', type=int, default=1, help='sequence length = T - short')
a = ap.parse_args()
cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lstm', 'lstm.yml'))
eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
eng.setup_optimizer('Adam', 1e-4)
T = a.width // 4 - 1
if a.live:
    from lstm_ctc_ocr_amd.utils.pipeline import DeviceBatchStream
    stream = DeviceBatchStream('cuda:0', 64, workers=None, pool=0)
    time.sleep(3.0)
    it = iter(stream)
    nxt = lambda i: next(it)
else:
    rng = np.random.RandomState(0)
    pool = []
    for _ in range(16):
        x = torch.from_numpy(rng.rand(64, a.width, 32).astype(np.float32)).cuda()
        ll = rng.randint(4, 7, 64).astype(np.int32)
        lab = torch.from_numpy(rng.randint(1, 63, int(ll.sum())).astype(np.int32)).cuda()
        pool.append((x, lab, torch.from_numpy(ll).cuda(), torch.full((64,), T - a.short, dtype=torch.int32).cuda()))
    nxt = lambda i: pool[i % 16]
pending, losses, t0 = None, [], time.time()
try:
    for i in range(a.iters):
        b = nxt(i)
        eng.train_step(*b, fetch_loss=False)
        h = eng.report_async()
        if pending is not None:
            losses.append(eng.report_wait(pending))
        pending = h
    losses.append(eng.report_wait(pending))
    nan_at = [i for i, v in enumerate(losses) if v != v]
    print('%s W=%d len=T-%d FUSE_X=%s RINGFILL=%s: %d iterations, %d steps dropped by the guard (device count %d; at iterations %s), loss %.3f -> %.3f, %.1f s' % (
        'live' if a.live else 'synthetic', a.width, a.short, os.environ.get('OCR_LSTM_FUSE_X', '1'), os.environ.get('OCR_FUSE_RINGFILL', '1'),
        len(losses), len(nan_at), int(eng.scalars[73].item()), nan_at[:8], losses[0], float(np.nanmean(losses[-20:])), time.time() - t0), flush=True)
except nat.NativeError as e:
    print('%s W=%d len=T-%d FUSE_X=%s RINGFILL=%s: TIME-OUT reported at iteration %d (the report is one step behind); last losses %s; %s' % (
        'live' if a.live else 'synthetic', a.width, a.short, os.environ.get('OCR_LSTM_FUSE_X', '1'), os.environ.get('OCR_FUSE_RINGFILL', '1'),
        len(losses), [round(v, 3) for v in losses[-6:]], str(e)[:160]), flush=True)
if a.live:
    stream.close()
