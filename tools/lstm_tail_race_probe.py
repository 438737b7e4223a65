"""Reproducer for the forward-recurrence ring race found in round 5: once every row of a batch tile is past its length, a workgroup of the
persistent LSTM polls nothing and runs free; its slot refill (and, three steps later, its payload) then landed in ring slots that a slower
workgroup, still at the tile's last active step, was reading -> a time-out (refill) or a silently wrong h (payload).  The same inference
forward (N = 64, W = 88: T = 21, every sequence `short` steps shorter) is repeated while a second stream keeps HBM busy; counted: launches
whose BiLSTM output differs from the first one, and launches that reported an expired wait.
    python tools/lstm_tail_race_probe.py [--reps 20000] [--short 4]"""
import argparse
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import _native as nat  # noqa: E402
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20000)
ap.add_argument('--short', type=int, default=4)
ap.add_argument('--width', type=int, default=88)
ap.add_argument('--train', action='store_true', help='forward + CTC + backward: compare the backward recurrence\'s dz too')
a = ap.parse_args()
dev = torch.device('cuda:0')
eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
T = a.width // 4 - 1
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.rand(64, a.width, 32).astype(np.float32)).to(dev)
sl = torch.full((64,), T - a.short, dtype=torch.int32, device=dev)
sp = eng.plan(64, a.width)
mode = 'fb' if a.train else 'fwd'
if a.train:
    ll = rng.randint(3, 5, 64).astype(np.int32)
    labels = rng.randint(1, 63, int(ll.sum())).astype(np.int32)
    eng._bind(sp, x, sl, labels, ll)
else:
    eng._bind(sp, x, sl)
eng._run(sp, mode)
torch.cuda.synchronize()
key = [k for k in sp.buf if k.endswith('/hout')][0]
keys = [key] + ([k for k in sp.buf if k.endswith('/dz') and 'logits' in k] if a.train else [])
refs = [sp.buf[k].clone() for k in keys]
ref = refs[0]
stop = threading.Event()
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)


def noise():
    with torch.cuda.stream(side):
        n = 0
        while not stop.is_set():
            big_b.copy_(big_a, non_blocking=True)
            n += 1
            if n % 8 == 0:
                side.synchronize()


th = threading.Thread(target=noise); th.start()
diff = torch.zeros(1, dtype=torch.int64, device=dev)
errs = 0
for i in range(a.reps):
    eng._run(sp, mode)
    bad = (sp.buf[key] != ref).any()
    for k, r in zip(keys[1:], refs[1:]):
        bad = bad | (sp.buf[k] != r).any()
    diff += bad.to(torch.int64)
    if i % 500 == 499:
        torch.cuda.synchronize()
        errs += sum(int(w[-1].item() == 1) for w in sp.lstm_sync)
stop.set(); th.join()
torch.cuda.synchronize()
print('build %s (%s): W=%d T=%d len=T-%d, %d runs beside an HBM-copy stream: BiLSTM output differed from the first launch in %d, expired waits seen at %d of %d checks'
      % (nat.build_id(), 'forward + CTC + backward: hout and dz compared' if a.train else 'inference forward: hout compared', a.width, T, a.short, a.reps, int(diff.item()), errs, a.reps // 500), flush=True)
