"""Would the merged slab reduction (HBM-bound, 56 us) hide behind conv1 + pool backward (VALU-bound, 40 us) if both were ONE launch?
Upper bound from two streams outside any graph: each kernel alone, back to back on one stream, and concurrently on two streams
(the fork / join that made this a loss inside a captured hipGraph in round 2 does not exist for blocks of one launch).
    python tools/overlap_probe2.py        (GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd import ops  # noqa: E402
from lstm_ctc_ocr_amd.config import cfg  # noqa: E402
from lstm_ctc_ocr_amd.engine import Engine  # noqa: E402
from lstm_ctc_ocr_amd.models import get_network  # noqa: E402

dev = torch.device('cuda:0')
cfg.TRAIN.WEIGHT_DECAY = 1e-5
eng = Engine(get_network('LSTM_train'), device=dev, seed=3, use_graphs=False)
rng = np.random.RandomState(0)
N, W, L = 64, 256, 10
x = torch.from_numpy(rng.rand(N, W, 32).astype(np.float32)).to(dev)
labels = torch.from_numpy(rng.randint(1, 63, N * L).astype(np.int32)).to(dev)
ll = torch.full((N,), L, dtype=torch.int32, device=dev); sl = torch.full((N,), W // 4 - 1, dtype=torch.int32, device=dev)
sp = eng.plan(N, W)
eng._bind(sp, x, sl, labels, ll)
eng._run(sp, 'fb')
torch.cuda.synchronize()
ent = list(sp.w9_tables.values())[-1]
conv1 = [op for op in eng.ops if getattr(op, 'fused_pool', None) is not None][0]
print('reduce jobs', ent[1], 'blocks', ent[2], '| conv1 op', conv1.name)
side = torch.cuda.Stream(device=dev)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def both():
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.wgrad9_reduce_jobs(*ent)
    conv1.bwd(sp)
    main.wait_stream(side)


a = timed(lambda: conv1.bwd(sp))
b = timed(lambda: ops.wgrad9_reduce_jobs(*ent))
c = timed(lambda: (conv1.bwd(sp), ops.wgrad9_reduce_jobs(*ent)))
d = timed(both)
print('conv1 + pool backward alone %.1f us | merged slab reduction alone %.1f us | back to back %.1f us | two streams %.1f us' % (a, b, c, d))
