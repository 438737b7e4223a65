import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ['OCR_FAKE_WORLD'] = '2'
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network
sys.path.insert(0, '/root/repo/tests')
rng = np.random.RandomState(0)
N, W = 8, 88
x = rng.rand(N, W, 32).astype(np.float32); sl = np.full(N, W // 4 - 1, np.int32); ll = np.full(N, 4, np.int32); lab = rng.randint(1, 63, N * 4).astype(np.int32)
def grads(mutate):
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    eng.setup_optimizer('Adam', 0.0)
    if mutate == 'skip_early':
        orig = eng.allreduce_grads
        eng.allreduce_grads = lambda lo=0, hi=None: None if (lo, hi) == (0, eng.late_begin) else orig(lo, hi)
    if mutate == 'no_join':
        eng.comm_stream.wait_stream = lambda s: None
    eng.train_step(x, lab, ll, sl)
    return eng.grads.cpu().numpy().copy()
os.environ.pop('OCR_FAKE_WORLD'); base = grads(None); os.environ['OCR_FAKE_WORLD'] = '2'
for m in (None, 'skip_early'):
    g = grads(m)
    print(m, "max rel deviation from single-GPU gradient: %.3e" % (np.abs(g - base).max() / np.abs(base).max()))
