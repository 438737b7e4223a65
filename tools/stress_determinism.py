"""Race detector: the forward pass has no atomics, so its logits must be bit-identical from run to run.  Repeats the forward
(and the train step, compared with a tolerance) many times on the same batch and reports any deviation."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network
N, W = int(os.environ.get("N", 64)), int(os.environ.get("W", 256))
REPS = int(os.environ.get("REPS", 200))
eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
rng = np.random.RandomState(0)
x = rng.rand(N, W, 32).astype(np.float32)
sl = np.full(N, W // 4 - 1, np.int32)
if os.environ.get("RAGGED"):
    sl = rng.randint(W // 8, W // 4, N).astype(np.int32)
ref = eng.forward(x, sl).float().cpu()
bad = 0
for i in range(REPS):
    y = eng.forward(x, sl).float().cpu()
    if not torch.equal(y, ref):
        d = (y - ref).abs()
        bad += 1
        idx = torch.nonzero(d > 0)
        print("rep %d: %d elements differ, max %.3e, first at (t, n, c) = %s" % (i, idx.shape[0], float(d.max()), idx[0].tolist()), flush=True)
print("forward: %d / %d repetitions deviated" % (bad, REPS))

# forward+backward graph: per-sample CTC costs come from the forward half and must be bit-identical as well
L = 4
ll = np.full(N, L, np.int32)
labels = rng.randint(1, 63, N * L).astype(np.int32)
sp = eng.plan(N, W)
eng._bind(sp, x, sl, labels, ll)
eng._run(sp, 'fb'); torch.cuda.synchronize()
c0 = sp.costs.clone()
g0 = eng.grads.clone() if hasattr(eng, 'grads') else None
bad = 0
for i in range(REPS):
    eng._bind(sp, x, sl, labels, ll)
    eng._run(sp, 'fb'); torch.cuda.synchronize()
    if not torch.equal(sp.costs, c0):
        bad += 1
        print("fb rep %d: costs differ, max %.3e" % (i, float((sp.costs - c0).abs().max())), flush=True)
    elif g0 is not None:
        d = float((eng.grads - g0).abs().max()) / float(g0.abs().max())
        if d > 1e-3:
            bad += 1
            print("fb rep %d: gradients deviate by %.3e (relative to max)" % (i, d), flush=True)
print("fb: %d / %d repetitions deviated" % (bad, REPS))
