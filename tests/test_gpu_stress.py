"""-m gpu: race detector.  The forward pass contains no atomics, so repeated runs on the same batch must give bit-identical
logits and per-sample CTC costs; the backward pass accumulates with fp32 atomics, so its gradients must repeat to rounding.
(A hand-placed `s_waitcnt lgkmcnt` that was two reads short in one template instance of the convolution kernel passed every
parity test and showed up here as ~1 % of repetitions with garbage gradients.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.models import get_network


@pytest.mark.parametrize("N,W,ragged,reps", [(8, 88, False, 300), (8, 88, True, 300), (64, 256, False, 60)])
def test_repeated_runs_agree(dev, N, W, ragged, reps):
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    rng = np.random.RandomState(0)
    x = rng.rand(N, W, 32).astype(np.float32)
    sl = rng.randint(W // 8, W // 4, N).astype(np.int32) if ragged else np.full(N, W // 4 - 1, np.int32)
    ll = np.full(N, 4, np.int32)
    labels = rng.randint(1, 63, N * 4).astype(np.int32)
    ref = eng.forward(x, sl).clone()
    for i in range(reps):
        assert torch.equal(eng.forward(x, sl), ref), "forward repetition %d deviates" % i
    sp = eng.plan(N, W)
    eng._bind(sp, x, sl, labels, ll)
    eng._run(sp, 'fb')
    torch.cuda.synchronize()
    costs, grads = sp.costs.clone(), eng.grads.clone()
    assert bool(torch.isfinite(grads).all())
    scale = float(grads.abs().max())
    for i in range(reps):
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        assert torch.equal(sp.costs, costs), "costs of repetition %d deviate" % i
        assert float((eng.grads - grads).abs().max()) < 1e-3 * scale, "gradients of repetition %d deviate" % i
    for word in getattr(sp, 'lstm_sync', ()):             # error word of the persistent LSTM kernels: 1 = a wait timed out (0, or -1 in a
        assert int(word[-1].item()) != 1                   # block the step prepared inside its first kernel: nothing happened)
