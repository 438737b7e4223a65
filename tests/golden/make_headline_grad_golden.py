"""Golden vectors for the BACKWARD half of the benchmarked step: the gradient of loss = mean CTC cost + L2 term and
the parameters after ONE clip + Adam step (lib/lstm/train.py:79-83 over lib/networks/LSTM_train.py:22-38), from the fp32 AND the
bf16-simulating CPU oracle (oracle/graph.py), at the two sizes whose kernels the small-shape tests do not reach:

  headline  BASELINE configs[1]: N = 64, W = 256 (T = 63), 10-character labels   — the inputs of graph_c2.npz
  ragged    BASELINE configs[3]: N = 64, W_i in [80, 320] padded to the batch maximum (320, T = 79), 4..10-character labels,
            per-sample time_step_len = W_i // 4 - 1 (lib/lstm/utils/gen.py:54)

At these sizes the dispatcher picks full-chip conv_k3 tiles, wgrad9p with 64 pixel splits and the deferred slab reduction, none of which
the N = 8, W = 88 whole-graph tests run.  Parameters: the seeded draw of graph_small.npz (og.init_params(seed=11) + its stored biases /
BN affine).  Per parameter tensor the file keeps the L2 norm of the gradient and a seeded sample of GRAD_SAMPLE entries (the full
gradient is 28.6 MB per oracle), the same entries of the post-Adam parameters, the clip norm and the loss.

    python tests/golden/make_headline_grad_golden.py      -> tests/golden/graph_c2_grad.npz, graph_v_grad.npz   (about half a minute)
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import graph as og  # noqa: E402

GRAD_SAMPLE = 2048
LR, WD = 1e-4, 1e-5


def fixture_params():
    d = np.load(os.path.join(HERE, 'graph_small.npz'))
    params = og.init_params(seed=11)
    for k in d.files:
        if k.startswith('p/'):
            params[k[2:]] = torch.from_numpy(d[k])
    chk = sum(float(v.double().abs().sum()) for k, v in params.items() if k.endswith('weights'))
    assert abs(chk - float(d['weight_checksum'])) < 1e-6 * chk
    return params


def headline_inputs():
    r2 = np.random.RandomState(64256)            # the seed graph_c2.npz was made with (make_golden.py)
    N, W, L = 64, 256, 10
    x = r2.rand(N, W, 32).astype(np.float32)
    labels = r2.randint(1, 63, N * L).astype(np.int32)
    return x, labels, np.full(N, L, np.int32), np.full(N, W // 4 - 1, np.int32)


def ragged_inputs():
    r = np.random.RandomState(80320)
    N, WMAX = 64, 320
    widths = (r.randint(80 // 4, 320 // 4 + 1, N) * 4).astype(np.int64)       # gen.py:58 pads every image to a multiple of 4
    widths[0], widths[1] = WMAX, 80                                            # both extremes present
    x = r.rand(N, WMAX, 32).astype(np.float32)
    for n in range(N):
        x[n, widths[n]:] = 0.0                                                 # groupBatch right-pads with 0 (gen.py:62)
    sl = (widths // 4 - 1).astype(np.int32)
    ll = r.randint(4, 11, N).astype(np.int32)
    labels = r.randint(1, 63, int(ll.sum())).astype(np.int32)
    return x, labels, ll, sl


def sample_index(name, numel):
    r = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return np.sort(r.choice(numel, size=min(GRAD_SAMPLE, numel), replace=False)).astype(np.int64)


def one(tag, inputs, params):
    x, labels, ll, sl = inputs
    out = {}
    names = sorted(params)
    out['names'] = np.array(names)
    for otag, sim in (('bf16sim', True), ('fp32', False)):
        t0 = time.time()
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        total, ctc, _ = og.loss_fn(leaves, torch.from_numpy(x), labels, ll, sl.tolist(), WD, sim_bf16=sim)
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
        clipped, norm = og.clip_by_global_norm(grads, 10.0)
        new = og.adam_step({k: v.detach().clone() for k, v in params.items()}, clipped, {}, LR)
        out['loss_total_' + otag] = np.float64(float(total.detach()))
        out['loss_ctc_' + otag] = np.float64(float(ctc.detach()))
        out['clip_norm_' + otag] = np.float64(norm)
        out['grad_norm_' + otag] = np.array([float((grads[k] - WD * params[k] if og.REGULARISED(k) else grads[k]).double().norm())
                                             for k in names])
        for k in names:
            idx = torch.from_numpy(sample_index(k, grads[k].numel()))
            # the device adds wd * w inside the optimiser kernel, so its gradient buffer holds the CTC term alone: store that
            # (the clip norm above is the norm of the FULL gradient, which is what optim_prep reports)
            g = (grads[k] - WD * params[k] if og.REGULARISED(k) else grads[k]).reshape(-1)[idx]
            out['grad_%s/%s' % (otag, k)] = g.numpy().astype(np.float32)
            out['new_%s/%s' % (otag, k)] = new[k].reshape(-1)[idx].numpy().astype(np.float32)
        print('%s %s: loss %.5f ctc %.5f clip norm %.5f  (%.1f s)' % (tag, otag, float(total.detach()), float(ctc.detach()), norm, time.time() - t0))
    out['x_checksum'] = np.float64(np.abs(x.astype(np.float64)).sum())
    out['lr'], out['wd'] = np.float64(LR), np.float64(WD)
    np.savez_compressed(os.path.join(HERE, 'graph_%s_grad.npz' % tag), **out)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    params = fixture_params()
    one('c2', headline_inputs(), params)
    one('v', ragged_inputs(), params)


if __name__ == '__main__':
    main()
