"""Generates the committed golden vectors from the pinned oracle.  The reference cannot run here (TensorFlow 1.0.1 /
warp-ctc are not installable) and holds no fixtures of its own, so these are oracle outputs on seeded inputs — they
freeze the oracle (CPU test) and give the -m gpu tests fixed vectors that do not depend on the RNG of the GPU box.
    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import ctc as octc  # noqa: E402
from oracle import decode as odec  # noqa: E402
from oracle import graph as og  # noqa: E402


def main():
    rng = np.random.RandomState(20260925)
    # (1) CTC batch with repeats, variable lengths and one infeasible sample
    T, N, C = 12, 5, 8
    acts = (rng.randn(T, N, C) * 2).astype(np.float32)
    labels = [[1, 2, 3], [4, 4, 4], [7], [2, 2, 2, 2, 2, 2, 2], [5, 6]]
    flat = np.array([v for l in labels for v in l], np.int32)
    ll = np.array([len(l) for l in labels], np.int32)
    il = np.array([12, 9, 1, 12, 7], np.int32)
    costs, grads = octc.ctc_loss_c(acts, flat, ll, il)
    greedy = odec.dense(odec.greedy_decode(acts, il))
    beam = odec.dense(odec.reference_decode(acts, il, beam_width=100))
    np.savez(os.path.join(HERE, 'ctc_small.npz'), acts=acts, flat_labels=flat, label_lengths=ll, input_lengths=il,
             costs=costs, grads=grads, greedy=greedy, beam=beam)
    # (2) whole-graph logits for seeded parameters, fp32 and bf16-rounded arithmetic
    params = og.init_params(seed=11)     # big weight tensors are regenerated from this seed by the tests (checksummed below)
    for k in params:                                       # non-trivial biases / BN affine so every term is exercised
        if k.endswith('biases') or k.endswith('beta'):
            params[k] = torch.from_numpy((rng.randn(*params[k].shape) * 0.05).astype(np.float32))
        if k.endswith('gamma'):
            params[k] = torch.from_numpy((1 + rng.randn(*params[k].shape) * 0.1).astype(np.float32))
    x = rng.rand(3, 40, 32).astype(np.float32)
    sl = [9, 6, 9]
    x[1, 28:] = 0
    l32 = og.forward(params, torch.from_numpy(x), sl, sim_bf16=False).numpy()
    l16 = og.forward(params, torch.from_numpy(x), sl, sim_bf16=True).numpy()
    lab = np.array([3, 9, 1, 22, 40], np.int32); lablen = np.array([2, 1, 2], np.int32)
    total, ctc, _ = og.loss_fn(params, torch.from_numpy(x), lab, lablen, sl, 1e-5, sim_bf16=True)
    np.savez(os.path.join(HERE, 'graph_small.npz'), x=x, seq_len=np.array(sl, np.int32), logits_fp32=l32, logits_bf16sim=l16,
             labels=lab, label_lengths=lablen, loss_total=np.float64(total.item()), loss_ctc=np.float64(ctc.item()),
             weight_checksum=np.float64(sum(float(v.double().abs().sum()) for k, v in params.items() if k.endswith('weights'))),
             **{'p/' + k: v.numpy() for k, v in params.items() if not k.endswith('weights')})
    # (3) BASELINE configs[1] at full size: N = 64, W = 256 (T = 63), 10-character labels, seeded parameters (same biases / BN
    #     affine as above), bf16-rounded arithmetic.  Inputs and labels are regenerated from the seed by the tests.
    r2 = np.random.RandomState(64256)
    N, W, L = 64, 256, 10
    x2 = r2.rand(N, W, 32).astype(np.float32)
    lab2 = r2.randint(1, 63, N * L).astype(np.int32)
    ll2 = np.full(N, L, np.int32)
    sl2 = [W // 4 - 1] * N
    lg = og.forward(params, torch.from_numpy(x2), sl2, sim_bf16=True)
    costs2 = og._CTC.apply(lg, lab2, ll2, np.asarray(sl2, np.int32)).numpy()
    np.savez_compressed(os.path.join(HERE, 'graph_c2.npz'), logits_bf16sim=lg.numpy().astype(np.float32), costs=costs2,
                        greedy=odec.dense(odec.greedy_decode(lg.numpy(), np.asarray(sl2, np.int32))),
                        x_checksum=np.float64(np.abs(x2.astype(np.float64)).sum()), seed=np.int64(64256))
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
