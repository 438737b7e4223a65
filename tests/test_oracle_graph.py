"""Pins the layer-level oracle (oracle/graph.py) on independent implementations: a re-packed torch.nn.LSTM for the
TF-1.0 LSTMCell restatement, plain numpy loops for conv / pool / batch-norm, and TF's documented sequence-length rules."""
import math

import numpy as np
import torch

from oracle import graph as og


def test_lstm_cell_matches_repacked_torch_lstm():
    torch.manual_seed(0)
    N, T, D, U = 3, 7, 12, 8
    x = torch.randn(N, T, D)
    W = torch.randn(D + U, 4 * U) * 0.3       # TF layout, gate order i, j, f, o
    b = torch.randn(4 * U) * 0.1
    out = og.lstm_direction(x, [T] * N, W, b, reverse=False, sim=False, forget_bias=1.0)
    ref = torch.nn.LSTM(D, U, batch_first=True)           # torch gate order i, f, g, o; two biases; no forget bias
    i, j, f, o = W.split(U, dim=1)
    bi, bj, bf_, bo = b.split(U)
    Wt = torch.cat([i, f, j, o], 1)
    with torch.no_grad():
        ref.weight_ih_l0.copy_(Wt[:D].t()); ref.weight_hh_l0.copy_(Wt[D:].t())
        ref.bias_ih_l0.copy_(torch.cat([bi, bf_ + 1.0, bj, bo])); ref.bias_hh_l0.zero_()
    exp, _ = ref(x)
    assert float((out - exp).abs().max()) < 1e-5


def test_lstm_cell_equations_on_tensorflows_own_test_vector():
    """TensorFlow's rnn_cell test `testBasicLSTMCell` (tensorflow/contrib/rnn/python/kernel_tests/core_rnn_cell_test.py in the 1.0
    line): two stacked BasicLSTMCell(2) under constant_initializer(0.5) (biases 0), input [[1, 1]], every state entry 0.1 ->
    output [[0.24024698, 0.24024698]] and state [c1 c1 h1 h1 c2 c2 h2 h2] = [0.68967271 x2, 0.44848421 x2, 0.39897051 x2, 0.24024698 x2].
    A vector TensorFlow itself holds for the cell the reference builds (network.py:104-107: LSTMCell without peepholes or projection
    computes the same equations): it pins forget_bias = 1 added to f, c' = sigmoid(f + 1) c + sigmoid(i) tanh(j), h' = sigmoid(o) tanh(c')
    and the [x, h] operand order of this oracle.  (All weights are equal, so it does not distinguish the gate ORDER — that one is
    pinned on the re-packed torch.nn.LSTM above.)"""
    U = 2
    W = torch.full((2 + U, 4 * U), 0.5)
    b = torch.zeros(4 * U)
    s0 = (torch.full((1, U), 0.1), torch.full((1, U), 0.1))
    h1, (c1, hs1) = og.lstm_direction(torch.tensor([[[1.0, 1.0]]]), [1], W, b, False, False, state0=s0, return_state=True)
    h2, (c2, hs2) = og.lstm_direction(h1, [1], W, b, False, False, state0=s0, return_state=True)
    got = torch.cat([c1, hs1, c2, hs2], 1)[0]
    want = torch.tensor([0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698])
    assert float((got - want).abs().max()) < 2e-7, got
    assert float((h2[0, 0] - torch.tensor([0.24024698, 0.24024698])).abs().max()) < 2e-7


def test_conv_pool_clip_on_tensorflows_own_test_vectors():
    """Vectors TensorFlow's unit tests hold for the ops the reference graph is built from (1.0 line):
    conv_ops_test.py testConv2D2x2Filter (input 1..18 as [1,2,3,3] NHWC, filter 1..36 as [2,2,3,3] HWIO, VALID — the shape class of
    conv5) and testConv2D1x1Filter; pooling_ops_test.py max-pool VALID 2x2 / stride 2 on 1..27 as [1,3,3,3]; clip_ops_test.py
    testClipByGlobalNormClipped (train.py:79-83 clips by global norm).  They pin the oracle's tensor layout conventions (NHWC x HWIO,
    cross-correlation, channel contraction order) and the clip formula on TensorFlow's own numbers."""
    x = torch.arange(1, 19, dtype=torch.float32).reshape(1, 2, 3, 3)
    w22 = torch.arange(1, 37, dtype=torch.float32).reshape(2, 2, 3, 3)
    assert og.conv_single(x, w22, torch.zeros(3), 'VALID', False).flatten().tolist() == [2271.0, 2367.0, 2463.0, 2901.0, 3033.0, 3165.0]
    w11 = torch.arange(1, 10, dtype=torch.float32).reshape(1, 1, 3, 3)
    assert og.conv_single(x, w11, torch.zeros(3), 'VALID', False).flatten().tolist() == [
        30.0, 36.0, 42.0, 66.0, 81.0, 96.0, 102.0, 126.0, 150.0, 138.0, 171.0, 204.0, 174.0, 216.0, 258.0, 210.0, 261.0, 312.0]
    xp = torch.arange(1, 28, dtype=torch.float32).reshape(1, 3, 3, 3)
    assert og.max_pool(xp[:, :2, :2], 2, 2).flatten().tolist() == [13.0, 14.0, 15.0]
    clipped, norm = og.clip_by_global_norm({'x0': torch.tensor([[-2.0, 0.0, 0.0], [4.0, 0.0, 0.0]]), 'x1': torch.tensor([1.0, -2.0])}, clip=4.0)
    assert norm == 5.0
    assert torch.allclose(clipped['x0'], torch.tensor([[-1.6, 0.0, 0.0], [3.2, 0.0, 0.0]])) and torch.allclose(clipped['x1'], torch.tensor([0.8, -1.6]))


def test_optimizer_formulas_on_tensorflows_own_test_vectors():
    """momentum_test.py::testBasic and rmsprop_test.py::testWithoutMomentum of the 1.0 line (learning rate 2.0; RMSProp decay 0.9,
    epsilon 1.0): the accumulator / mean-square values TF lists after each step — 0.1 then 0.9 * 0.1 + 0.1; rms 0.901 then
    0.901 * 0.9 + 0.001 (the slot starts at ONE) — and the variables that follow from them."""
    v = {'v0': torch.tensor([1.0, 2.0], dtype=torch.float64), 'v1': torch.tensor([3.0, 4.0], dtype=torch.float64)}
    g = {'v0': torch.tensor([0.1, 0.1], dtype=torch.float64), 'v1': torch.tensor([0.01, 0.01], dtype=torch.float64)}
    st = {}
    p = og.momentum_step(dict(v), g, st, lr=2.0, momentum=0.9)
    assert torch.allclose(st['acc/v0'], torch.tensor([0.1, 0.1], dtype=torch.float64)) and torch.allclose(p['v0'], torch.tensor([1.0 - 0.1 * 2.0, 2.0 - 0.1 * 2.0], dtype=torch.float64))
    p = og.momentum_step(p, g, st, lr=2.0, momentum=0.9)
    a2 = 0.9 * 0.1 + 0.1
    assert torch.allclose(st['acc/v0'], torch.tensor([a2, a2], dtype=torch.float64))
    assert torch.allclose(p['v0'], torch.tensor([1.0 - 0.1 * 2.0 - a2 * 2.0, 2.0 - 0.1 * 2.0 - a2 * 2.0], dtype=torch.float64))
    assert torch.allclose(p['v1'], torch.tensor([2.98 - (0.9 * 0.01 + 0.01) * 2.0, 3.98 - (0.9 * 0.01 + 0.01) * 2.0], dtype=torch.float64))
    st = {}
    p = og.rmsprop_step(dict(v), g, st, lr=2.0, decay=0.9, eps=1.0)
    assert torch.allclose(st['rms/v0'], torch.tensor([0.901, 0.901], dtype=torch.float64)) and torch.allclose(st['rms/v1'], torch.tensor([0.90001, 0.90001], dtype=torch.float64))
    s1 = 0.1 * 2.0 / (0.901 + 1.0) ** 0.5
    assert torch.allclose(p['v0'], torch.tensor([1.0 - s1, 2.0 - s1], dtype=torch.float64))
    p = og.rmsprop_step(p, g, st, lr=2.0, decay=0.9, eps=1.0)
    r2 = 0.901 * 0.9 + 0.001
    assert torch.allclose(st['rms/v0'], torch.tensor([r2, r2], dtype=torch.float64))
    s2 = 0.1 * 2.0 / (r2 + 1.0) ** 0.5
    assert torch.allclose(p['v0'], torch.tensor([1.0 - s1 - s2, 2.0 - s1 - s2], dtype=torch.float64))


def test_bidirectional_sequence_length_semantics():
    torch.manual_seed(1)
    N, T, D, U = 2, 6, 5, 4
    x = torch.randn(N, T, D)
    W = torch.randn(D + U, 4 * U) * 0.3; b = torch.zeros(4 * U)
    lens = [6, 3]
    fw = og.lstm_direction(x, lens, W, b, False, False)
    bw = og.lstm_direction(x, lens, W, b, True, False)
    # outputs past the length are zero (dynamic_rnn), and the padded tail never influences valid frames
    assert float(fw[1, 3:].abs().max()) == 0 and float(bw[1, 3:].abs().max()) == 0
    x2 = x.clone(); x2[1, 3:] = 99.0
    assert torch.equal(og.lstm_direction(x2, lens, W, b, False, False)[1, :3], fw[1, :3])
    assert torch.equal(og.lstm_direction(x2, lens, W, b, True, False)[1, :3], bw[1, :3])
    # backward direction == forward direction on the sequence reversed WITHIN its length
    xr = x.clone(); xr[1, :3] = x[1, :3].flip(0); xr[0] = x[0].flip(0)
    fr = og.lstm_direction(xr, lens, W, b, False, False)
    assert float((fr[1, :3].flip(0) - bw[1, :3]).abs().max()) < 1e-6
    assert float((fr[0].flip(0) - bw[0]).abs().max()) < 1e-6


def test_conv_pool_bn_against_numpy_loops():
    rng = np.random.RandomState(0)
    N, W, H, Ci, Co = 2, 5, 4, 3, 2
    x = rng.randn(N, W, H, Ci).astype(np.float32); w = rng.randn(3, 3, Ci, Co).astype(np.float32); b = rng.randn(Co).astype(np.float32)
    got = og.conv_single(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 'SAME', False).numpy()
    exp = np.zeros((N, W, H, Co), np.float32)
    for n in range(N):
        for i in range(W):
            for j in range(H):
                for a in range(3):
                    for c in range(3):
                        ii, jj = i + a - 1, j + c - 1
                        if 0 <= ii < W and 0 <= jj < H:
                            exp[n, i, j] += x[n, ii, jj] @ w[a, c]
    assert np.abs(got - (exp + b)).max() < 1e-4
    # VALID 2x2 (conv5): output [W-1, H-1]
    w2 = rng.randn(2, 2, Ci, Co).astype(np.float32)
    got = og.conv_single(torch.from_numpy(x), torch.from_numpy(w2), torch.zeros(Co), 'VALID', False).numpy()
    assert got.shape == (N, W - 1, H - 1, Co)
    assert abs(got[0, 1, 2, 0] - sum(x[0, 1 + a, 2 + c] @ w2[a, c, :, 0] for a in range(2) for c in range(2))) < 1e-4
    # pooling: first argument pools the time axis W, second the feature axis H (network.py:343-350 with TF height = W)
    p = og.max_pool(torch.from_numpy(x[:, :4]), 1, 2).numpy()
    assert p.shape == (N, 4, 2, Ci) and np.allclose(p[0, 1, 1], np.maximum(x[0, 1, 2], x[0, 1, 3]))
    # batch norm: biased variance over (N, W, H), eps 1e-3
    y = og.batch_norm_train(torch.from_numpy(x), torch.ones(Ci) * 2, torch.ones(Ci) * 0.5).numpy()
    flat = x.reshape(-1, Ci)
    exp = (flat - flat.mean(0)) / np.sqrt(flat.var(0) + 1e-3) * 2 + 0.5
    assert np.abs(y.reshape(-1, Ci) - exp).max() < 1e-4


def test_full_graph_shapes_loss_and_optimizer_formulas():
    p = og.init_params()
    assert sum(v.numel() for v in p.values()) == 7158592          # SURVEY §8d parameter count
    x = torch.rand(2, 32, 32)
    logits = og.forward(p, x, [7, 5])
    assert tuple(logits.shape) == (7, 2, 64)                      # T = W/4 - 1, time-major
    # frames past a sample's length carry the FC bias only
    assert float((logits[5:, 1] - p['logits/biases']).abs().max()) < 1e-6
    total, ctc, _ = og.loss_fn(p, x, [1, 2, 3], [2, 1], [7, 5], 1e-5)
    reg = sum(0.5e-5 * float((v ** 2).sum()) for k, v in p.items() if og.REGULARISED(k))
    assert abs(float(total) - float(ctc) - reg) < 1e-5
    # clip_by_global_norm + Adam (TF form) on a toy problem
    g = {'a': torch.tensor([30.0, 40.0])}
    clipped, norm = og.clip_by_global_norm(g, 10.0)
    assert abs(norm - 50.0) < 1e-6 and torch.allclose(clipped['a'], torch.tensor([6.0, 8.0]))
    st = {}
    new = og.adam_step({'a': torch.zeros(2)}, clipped, st, lr=0.1)
    assert torch.allclose(new['a'], torch.tensor([-0.1, -0.1]), atol=1e-6)   # first Adam step = -lr * sign(g)


def test_batch_norm_train_against_torchs_own_batch_norm():
    """The oracle's restatement of tf.contrib.layers.batch_norm(is_training=True, epsilon=1e-3, biased variance over N, W, H — network.py:176-178)
    pinned on an INDEPENDENT implementation: torch.nn.functional.batch_norm(training=True, eps=1e-3) after the NHWC -> NCHW permute,
    forward and all three gradients; including the reference's quirk Q2: batch-norm stays in training mode at inference,
    so a batch of ONE image normalises with that image's own statistics, and zero-padded columns are part of the statistics."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for shape, pad_from in (((6, 9, 4, 16), None), ((1, 12, 4, 8), None), ((3, 10, 2, 8), 6)):
        z = torch.randn(shape, generator=g, dtype=torch.float64) * 1.7 + 0.3
        if pad_from is not None:
            z[:, pad_from:] = 0.25                       # what a padded column holds after conv + bias of an all-zero input: a constant
        gamma = torch.rand(shape[3], generator=g, dtype=torch.float64) + 0.5
        beta = torch.randn(shape[3], generator=g, dtype=torch.float64)
        dy = torch.randn(shape, generator=g, dtype=torch.float64)
        za, ga, ba = (t.clone().requires_grad_(True) for t in (z, gamma, beta))
        ya = og.batch_norm_train(za, ga, ba)
        ya.backward(dy)
        zb, gb, bb = (t.clone().requires_grad_(True) for t in (z, gamma, beta))
        yb = F.batch_norm(zb.permute(0, 3, 1, 2), None, None, gb, bb, training=True, momentum=0.0, eps=1e-3).permute(0, 2, 3, 1)
        yb.backward(dy)
        assert float((ya - yb).abs().max()) < 1e-12
        for a, b in ((za.grad, zb.grad), (ga.grad, gb.grad), (ba.grad, bb.grad)):
            assert float((a - b).abs().max()) < 1e-10 * max(1.0, float(b.abs().max()))
        # the statistics really include the padded columns: normalising the unpadded part alone gives something else
        if pad_from is not None:
            alone = og.batch_norm_train(z[:, :pad_from], gamma, beta)
            assert float((alone - ya.detach()[:, :pad_from]).abs().max()) > 1e-2
    # epsilon is 1e-3 (contrib default), not torch's 1e-5: a low-variance channel tells them apart
    z = torch.zeros(2, 3, 1, 1, dtype=torch.float64); z[0, 0, 0, 0] = 0.02
    y = og.batch_norm_train(z, torch.ones(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))
    var = float(z.var(unbiased=False))
    assert abs(float(y[0, 0, 0, 0]) - (0.02 - 0.02 / 6) / math.sqrt(var + 1e-3)) < 1e-12


def test_adam_on_a_hand_derived_three_step_trajectory():
    """TF-1.0 AdamOptimizer (train.py:74): m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);
    w -= lr_t m / (sqrt(v) + eps) — epsilon OUTSIDE the bias correction ("epsilon hat" of the paper's section 2), unlike torch.optim.Adam.
    Three steps of a scalar with constant gradient g = 2 and of one with gradients 1, -1, 3, worked out by hand in exact fractions:
    constant g: m_t = g (1 - b1^t), v_t = g^2 (1 - b2^t)  =>  step = lr_t m_t / (sqrt(v_t) + eps) = lr g (.) / (|g| + eps / sqrt(1 - b2^t))."""
    from fractions import Fraction as Fr
    b1, b2, lr, eps = Fr(9, 10), Fr(999, 1000), Fr(1, 100), 1e-8
    # (a) constant gradient: closed form
    w, st = {'a': torch.tensor([1.0], dtype=torch.float64)}, {}
    want = 1.0
    for t in (1, 2, 3):
        w = og.adam_step(w, {'a': torch.tensor([2.0], dtype=torch.float64)}, st, lr=float(lr))
        want -= float(lr) * 2.0 / (2.0 + eps / math.sqrt(1.0 - float(b2) ** t))
        assert abs(float(w['a']) - want) < 1e-13, t
    # (b) gradients 1, -1, 3: moments as exact fractions
    m = v = Fr(0)
    w, st, want = {'a': torch.tensor([0.0], dtype=torch.float64)}, {}, 0.0
    for t, g in enumerate((1, -1, 3), 1):
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        lr_t = float(lr) * math.sqrt(1.0 - float(b2 ** t)) / (1.0 - float(b1 ** t))
        want -= lr_t * float(m) / (math.sqrt(float(v)) + eps)
        w = og.adam_step(w, {'a': torch.tensor([float(g)], dtype=torch.float64)}, st, lr=float(lr))
        assert abs(float(w['a']) - want) < 1e-13, t
    assert m == Fr(1, 10) * 3 + Fr(9, 10) * (Fr(-1, 10) + Fr(9, 100)) and abs(want + 0.01 - 0.0 + 0.00329) < 5e-3       # sanity of the hand values
    # the unit test TensorFlow itself ships (adam_test.py::testBasic: numpy reference with the same lr_t form, epsilon outside) moves
    # var0 = [1, 2] with grads [0.1, 0.1], lr 0.001 by exactly lr * sign on the first step up to eps: 1 - 0.001 * (0.1 / (0.1 + eps'))
    w = og.adam_step({'v': torch.tensor([1.0, 2.0], dtype=torch.float64)}, {'v': torch.tensor([0.1, 0.1], dtype=torch.float64)}, {}, lr=0.001)
    assert torch.allclose(w['v'], torch.tensor([0.999, 1.999], dtype=torch.float64), atol=1e-9)
