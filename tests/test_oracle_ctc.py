"""Pins the CTC oracle (oracle/ctc_ref.c, oracle/ctc.py) — the checker every device-side CTC claim rests on.
The reference holds no vectors for this path; the pins are warp-ctc's published known-answer test, exact
path enumeration, and torch's independent CPU implementation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctc as octc
from oracle import decode as odec


def test_warpctc_known_answer_vector():
    # warp-ctc tests/test_cpu.cpp::small_test == tensorflow_binding/tests/test_warpctc_op.py::_testBasic
    acts = np.array([[[0.1, 0.6, 0.1, 0.1, 0.1]], [[0.1, 0.1, 0.6, 0.1, 0.1]]], np.float32)
    for fn in (octc.ctc_loss_c, octc.ctc_loss_numpy):
        costs, grads = fn(acts, [1, 2], [2], [2])
        assert abs(float(costs[0]) - 2.46286) < 1e-5
        exp = np.full((2, 5), 0.177031); exp[0, 1] = exp[1, 2] = -0.708125
        assert np.abs(np.asarray(grads).reshape(2, 5) - exp).max() < 1e-5
    p = np.exp(acts[:, 0, :]) / np.exp(acts[:, 0, :]).sum(-1, keepdims=True)
    assert abs(-np.log(p[0, 1] * p[1, 2]) - 2.4628585) < 1e-6        # analytic: the only path is (1, 2)


@pytest.mark.parametrize("T,C,label", [(4, 3, [1, 2]), (5, 3, [1, 1]), (5, 4, [3]), (3, 3, []), (4, 4, [2, 2, 1][:2])])
def test_against_path_enumeration(T, C, label):
    rng = np.random.RandomState(T * 10 + C)
    acts = rng.randn(T, 1, C).astype(np.float32) * 1.5
    cost_bf, table = octc.ctc_brute_force(acts[:, 0, :], label)
    assert abs(sum(table.values()) - 1.0) < 1e-9
    costs, grads = octc.ctc_loss_c(acts, label, [len(label)], [T])
    assert abs(costs[0] - cost_bf) < 1e-5
    # gradient by central differences of the enumerated loss
    eps = 1e-3
    for t in range(T):
        for k in range(C):
            ap, am = acts.copy().astype(np.float64), acts.copy().astype(np.float64)
            ap[t, 0, k] += eps; am[t, 0, k] -= eps
            num = (octc.ctc_brute_force(ap[:, 0, :], label)[0] - octc.ctc_brute_force(am[:, 0, :], label)[0]) / (2 * eps)
            assert abs(num - grads[t, 0, k]) < 2e-4


def test_against_torch_ctc_loss_c2_shape():
    rng = np.random.RandomState(7)
    T, N, C = 63, 64, 64
    acts = rng.randn(T, N, C).astype(np.float32) * 2
    ll = rng.randint(4, 11, N)
    il = rng.randint(40, T + 1, N)
    labels = [rng.randint(1, C, l) for l in ll]
    labels[3][1] = labels[3][0]; labels[5][:] = labels[5][0]          # repeats
    flat = np.concatenate(labels).astype(np.int32)
    costs, grads = octc.ctc_loss_c(acts, flat, ll, il)
    a = torch.from_numpy(acts).requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(a, -1), torch.from_numpy(flat).long(), torch.from_numpy(il), torch.from_numpy(ll),
                     blank=0, reduction='none')
    ref.sum().backward()
    assert np.allclose(costs, ref.detach().numpy(), rtol=1e-5, atol=1e-4)
    assert np.abs(grads - a.grad.numpy()).max() < 3e-4      # torch runs the recursion in fp32, the oracle in fp64
    c2, g2 = octc.ctc_loss_numpy(acts[:, :4], np.concatenate(labels[:4]), ll[:4], il[:4])
    assert np.allclose(c2, costs[:4], rtol=1e-6) and np.abs(g2 - grads[:, :4]).max() < 1e-6


def test_infeasible_and_padding_semantics():
    rng = np.random.RandomState(1)
    acts = rng.randn(6, 3, 5).astype(np.float32)
    # sample 0: L + repeats = 4 + 3 > T = 6 -> cost 0, grad 0 (warp-ctc); sample 1 uses only 3 frames
    costs, grads = octc.ctc_loss_c(acts, [2, 2, 2, 2, 1, 3, 4], [4, 2, 1], [6, 3, 6])
    assert costs[0] == 0 and np.all(grads[:, 0] == 0)
    assert costs[1] > 0 and np.all(grads[3:, 1] == 0) and np.abs(grads[:3, 1]).sum() > 0
    assert np.allclose(grads[:, 2].sum(-1), 0, atol=1e-6)               # softmax - posterior sums to 0 per frame


def test_decoders_on_hand_cases_and_enumeration():
    # the two-blanks quirk — loss blank is 0, TF's decoder blank is C-1 and zeros are stripped later
    am = [5, 5, 0, 5, 0, 0, 7]
    logits = np.full((7, 1, 9), -9.0, np.float32)       # C = 9: class 8 is the TF decoder's blank
    for t, a in enumerate(am): logits[t, 0, a] = 9.0
    assert odec.greedy_decode(logits, [7]) == [[5, 5, 7]]
    assert odec.reference_decode(logits, [7]) == [[5, 5, 7]]
    logits[:, 0, :] = -9.0; logits[:, 0, 0] = 9.0
    assert odec.greedy_decode(logits, [7]) == [[]]
    # class C-1 frames act as the TF decoder's blank: the search itself keeps both fives of [5, blank, 5], and
    # merge_repeated=True (the reference's setting) then collapses them in the OUTPUT — TF's documented quirk; real
    # doubles survive in the reference only because they are separated by the symbol 0
    logits = np.full((3, 1, 8), -9.0, np.float32)
    logits[0, 0, 5] = logits[1, 0, 7] = logits[2, 0, 5] = 9.0
    assert odec.beam_search_tf(logits, [3], merge_repeated=False)[0] == [[5, 5]]
    assert odec.beam_search_tf(logits, [3], merge_repeated=True)[0] == [[5]]
    # exhaustive beam == most probable labelling from path enumeration (blank = C-1, no post-merge)
    rng = np.random.RandomState(3)
    for trial in range(5):
        T, C = 5, 4
        x = rng.randn(T, 1, C) * 2
        _, table = octc.ctc_brute_force(x[:, 0, :], [], blank=C - 1)
        best = max(table.items(), key=lambda kv: kv[1])
        seqs, scores = odec.beam_search_tf(x, [T], beam_width=10000, merge_repeated=False)
        assert tuple(seqs[0]) == best[0]
        assert abs(np.exp(scores[0]) - best[1]) < 1e-9


def test_tensorflow_ctc_loss_known_answers():
    """TensorFlow's own CTC loss vectors (tensorflow/python/kernel_tests/ctc_loss_op_test.py::testBasic, 1.0 line): two sequences of 5
    steps over depth 6 (blank = class 5, TF's convention), targets [0, 1, 2, 1, 0] and [0, 1, 1, 0], expected
    -log p = 3.34211 and 5.42262.  Both oracle implementations (numpy and the fp64 C one) must give them; the second target has a
    repeated label, i.e. a mandatory blank between the two 1s.  With the beam-search vector below and the LSTM cell vector in
    test_oracle_graph.py these are the TensorFlow-held numbers the oracle is pinned on."""
    m0 = np.array([[0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
                   [0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436],
                   [0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688],
                   [0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533],
                   [0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]])
    m1 = np.array([[0.30176, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
                   [0.24082, 0.397533, 0.0557226, 0.0546814, 0.0557528, 0.19549],
                   [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, 0.202456],
                   [0.280884, 0.429522, 0.0326593, 0.0339046, 0.0326856, 0.190345],
                   [0.423286, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]])
    act = np.stack([np.log(m0), np.log(m1)], 1).astype(np.float32)          # [T = 5, N = 2, C = 6]
    labels = np.array([0, 1, 2, 1, 0, 0, 1, 1, 0], np.int32)
    ll, il = np.array([5, 4], np.int32), np.array([5, 5], np.int32)
    for fn in (octc.ctc_loss_numpy, octc.ctc_loss_c):
        costs, grads = fn(act, labels, ll, il, blank=5)
        assert abs(costs[0] - 3.34211) < 2e-5 and abs(costs[1] - 5.42262) < 2e-5, (fn.__name__, costs)
        # d cost / d activation of a softmax input = softmax - posterior: every frame's gradient sums to zero
        assert np.abs(np.asarray(grads).sum(-1)).max() < 1e-5
        # TensorFlow lists the gradients too (gradient_log_prob_0 / _1 of the same test).  Both targets leave exactly ONE alignment
        # (5 frames for 5 labels; 5 frames for 4 labels with a mandatory blank between the repeated 1s), so the posterior is one-hot
        # and the listed rows are prob - onehot: e.g. TF's -0.366234 = 0.633766 - 1, -0.797544 = 0.202456 - 1 (blank column, frame 2)
        for n, (m, path) in enumerate(((m0, [0, 1, 2, 1, 0]), (m1, [0, 1, 5, 1, 0]))):
            want = m.copy()
            want[np.arange(5), path] -= 1.0
            assert np.abs(np.asarray(grads)[:, n, :] - want).max() < 2e-6, (fn.__name__, n)
        assert abs(np.asarray(grads)[0, 0, 0] - (-0.366234)) < 2e-6 and abs(np.asarray(grads)[2, 1, 5] - (-0.797544)) < 2e-6


def test_tensorflow_greedy_decoder_known_answer():
    """ctc_decoder_ops_test.py::testCTCGreedyDecoder (depth 4, blank = class 3, sequence lengths 4 and 5 of 6 steps): TF expects
    [0, 1] and [1, 1, 0] — repeats merged, blanks dropped, a repeat separated by a blank kept, frames past seq_len ignored."""
    m0 = [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.4, 0.6], [0.0, 0.0, 0.4, 0.6], [0.0, 0.9, 0.1, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]]
    m1 = [[0.1, 0.9, 0.0, 0.0], [0.0, 0.9, 0.1, 0.0], [0.0, 0.0, 0.1, 0.9], [0.0, 0.9, 0.1, 0.1], [0.9, 0.1, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]]
    x = np.stack([np.array(m0), np.array(m1)], 1)
    assert odec.greedy_decode(x, [4, 5], blank=3) == [[0, 1], [1, 1, 0]]


def test_tensorflow_beam_search_known_answer():
    """The only external vector that exists for tf.nn.ctc_beam_search_decoder (network.py:656): TensorFlow's own
    ctc_decoder_ops_test.py::testCTCDecoderBeamSearch — depth 6 (blank = class 5), 5 time steps, beam_width = 2,
    top_paths = 2, merge_repeated = False.  TF expects decoded[0] = [1, 0] and decoded[1] = [0, 1, 0]: with only two beams
    the search loses the truly most probable labelling ([0, 1, 0], which enumeration and any beam >= 3 return) to [1, 0] —
    the oracle must reproduce that pruning behaviour, not just the arg-max.  (The log-probabilities TF 1.0 lists for this
    vector, 0.584855 / 0.389139, are not probabilities of these labellings — exp(-a) + exp(-b) > 1 — so only the
    label sequences are pinned.)"""
    prob = np.array([[0.30999, 0.309938, 0.0679938, 0.0673362, 0.0708352, 0.173908],
                     [0.215136, 0.439699, 0.0370931, 0.0393967, 0.0381581, 0.230517],
                     [0.199959, 0.489485, 0.0233221, 0.0251417, 0.0233289, 0.238763],
                     [0.279611, 0.452966, 0.0204795, 0.0209126, 0.0194803, 0.20655],
                     [0.51286, 0.288951, 0.0243026, 0.0220788, 0.0219297, 0.129878],
                     [0.155251, 0.164444, 0.173517, 0.176138, 0.169979, 0.160671]], np.float64)   # row 5: beyond seq_len
    logits = (np.log(prob) + 2.0)[:, None, :]                  # "arbitrary offset - this is fine"
    seqs, scores = odec.beam_search_tf(logits, [5], beam_width=2, merge_repeated=False, top_paths=2)
    assert seqs[0] == [[1, 0], [0, 1, 0]]
    assert scores[0][0] > scores[0][1]
    # cross-check of the two labellings' true probabilities by path enumeration (blank = 5)
    _, table = octc.ctc_brute_force(logits[:5, 0, :], [], blank=5)
    assert max(table.items(), key=lambda kv: kv[1])[0] == (0, 1, 0)
    assert odec.beam_search_tf(logits, [5], beam_width=100, merge_repeated=False)[0] == [[0, 1, 0]]
    assert abs(np.exp(odec.beam_search_tf(logits, [5], beam_width=10000, merge_repeated=False)[1][0]) - table[(0, 1, 0)]) < 1e-9
