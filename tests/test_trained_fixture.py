"""End-to-end parity on TRAINED weights and RENDERED captcha batches (tests/golden/make_trained_fixture.py).

Anchor: the fp32 CPU oracle (no bf16 rounding anywhere) on the committed weights + batches -> tests/golden/trained_expect.npz.
  CPU leg : the oracle still reproduces the committed outputs (freezes the checker), the fixture is self-consistent.
  GPU leg : the MI355X path (bf16 storage, fp32 accumulation) on the same weights + batches:
            * greedy strings (blank 0) identical for ALL samples of every batch,
            * the reference's decode (beam 100, blank C-1, merge_repeated, zeros stripped) identical for ALL samples,
            * greedy-vs-beam disagreement count reported and equal to the oracle's,
            * loss: |device - fp32 oracle| <= max(1e-3 * loss, 5e-5).  A trained network's mean CTC cost is 2e-4 .. 2e-3; the
              bf16 activation storage leaves an ABSOLUTE error of ~1e-5 on it (0.3 .. 1.2 % of such a tiny loss), the same size
              as the gap between the fp32 oracle and the bf16-simulating oracle — so the relative bar of the north star is
              asserted where the loss is not itself at the noise floor: on the same rendered batches with untrained weights
              (test_random_init_loss_within_1e3_of_fp32_oracle) and at kernel level (same logits in -> cost within 1e-4,
              test_gpu_kernels.py::test_ctc_*).
"""
import os
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
sys.path.insert(0, G)
import make_trained_fixture as fx  # noqa: E402

from oracle import ctc as octc  # noqa: E402
from oracle import decode as odec  # noqa: E402
from oracle import graph as og  # noqa: E402

strip = lambda rows: [[int(v) for v in r if v != 0] for r in rows]


def truth_of(d, name):
    out, pos = [], 0
    for n in d[name + '/label_len']:
        out.append(d[name + '/labels'][pos:pos + n].tolist())
        pos += n
    return out


# ------------------------------------------------------------------------------------------------ CPU leg
def test_fixture_is_self_consistent():
    d, e = np.load(fx.BATCHES), np.load(fx.EXPECT)
    w = fx.load_weights()
    assert sum(v.size for v in w.values()) == 7158592                      # parameters of VGG-7 + BiLSTM(256)
    for v in w.values():                                                   # bf16-representable by construction
        assert np.array_equal(fx.from_bf16_bits(fx.to_bf16_bits(v)).reshape(v.shape), v)
    assert d['C1/x_u8'].shape == (8, 88, 32) and d['C2/x_u8'].shape == (64, 256, 32) and d['V0/x_u8'].shape == (64, 320, 32)
    assert len({d[n + '/x_u8'].shape[1] for n in ('V0', 'V1', 'V2')}) == 3          # three distinct padded widths
    k = int(np.argmin(d['V0/seq_len']))
    assert d['V0/label_len'][k] > d['V0/seq_len'][k] and e['V0/costs'][k] == 0      # the infeasible sample (warp-ctc: cost 0)
    for name in fx.NAMES:
        acc = np.mean([g == t for g, t in zip(strip(e[name + '/greedy']), truth_of(d, name))])
        dis = sum(g != b for g, b in zip(strip(e[name + '/greedy']), strip(e[name + '/beam'])))
        print('%s: oracle accuracy %.3f, greedy != beam on %d samples, mean cost %.6f' % (name, acc, dis, e[name + '/costs'].mean()))
        assert acc >= 0.98 and dis == 0          # trained: every frame is decisive, best path == beam search top-1


@pytest.mark.parametrize('name', ['C1', 'C2'])
def test_oracle_reproduces_trained_fixture(name):
    d, e = np.load(fx.BATCHES), np.load(fx.EXPECT)
    params = {k: torch.from_numpy(v.copy()) for k, v in fx.load_weights().items()}
    x, labels, ll, sl = fx.load_batch(d, name)
    with torch.no_grad():
        lg = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=False)
        costs = og._CTC.apply(lg, labels, ll, sl).numpy()
    assert np.abs(lg.numpy() - e[name + '/logits']).max() < 2e-3             # fp32 summation order of the host BLAS only
    assert np.allclose(costs, e[name + '/costs'], rtol=1e-3, atol=1e-7)
    assert strip(odec.dense(odec.greedy_decode(lg.numpy(), sl))) == strip(e[name + '/greedy'])
    if name == 'C1':                                                         # pure-Python beam search: 8 samples only
        assert odec.reference_decode(lg.numpy(), sl, beam_width=100) == strip(e[name + '/beam'])


# ------------------------------------------------------------------------------------------------ GPU leg
@pytest.fixture(scope='module')
def trained_engine(dev):
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    cfg.TRAIN.WEIGHT_DECAY = 1e-5
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=1, max_label_len=31)
    eng.load_arrays(fx.load_weights())
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(fx.NAMES))
def test_device_matches_fp32_oracle_on_trained_weights(trained_engine, name):
    eng = trained_engine
    d, e = np.load(fx.BATCHES), np.load(fx.EXPECT)
    x, labels, ll, sl = fx.load_batch(d, name)
    N = x.shape[0]
    logits = eng.forward(x, sl).float().cpu().numpy()
    err32 = max(np.abs(logits[:sl[n], n] - e[name + '/logits'][:sl[n], n]).max() for n in range(N))
    errsim = max(np.abs(logits[:sl[n], n] - e[name + '/logits_bf16sim'][:sl[n], n]).max() for n in range(N))
    scale = np.abs(e[name + '/logits']).max()
    greedy = eng.decode(x, sl, method='greedy')
    beam = eng.decode(x, sl, method='beam')
    want_g, want_b = strip(e[name + '/greedy']), strip(e[name + '/beam'])
    same_g = sum(a == b for a, b in zip(greedy, want_g))
    same_b = sum(a == b for a, b in zip(beam, want_b))
    dis = sum(a != b for a, b in zip(greedy, beam))
    sp = eng.plan(*x.shape[:2])
    eng._bind(sp, x, sl, labels, ll)
    eng._run(sp, 'fb')
    torch.cuda.synchronize()
    costs = sp.costs.cpu().numpy().astype(np.float64)
    ref, sim = e[name + '/costs'], e[name + '/costs_bf16sim']
    dl, rl = abs(costs.mean() - ref.mean()), abs(costs.mean() - ref.mean()) / ref.mean()
    print('%s: strings greedy %d/%d beam %d/%d identical, greedy != beam on %d; logits |dev - fp32| %.3f, |dev - bf16sim| %.3f '
          '(max |logit| %.1f); mean cost dev %.6f fp32 %.6f bf16sim %.6f -> abs %.2e rel %.2e (fp32 vs bf16sim rel %.2e)'
          % (name, same_g, N, same_b, N, dis, err32, errsim, scale, costs.mean(), ref.mean(), sim.mean(), dl, rl,
             abs(sim.mean() - ref.mean()) / ref.mean()))
    assert same_g == N and same_b == N                       # bar: identical strings, every sample, both decoders
    assert dis == sum(a != b for a, b in zip(want_g, want_b))
    assert err32 < 0.02 * scale and errsim < 0.01 * scale
    assert dl <= max(1e-3 * ref.mean(), 5e-5)
    assert np.abs(costs - ref).max() <= max(0.1 * np.abs(ref).max(), 5e-4)
    if name == 'V0':                                         # infeasible label: cost 0 and an all-zero gradient row
        k = int(np.argmin(sl))
        assert costs[k] == 0.0
        assert float(eng.ops[-1].dy(sp).view(N, -1)[k].float().abs().max()) == 0.0


@pytest.mark.gpu
def test_random_init_loss_within_1e3_of_fp32_oracle(dev):
    """North star: 'CTC loss within 1e-3 relative on the same captcha batch' — rendered C2 batch (N = 64, W = 256, 10
    characters), untrained weights (loss ~ 40: far from the noise floor), device vs the fp32 oracle, no bf16 simulation."""
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    cfg.TRAIN.WEIGHT_DECAY = 1e-5
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
    d = np.load(fx.BATCHES)
    for name in ('C1', 'C2'):
        x, labels, ll, sl = fx.load_batch(d, name)
        params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
        with torch.no_grad():
            lg = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=False)
            ref = og._CTC.apply(lg, labels, ll, sl).numpy().astype(np.float64)
        sp = eng.plan(*x.shape[:2])
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        costs = sp.costs.cpu().numpy().astype(np.float64)
        rel = abs(costs.mean() - ref.mean()) / ref.mean()
        print('%s random init: loss dev %.6f fp32 oracle %.6f rel %.2e; per-sample max rel %.2e'
              % (name, costs.mean(), ref.mean(), rel, (np.abs(costs - ref) / ref).max()))
        assert rel < 1e-3
        assert (np.abs(costs - ref) / ref).max() < 1e-3


@pytest.mark.gpu
def test_device_beam_search_tensorflow_known_answer(dev):
    """TF's ctc_decoder_ops_test.py::testCTCDecoderBeamSearch (see tests/test_oracle_ctc.py): beam_width 2 -> top path [1, 0]."""
    from lstm_ctc_ocr_amd import ops
    prob = np.array([[0.30999, 0.309938, 0.0679938, 0.0673362, 0.0708352, 0.173908],
                     [0.215136, 0.439699, 0.0370931, 0.0393967, 0.0381581, 0.230517],
                     [0.199959, 0.489485, 0.0233221, 0.0251417, 0.0233289, 0.238763],
                     [0.279611, 0.452966, 0.0204795, 0.0209126, 0.0194803, 0.20655],
                     [0.51286, 0.288951, 0.0243026, 0.0220788, 0.0219297, 0.129878],
                     [0.155251, 0.164444, 0.173517, 0.176138, 0.169979, 0.160671]], np.float32)
    # C must be a multiple of 8 for no kernel here: the decoder takes any C; blank = C-1 = 5
    acts = torch.from_numpy((np.log(prob) + 2.0)[:, None, :].astype(np.float32)).to(dev)
    il = torch.tensor([5], dtype=torch.int32, device=dev)
    for width, want in ((2, [1, 0]), (3, [0, 1, 0]), (100, [0, 1, 0])):
        out, lens, _ = ops.ctc_beam_decode(acts, il, beam_width=width, merge_repeated=False)
        assert out[0, :int(lens[0])].tolist() == want, (width, out[0].tolist())
