"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the pinned oracle).
CPU leg: the oracle still reproduces them bit-for-bit-ish (freezes the checker).  GPU leg: the HIP path against the
same fixed vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decode as odec
from oracle import graph as og

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_graph_fixture():
    d = np.load(os.path.join(G, 'graph_small.npz'))
    params = og.init_params(seed=11)
    for k in d.files:
        if k.startswith('p/'):
            params[k[2:]] = torch.from_numpy(d[k])
    chk = sum(float(v.double().abs().sum()) for k, v in params.items() if k.endswith('weights'))
    assert abs(chk - float(d['weight_checksum'])) < 1e-6 * chk, 'seeded weights differ from the ones the fixture was made with'
    return d, params


def c2_inputs():
    """Inputs of the headline-configuration fixture (BASELINE configs[1]: N = 64, W = 256, T = 63, 10-character labels)."""
    d = np.load(os.path.join(G, 'graph_c2.npz'))
    r2 = np.random.RandomState(int(d['seed']))
    N, W, L = 64, 256, 10
    x = r2.rand(N, W, 32).astype(np.float32)
    labels = r2.randint(1, 63, N * L).astype(np.int32)
    assert abs(float(np.abs(x.astype(np.float64)).sum()) - float(d['x_checksum'])) < 1e-6 * float(d['x_checksum'])
    return d, x, labels, np.full(N, L, np.int32), np.full(N, W // 4 - 1, np.int32)


def test_oracle_reproduces_headline_fixture():
    d, x, labels, ll, sl = c2_inputs()
    _, params = load_graph_fixture()
    lg = og.forward(params, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
    # 5e-4: on the machine that made the fixture the difference is 0; another CPU / BLAS thread count sums fp32 in another order and flips single bf16
    # roundings of intermediate activations (measured 1.5e-4 on the GPU box's 256-core host) — an edit of the oracle's arithmetic shows as >= 1e-2
    assert np.abs(lg.numpy() - d['logits_bf16sim']).max() < 5e-4
    costs = og._CTC.apply(lg, labels, ll, sl).numpy()
    assert np.allclose(costs, d['costs'], rtol=2e-4)


def test_oracle_reproduces_ctc_fixture():
    d = np.load(os.path.join(G, 'ctc_small.npz'))
    costs, grads = octc.ctc_loss_c(d['acts'], d['flat_labels'], d['label_lengths'], d['input_lengths'])
    assert np.allclose(costs, d['costs'], rtol=1e-6, atol=1e-6) and np.abs(grads - d['grads']).max() < 1e-6
    assert d['costs'][3] == 0 and np.all(d['grads'][:, 3] == 0)                 # the infeasible sample
    assert np.array_equal(odec.dense(odec.greedy_decode(d['acts'], d['input_lengths'])), d['greedy'])
    assert np.array_equal(odec.dense(odec.reference_decode(d['acts'], d['input_lengths'])), d['beam'])


def test_oracle_reproduces_graph_fixture():
    d, params = load_graph_fixture()
    l32 = og.forward(params, torch.from_numpy(d['x']), d['seq_len'].tolist(), sim_bf16=False).numpy()
    assert np.abs(l32 - d['logits_fp32']).max() < 1e-4
    total, ctc, _ = og.loss_fn(params, torch.from_numpy(d['x']), d['labels'], d['label_lengths'], d['seq_len'].tolist(), 1e-5,
                               sim_bf16=True)
    assert abs(float(total) - float(d['loss_total'])) < 1e-3 * float(d['loss_total'])


@pytest.mark.gpu
def test_device_ctc_matches_fixture(dev):
    from lstm_ctc_ocr_amd import ops
    d = np.load(os.path.join(G, 'ctc_small.npz'))
    t = lambda a: torch.from_numpy(a).to(dev)
    costs, grads = ops.ctc_loss(t(d['acts']), t(d['flat_labels']), t(d['label_lengths']), t(d['input_lengths']), 7)
    assert np.allclose(costs.cpu().numpy(), d['costs'], rtol=1e-4, atol=1e-4)          # bar: 1e-3 relative
    assert np.abs(grads.cpu().numpy() - d['grads']).max() < 1e-4
    out, lens = ops.ctc_greedy_decode(t(d['acts']), t(d['input_lengths']))
    out, lens = out.cpu().numpy(), lens.cpu().numpy()
    for n in range(out.shape[0]):
        want = [v for v in d['greedy'][n] if v != 0]
        assert out[n, :lens[n]].tolist() == want                                       # bar: identical
    out, lens, _ = ops.ctc_beam_decode(t(d['acts']), t(d['input_lengths']), beam_width=100)
    out, lens = out.cpu().numpy(), lens.cpu().numpy()
    for n in range(out.shape[0]):
        got = [v for v in out[n, :lens[n]] if v != 0]
        assert got == [v for v in d['beam'][n] if v != 0]                              # the reference's decode + zero stripping


@pytest.mark.gpu
def test_device_graph_matches_fixture(dev):
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    d, params = load_graph_fixture()
    cfg.TRAIN.WEIGHT_DECAY = 1e-5
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=1, use_graphs=False)
    eng.load_arrays({k: v.numpy() for k, v in params.items()})
    sl = d['seq_len']
    logits = eng.forward(d['x'], sl).float().cpu().numpy()
    for n in range(len(sl)):
        assert np.abs(logits[:sl[n], n] - d['logits_bf16sim'][:sl[n], n]).max() < 5e-3
    print('max |device - fp32 oracle| logits:', np.abs(logits - d['logits_fp32']).max())
    eng.setup_optimizer('Adam', 1e-4)
    loss = eng.train_step(d['x'], d['labels'], d['label_lengths'], sl)
    assert abs(loss - float(d['loss_total'])) < 1e-3 * float(d['loss_total'])         # bar: CTC loss within 1e-3 relative


@pytest.mark.gpu
def test_device_matches_headline_fixture(dev):
    """The device path at the full size of BASELINE configs[1] against the committed oracle output: logits, per-sample CTC costs
    (bar 1e-3 relative) and both decoders on the device's own logits."""
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    d, x, labels, ll, sl = c2_inputs()
    _, params = load_graph_fixture()
    eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=1)
    eng.load_arrays({k: v.numpy() for k, v in params.items()})
    logits = eng.forward(x, sl).float().cpu().numpy()
    assert logits.shape == (63, 64, cfg.NCLASSES)
    assert np.abs(logits - d['logits_bf16sim']).max() < 5e-3
    sp = eng.plan(64, 256)
    eng._bind(sp, x, sl, labels, ll)
    eng._run(sp, 'fb')
    torch.cuda.synchronize()
    costs = sp.costs.cpu().numpy()
    assert np.abs(costs - d['costs']).max() < 1e-3 * np.abs(d['costs']).max()
    # kernel level: the device's decoders on the device's own logits, all 64 samples (random weights give near-uniform
    # posteriors, so END-TO-END string identity is asserted on trained weights instead: tests/test_trained_fixture.py)
    assert odec.dense(eng.decode(x, sl, method='greedy')).tolist() == odec.dense(odec.greedy_decode(logits, sl)).tolist()
    assert eng.decode(x, sl, method='beam') == odec.reference_decode(logits, sl, beam_width=100)


def _grad_fixture(tag):
    import sys
    sys.path.insert(0, G)
    import make_headline_grad_golden as mh
    d = np.load(os.path.join(G, 'graph_%s_grad.npz' % tag))
    x, labels, ll, sl = mh.headline_inputs() if tag == 'c2' else mh.ragged_inputs()
    assert abs(float(np.abs(x.astype(np.float64)).sum()) - float(d['x_checksum'])) < 1e-9 * float(d['x_checksum'])
    return mh, d, (x, labels, ll, sl)


def test_oracle_reproduces_headline_gradient_fixture():
    """graph_c2_grad.npz freezes the oracle's backward + clip + Adam at N = 64, W = 256 (bf16-simulating walk)."""
    mh, d, (x, labels, ll, sl) = _grad_fixture('c2')
    params = mh.fixture_params()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    total, ctc, _ = og.loss_fn(leaves, torch.from_numpy(x), labels, ll, sl.tolist(), float(d['wd']), sim_bf16=True)
    total.backward()
    assert abs(float(total.detach()) - float(d['loss_total_bf16sim'])) < 1e-5 * float(d['loss_total_bf16sim'])
    grads = {k: v.grad for k, v in leaves.items()}
    _, norm = og.clip_by_global_norm(grads, 10.0)
    assert abs(norm - float(d['clip_norm_bf16sim'])) < 1e-4 * norm
    for k in ('conv2/weights', 'conv4_2/weights', 'logits/fw/weights', 'logits/weights', 'conv4_1/conv4_1/gamma'):
        g = (grads[k] - float(d['wd']) * params[k] if og.REGULARISED(k) else grads[k]).reshape(-1)
        got = g[torch.from_numpy(mh.sample_index(k, g.numel()))].numpy().astype(np.float64)
        ref = d['grad_bf16sim/' + k].astype(np.float64)
        assert np.linalg.norm(got - ref) < 2e-3 * np.linalg.norm(ref), k      # thread-count dependent fp32 summation order + routing flips


# Per-tensor bars for |g_dev - g_oracle|_2 / |g_oracle|_2 AT THE BENCHMARKED SIZE (N = 64): 3x the larger of the two measurements of
# profiles/r04a_grad_parity.log (headline W = 256 / ragged W = 320 fixture; MI355X, round 4) — conv1..conv3_2 3.0e-3..3.6e-3, conv4_1 1.6e-3..2.3e-3,
# conv4_2 1.3e-3..1.8e-3 (beta 1.3e-3, gamma 5.7e-4), conv5 3.6e-4..6.3e-4, logits/{fw,bw} 1.9e-4..7.5e-4, logits/weights 2.4e-4,
# logits/biases 1.9e-5..9.3e-5.  (The N = 8 bars of tests/test_gpu_engine.py are ~10x looser: at N = 8 a single max-pool / ReLU routing flip
# weighs 8x more.  THIS table is the whole-graph net at the size that is benchmarked; the tight per-kernel net is tests/test_gpu_kernels.py.)
GRAD_L2_BAR_N64 = {'default': 1.1e-2,
                   'conv4_1/weights': 7e-3, 'conv4_1/conv4_1/beta': 7e-3, 'conv4_1/conv4_1/gamma': 7e-3,
                   'conv4_2/weights': 5.5e-3, 'conv4_2/conv4_2/beta': 4e-3, 'conv4_2/conv4_2/gamma': 1.8e-3,
                   'conv5/weights': 2e-3, 'conv5/biases': 2e-3, 'logits/weights': 7.5e-4, 'logits/biases': 3e-4,
                   'logits/fw/weights': 2e-3, 'logits/fw/biases': 2e-3, 'logits/bw/weights': 2e-3, 'logits/bw/biases': 2e-3}
GRAD_NORM_RATIO_N64 = (0.995, 1.005)       # measured 0.9997 .. 1.0008


def test_headline_gradient_bars_are_tighter_than_the_small_batch_bars():
    """The N = 64 table must never be looser than the N = 8 one it replaced (a kernel wrong by 1 % at the benchmarked size fails it)."""
    from test_gpu_engine import GRAD_L2_BAR
    for k in set(GRAD_L2_BAR) | set(GRAD_L2_BAR_N64):
        assert GRAD_L2_BAR_N64.get(k, GRAD_L2_BAR_N64['default']) <= GRAD_L2_BAR.get(k, GRAD_L2_BAR['default']), k
    assert GRAD_L2_BAR_N64['default'] <= 1.1e-2


def _device_gradient_parity(tag, N, W):
    """The benchmarked step's BACKWARD half at the size it is benchmarked (lib/lstm/train.py:79-83 over LSTM_train.py:22-38): per
    parameter tensor, the device gradient against the committed sample of the bf16-simulating oracle's (relative L2 <= GRAD_L2_BAR_N64
    above = 3x what was measured at this size, and <= 20 % to the fp32 oracle's), full-tensor norm ratio 0.995..1.005, clip norm within
    1 %; then ONE clip + Adam step from the same state against the oracle's post-step parameters."""
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    from test_gpu_engine import GRAD_L2_BAR_FP32
    mh, d, (x, labels, ll, sl) = _grad_fixture(tag)
    params = mh.fixture_params()
    lr, wd = float(d['lr']), float(d['wd'])
    old_wd = cfg.TRAIN.WEIGHT_DECAY
    cfg.TRAIN.WEIGHT_DECAY = wd
    try:
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=1)
        eng.load_arrays({k: v.numpy() for k, v in params.items()})
        sp = eng.plan(N, W)
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        ctc_dev = float(sp.costs.cpu().numpy().astype(np.float64).mean())
        for otag in ('bf16sim', 'fp32'):
            assert abs(ctc_dev - float(d['loss_ctc_' + otag])) < 1e-3 * float(d['loss_ctc_' + otag]), (otag, ctc_dev)
        names = d['names'].tolist()
        bad = []
        for i, name in enumerate(names):
            g = eng.grad(name).reshape(-1)
            idx = torch.from_numpy(mh.sample_index(name, g.numel())).to(g.device)
            got = g[idx].double().cpu().numpy()
            ref, ref32 = d['grad_bf16sim/' + name].astype(np.float64), d['grad_fp32/' + name].astype(np.float64)
            if name in ('conv4_1/biases', 'conv4_2/biases'):      # mathematically zero (BN removes the mean): rounding noise on both sides
                scale = np.abs(d['grad_bf16sim/' + name.replace('biases', 'weights')]).max()
                assert np.isfinite(got).all() and np.abs(got).max() < scale, name
                continue
            e_sim = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            e_f32 = float(np.linalg.norm(got - ref32) / np.linalg.norm(ref32))
            nr = float(g.double().norm().cpu()) / float(d['grad_norm_bf16sim'][i])
            print('grad %-28s L2-rel vs bf16sim %.3e  vs fp32 %.3e  norm ratio %.4f' % (name, e_sim, e_f32, nr))
            if not (e_sim < GRAD_L2_BAR_N64.get(name, GRAD_L2_BAR_N64['default']) and e_f32 < GRAD_L2_BAR_FP32
                    and GRAD_NORM_RATIO_N64[0] < nr < GRAD_NORM_RATIO_N64[1]):
                bad.append((name, e_sim, e_f32, nr))
        assert not bad, bad
        # ---- one optimisation step from the same state (captured graph: forward + CTC + backward + clip + Adam + re-pack)
        eng.setup_optimizer('Adam', lr)
        loss = eng.train_step(x, labels, ll, sl)
        assert abs(loss - float(d['loss_total_bf16sim'])) < 1e-3 * float(d['loss_total_bf16sim'])
        assert abs(eng.last_gnorm - float(d['clip_norm_bf16sim'])) < 1e-2 * float(d['clip_norm_bf16sim']), eng.last_gnorm
        after = eng.state_arrays()
        worst = 1.0
        for name in names:
            if name in ('conv4_1/biases', 'conv4_2/biases'):
                continue
            idx = mh.sample_index(name, after[name].size)
            a = after[name].reshape(-1)[idx].astype(np.float64)
            before = params[name].reshape(-1).numpy()[idx].astype(np.float64)
            ref = d['new_bf16sim/' + name].astype(np.float64)
            # first Adam step = lr * g / (|g| + 3e-7 / clip scale): +-lr wherever the gradient is not tiny, so an entry can differ by at most
            # 2 lr, and does where device and oracle disagree about the SIGN of a near-zero gradient entry
            assert np.abs(a - ref).max() <= 2.02 * lr, name
            assert np.abs(a - before).max() <= 1.01 * lr, name
            big = np.abs(d['grad_bf16sim/' + name]) > 0.05 * np.abs(d['grad_bf16sim/' + name]).max()
            agree = float((np.sign(a - before) == np.sign(ref - before))[big].mean())
            mean_err = float(np.abs(a - ref).mean() / lr)
            worst = min(worst, agree)
            print('adam %-28s update sign agreement on the large entries %.4f, mean |dev - oracle| %.3f lr' % (name, agree, mean_err))
            assert agree > 0.995 and mean_err < 0.15, (name, agree, mean_err)
    finally:
        cfg.TRAIN.WEIGHT_DECAY = old_wd


@pytest.mark.gpu
def test_device_matches_headline_fixture_gradients(dev):
    _device_gradient_parity('c2', 64, 256)


@pytest.mark.gpu
def test_device_matches_ragged_fixture_gradients(dev):
    """BASELINE configs[3]: one ragged batch, W_i in [80, 320] padded to 320 (conv_k3w tiles, masked CTC, per-sample LSTM lengths)."""
    _device_gradient_parity('v', 64, 320)


def test_deep_fixture_parameters_are_reproducible():
    """tests/golden/deep_c4.npz was made from host_parameters(RESNET_train, seed): the same draw must come out today."""
    sys_path = os.path.join(G)
    import sys
    sys.path.insert(0, sys_path)
    import make_deep_golden as mdg
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.layout import host_parameters
    old = (cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID)
    try:
        d = np.load(os.path.join(G, 'deep_c4.npz'))
        params = host_parameters(mdg.build(), mdg.SEED)
        chk = sum(float(v.double().abs().sum()) for v in params.values())
        assert abs(chk - float(d['param_checksum'])) < 1e-9 * chk
        x, labels, ll, sl = mdg.inputs()
        assert abs(float(np.abs(x.astype(np.float64)).sum()) - float(d['x_checksum'])) < 1e-9 * float(d['x_checksum'])
        assert d['logits_fp32'].shape == (63, 32, 96) and sum(v.numel() for v in params.values()) == 32925664
    finally:
        cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = old


@pytest.mark.gpu
def test_device_matches_deep_fixture(dev):
    """BASELINE configs[4] at FULL size (ResNet-34-style [3,4,6,3] + 2 x BiLSTM(512 units per direction: NUM_HID = 1024), 96 classes,
    batch 32, W = 256, ragged lengths) against the committed oracle outputs: logits, per-sample CTC costs (bar 1e-3 relative, against
    the pure-fp32 oracle too), both decoders on the device's own logits, and the GRADIENT of the mean cost for every parameter tensor
    (norm + a seeded sample of 2048 entries per tensor from the bf16-simulating oracle's autograd)."""
    import sys
    sys.path.insert(0, G)
    import make_deep_golden as mdg
    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    old = (cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID)
    try:
        d = np.load(os.path.join(G, 'deep_c4.npz'))
        eng = Engine(mdg.build(), device='cuda:0', seed=mdg.SEED)
        chk = sum(float(torch.from_numpy(v).double().abs().sum()) for v in eng.state_arrays().values())
        assert abs(chk - float(d['param_checksum'])) < 1e-6 * chk
        x, labels, ll, sl = mdg.inputs()
        logits = eng.forward(x, sl).float().cpu().numpy()
        assert logits.shape == (63, 32, 96)
        e_sim = max(np.abs(logits[:sl[n], n] - d['logits_bf16sim'][:sl[n], n]).max() for n in range(32))
        e_f32 = max(np.abs(logits[:sl[n], n] - d['logits_fp32'][:sl[n], n]).max() for n in range(32))
        scale = np.abs(d['logits_fp32']).max()
        sp = eng.plan(32, 256)
        eng._bind(sp, x, sl, labels, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        costs = sp.costs.cpu().numpy().astype(np.float64)
        r_sim = np.abs(costs - d['costs_bf16sim']).max() / np.abs(d['costs_bf16sim']).max()
        r_f32 = np.abs(costs - d['costs_fp32']).max() / np.abs(d['costs_fp32']).max()
        print('deep fixture: logits |dev - bf16sim| %.2e, |dev - fp32| %.2e (max |logit| %.3f); costs rel vs bf16sim %.2e, vs fp32 %.2e'
              % (e_sim, e_f32, scale, r_sim, r_f32))
        assert e_sim < 2e-2 * max(scale, 1.0) and e_f32 < 5e-2 * max(scale, 1.0)
        assert r_sim < 1e-3 and r_f32 < 1e-3
        # ---- gradients (the 'fb' run above left them in eng.grads), per parameter tensor on the golden sample: relative L2 distance to
        #      the bf16-simulating oracle's gradient and the ratio of the full norms.  The yardstick is the distance between the two
        #      ORACLES (bf16-simulating vs pure fp32 forward, stored in the golden): 1 % near the loss, 0.6-0.9 from res3_0 down to
        #      conv1 — the early layers' gradient of a randomly initialised 34-layer batch-norm ResNet is chaotic under bf16-level
        #      forward differences (make_deep_golden.py).  Bar: the device is no farther from the bf16-simulating oracle than
        #      max(15 %, the fp32 oracle's distance), and its norm is within 15 %; a structural error (missed residual contribution,
        #      wrong mask / halo / K-split hand-off) shows as O(1) in the layers near the loss, where the bar is 15 %.
        bad, table = [], []
        for i, name in enumerate(d['grad_names'].tolist()):
            g = eng.grad(name).reshape(-1)
            idx = torch.from_numpy(mdg.sample_index(name, g.numel())).to(g.device)
            got = g[idx].double().cpu().numpy()
            ref, ref32 = d['grad_sample/' + name].astype(np.float64), d['grad32_sample/' + name].astype(np.float64)
            e_ref = float(np.linalg.norm(ref32 - ref) / max(np.linalg.norm(ref), 1e-30))
            nr = float(g.double().norm().cpu()) / max(float(d['grad_norm'][i]), 1e-30)
            if e_ref > 0.95:          # the two oracles are uncorrelated here: a mathematically zero gradient (bias in front of a batch
                if not nr < 4.0:      # norm), what is stored is summation noise — only its size is comparable
                    bad.append((name, 'noise-only tensor', nr))
                continue
            e2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            table.append((name, e2, e_ref, nr))
            if not (e2 < max(0.15, 1.05 * e_ref) and 0.85 < nr < 1.15):
                bad.append((name, e2, e_ref, nr))
        for name in ('logits/weights', 'logits/fw/weights', 'logits/stack0/fw/weights', 'conv5/weights', 'res4_2_b/weights', 'res4_0_a/weights',
                     'res3_5_b/weights', 'res3_0_a/weights', 'res2_3_b/weights', 'res2_0_a/weights', 'res1_2_b/weights', 'res1_0_a/weights',
                     'conv1/weights'):
            row = [t for t in table if t[0] == name]
            if row:
                print('grad %-26s device-vs-oracle L2 %.3f   (fp32 oracle vs oracle %.3f)   norm ratio %.3f' % row[0])
        print('deep fixture gradients: %d tensors compared' % len(table))
        assert not bad, bad[:8]
        assert eng.decode(x, sl, method='greedy') == odec.greedy_decode(logits, sl)
        assert eng.decode(x, sl, method='beam')[:4] == odec.reference_decode(logits[:, :4], sl[:4], beam_width=100)   # (pure-Python search: 4 samples)
    finally:
        cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID = old
