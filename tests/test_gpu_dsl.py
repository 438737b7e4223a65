"""-m gpu: the layers of the reference DSL beyond the shipped graphs — un-biased and strided `conv`, stand-alone
`batch_normalization` (training and inference mode), real `dropout`, `avg_pool`, `concat`, `softmax` — lowered by the engine and
checked against the plan-walking oracle (forward, loss, every gradient), plus kernel-level checks of the new entry points."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lstm_ctc_ocr_amd import ops
from lstm_ctc_ocr_amd.config import cfg
from lstm_ctc_ocr_amd.engine import Engine
from lstm_ctc_ocr_amd.network import Network
from oracle import graph as og
from oracle import plan_exec

BF = torch.bfloat16


class WideDslNet(Network):
    def __init__(self):
        self.inputs = []
        self.data = self.placeholder('data', 'float32', [None, None, 32])
        self.labels = self.placeholder('labels', 'int32', [None])
        self.time_step_len = self.placeholder('time_step_len', 'int32', [None])
        self.labels_len = self.placeholder('labels_len', 'int32', [None])
        self.keep_prob = self.placeholder('keep_prob', 'float32', [])
        self.layers = {'data': self.data, 'labels': self.labels, 'time_step_len': self.time_step_len, 'labels_len': self.labels_len}
        self.trainable = True
        self.setup()

    def setup(self):
        (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
             .conv(3, 3, 64, 1, 2, name='s2', biased=False, relu=True)                 # un-biased, stride 2 along the feature axis
             .batch_normalization(name='bnA', relu=False, is_training=True)
             .dropout(self.keep_prob, name='drop'))
        self.feed('drop').conv(3, 3, 64, 1, 1, name='ca', relu=False)
        (self.feed('ca', 'drop').concat(3, name='cat')
             .batch_normalization(name='bnB', relu=True, is_training=False)             # inference mode: moving statistics
             .avg_pool(1, 2, 1, 2, name='ap1', padding='VALID').avg_pool(1, 2, 1, 2, name='ap2', padding='VALID')
             .conv_single(2, 2, 128, 1, 1, padding='VALID', name='c5', relu=False)
             .reshape_squeeze_layer(d=128, name='rs'))
        self.feed('rs', 'time_step_len').bi_lstm(64, 1, name='logits')


def l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_wide_dsl_network_matches_oracle(dev):
    old = cfg.TRAIN.WEIGHT_DECAY
    cfg.TRAIN.WEIGHT_DECAY = 0.0
    try:
        net = WideDslNet()
        eng = Engine(net, device='cuda:0', seed=7)
        g = torch.Generator().manual_seed(1)
        arrays = {}
        for name, spec in eng.specs.items():                       # non-trivial BN affine / moving statistics, small biases
            v = eng.param(name).cpu()
            if name.endswith(('gamma', 'moving_variance')):
                v = 1.0 + 0.2 * (torch.rand(spec.shape, generator=g) - 0.5)
            elif name.endswith(('beta', 'moving_mean', 'biases')):
                v = 0.1 * (torch.rand(spec.shape, generator=g) - 0.5)
            arrays[name] = v.numpy()
        eng.load_arrays(arrays)
        params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
        rng = np.random.RandomState(3)
        N, W = 16, 64
        x = rng.rand(N, W, 32).astype(np.float32)
        sl = np.full(N, W // 2 - 1, np.int32); sl[3] = 20
        ll = np.full(N, 3, np.int32)
        lab = rng.randint(1, 63, N * 3).astype(np.int32)
        # inference: dropout keeps everything
        logits = eng.forward(x, sl).float().cpu()
        ref = plan_exec.forward(net, params, torch.from_numpy(x), sl.tolist(), sim_bf16=True, keep_prob=1.0)
        assert tuple(logits.shape) == (W // 2 - 1, N, cfg.NCLASSES)
        for n in range(N):
            assert float((logits[:sl[n], n] - ref[:sl[n], n]).abs().max()) < 1e-2
        # training step body: keep_prob 0.5 (train.py:126), mask salted with the optimiser step count (0 here)
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        lg = plan_exec.forward(net, leaves, torch.from_numpy(x), sl.tolist(), sim_bf16=True, keep_prob=0.5, step=0)
        costs = og._CTC.apply(lg, lab, ll, np.asarray(sl, np.int32))
        costs.mean().backward()
        sp = eng.plan(N, W)
        eng._bind(sp, x, sl, lab, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        dev_cost = float(sp.costs.cpu().numpy().mean())
        assert abs(dev_cost - float(costs.mean())) / float(costs.mean()) < 2e-3, (dev_cost, float(costs.mean()))
        bad = []
        for name in eng.specs:
            if name.endswith(('moving_mean', 'moving_variance')):
                assert float(eng.grad(name).abs().max()) == 0.0                       # not trainable: never touched
                continue
            r = leaves[name].grad
            if r is None or float(r.abs().max()) < 1e-9:
                continue
            e = l2(eng.grad(name).cpu(), r)
            print('grad %-28s L2-rel %.3e' % (name, e))
            if not e < 1e-2:                      # measured 2e-5 .. 4.5e-3 on MI355X (round 2)
                bad.append((name, e))
        assert not bad, bad
        eng.setup_optimizer('Adam', 1e-3)
        l0 = eng.train_step(x, lab, ll, sl)
        l1 = [eng.train_step(x, lab, ll, sl) for _ in range(20)][-1]
        assert np.isfinite(l1) and l1 < l0
    finally:
        cfg.TRAIN.WEIGHT_DECAY = old


class FcNet(WideDslNet):
    """conv stack -> reshape_squeeze -> fc(256, relu) -> dropout slot -> fc(128, no relu) -> bi_lstm: `Network.fc` (network.py:415-447) on the
    row tensor, twice, one of them feeding the other."""

    def setup(self):
        (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
             .conv_single(3, 3, 64, 1, 1, name='c2').max_pool(2, 2, 2, 2, padding='VALID', name='p2')
             .max_pool(1, 2, 1, 2, padding='VALID', name='p3').max_pool(1, 2, 1, 2, padding='VALID', name='p4')
             .conv_single(2, 2, 128, 1, 1, padding='VALID', name='c5', relu=False)
             .reshape_squeeze_layer(d=128, name='rs')
             .fc(256, name='fc1')
             .fc(128, name='fc2', relu=False))
        self.feed('fc2', 'time_step_len').bi_lstm(64, 1, name='logits')


def test_fc_layer_matches_oracle(dev):
    old = cfg.TRAIN.WEIGHT_DECAY
    cfg.TRAIN.WEIGHT_DECAY = 0.0
    try:
        net = FcNet()
        assert net.param_specs['fc1/weights'].shape == (128, 256) and net.param_specs['fc1/weights'].regularized
        assert net.param_specs['fc2/biases'].shape == (128,) and not net.param_specs['fc2/biases'].regularized
        with pytest.raises(NotImplementedError):                    # a 4-D map of dynamic width has no static flattened size
            net.feed('p2').fc(10, name='bad')
        eng = Engine(net, device='cuda:0', seed=11)
        g = torch.Generator().manual_seed(2)
        arrays = {}
        for name, spec in eng.specs.items():                        # the reference's stddev-0.01 init leaves fc gradients at the bf16 noise floor: widen it
            v = eng.param(name).cpu()
            if name.startswith('fc') and name.endswith('weights'):
                v = 0.12 * torch.randn(spec.shape, generator=g)
            elif name.endswith('biases'):
                v = 0.1 * (torch.rand(spec.shape, generator=g) - 0.5)
            arrays[name] = v.numpy()
        eng.load_arrays(arrays)
        params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
        rng = np.random.RandomState(5)
        N, W = 16, 64
        x = rng.rand(N, W, 32).astype(np.float32)
        sl = np.full(N, W // 4 - 1, np.int32); sl[2] = 9; sl[7] = 4
        ll = np.full(N, 3, np.int32)
        lab = rng.randint(1, 63, N * 3).astype(np.int32)
        logits = eng.forward(x, sl).float().cpu()
        ref = plan_exec.forward(net, params, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
        assert tuple(logits.shape) == (W // 4 - 1, N, cfg.NCLASSES)
        for n in range(N):
            assert float((logits[:sl[n], n] - ref[:sl[n], n]).abs().max()) < 1e-2
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        lg = plan_exec.forward(net, leaves, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
        costs = og._CTC.apply(lg, lab, ll, np.asarray(sl, np.int32))
        costs.mean().backward()
        sp = eng.plan(N, W)
        eng._bind(sp, x, sl, lab, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        dev_cost = float(sp.costs.cpu().numpy().mean())
        assert abs(dev_cost - float(costs.mean())) / float(costs.mean()) < 2e-3, (dev_cost, float(costs.mean()))
        bad = []
        for name in eng.specs:
            r = leaves[name].grad
            if r is None or float(r.abs().max()) < 1e-9:
                continue
            e = l2(eng.grad(name).cpu(), r)
            print('grad %-28s L2-rel %.3e' % (name, e))
            if not e < 1e-2:
                bad.append((name, e))
        assert not bad, bad
        assert float(eng.grad('fc1/weights').abs().max()) > 0 and float(eng.grad('fc2/biases').abs().max()) > 0
        eng.setup_optimizer('Adam', 1e-3)
        l0 = eng.train_step(x, lab, ll, sl)
        l1 = [eng.train_step(x, lab, ll, sl) for _ in range(20)][-1]
        assert np.isfinite(l1) and l1 < l0
    finally:
        cfg.TRAIN.WEIGHT_DECAY = old


class UniLstmNet(WideDslNet):
    """conv stack -> `lstm` (network.py:130-152): two stacked unidirectional LSTMCell(64) + FC, ragged lengths."""

    def setup(self):
        (self.feed('data').conv_single(3, 3, 64, 1, 1, name='c1', c_i=1).max_pool(2, 2, 2, 2, padding='VALID', name='p1')
             .conv_single(3, 3, 64, 1, 1, name='c2').max_pool(2, 2, 2, 2, padding='VALID', name='p2')
             .max_pool(1, 2, 1, 2, padding='VALID', name='p3').max_pool(1, 2, 1, 2, padding='VALID', name='p4')
             .conv_single(2, 2, 128, 1, 1, padding='VALID', name='c5', relu=False)
             .reshape_squeeze_layer(d=128, name='rs'))
        self.feed('rs', 'time_step_len').lstm(64, 2, name='logits')


def test_dropout_with_a_constant_keep_prob_in_the_graph(dev):
    """`.dropout(0.8, ...)` — a Python number instead of the keep_prob input slot — is a constant of the graph (tf.nn.dropout
    applies it in every run); the engine used to ignore it.  Inference forward == the oracle with the same constant, and != keep-all."""
    class ConstDrop(WideDslNet):
        def setup(self):
            self.keep_prob = 0.8
            WideDslNet.setup(self)
    old = cfg.TRAIN.WEIGHT_DECAY
    cfg.TRAIN.WEIGHT_DECAY = 0.0
    try:
        net = ConstDrop()
        eng = Engine(net, device='cuda:0', seed=7)
        drop = [op for op in eng.ops if op.name == 'drop'][0]
        assert drop.const_keep_prob == 0.8
        params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
        rng = np.random.RandomState(3)
        N, W = 8, 64
        x = rng.rand(N, W, 32).astype(np.float32)
        sl = np.full(N, W // 2 - 1, np.int32)
        logits = eng.forward(x, sl).float().cpu()
        ref = plan_exec.forward(net, params, torch.from_numpy(x), sl.tolist(), sim_bf16=True, keep_prob=1.0)       # the layer's own 0.8 wins
        keep_all = plan_exec.forward(WideDslNet(), params, torch.from_numpy(x), sl.tolist(), sim_bf16=True, keep_prob=1.0)
        assert float((logits - ref).abs().max()) < 1e-2
        assert float((logits - keep_all).abs().max()) > 5 * float((logits - ref).abs().max())
    finally:
        cfg.TRAIN.WEIGHT_DECAY = old


def test_unidirectional_stacked_lstm_matches_oracle(dev):
    old = cfg.TRAIN.WEIGHT_DECAY
    cfg.TRAIN.WEIGHT_DECAY = 0.0
    try:
        net = UniLstmNet()
        names = set(net.param_specs)
        for li in range(2):                                   # TF-1.0 variable names of MultiRNNCell / dynamic_rnn
            assert 'logits/rnn/multi_rnn_cell/cell_%d/lstm_cell/weights' % li in names
        assert tuple(net.param_specs['logits/rnn/multi_rnn_cell/cell_1/lstm_cell/weights'].shape) == (64 + 64, 256)
        assert tuple(net.param_specs['logits/weights'].shape) == (64, cfg.NCLASSES)
        eng = Engine(net, device='cuda:0', seed=11)
        g = torch.Generator().manual_seed(2)
        arrays = {n: (0.1 * (torch.rand(sp.shape, generator=g) - 0.5)).numpy() for n, sp in eng.specs.items() if n.endswith('biases')}
        eng.load_arrays(arrays)
        params = {k: torch.from_numpy(v) for k, v in eng.state_arrays().items()}
        rng = np.random.RandomState(5)
        N, W = 16, 96
        T = W // 4 - 1
        x = rng.rand(N, W, 32).astype(np.float32)
        sl = rng.randint(5, T + 1, N).astype(np.int32); sl[0] = T; sl[1] = 1
        ll = np.minimum(rng.randint(1, 5, N), sl).astype(np.int32)
        lab = rng.randint(1, 63, int(ll.sum())).astype(np.int32)
        logits = eng.forward(x, sl).float().cpu()
        ref = plan_exec.forward(net, params, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
        assert tuple(logits.shape) == (T, N, cfg.NCLASSES)
        for n in range(N):
            assert float((logits[:sl[n], n] - ref[:sl[n], n]).abs().max()) < 1e-2
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        lg = plan_exec.forward(net, leaves, torch.from_numpy(x), sl.tolist(), sim_bf16=True)
        costs = og._CTC.apply(lg, lab, ll, np.asarray(sl, np.int32))
        costs.mean().backward()
        sp = eng.plan(N, W)
        eng._bind(sp, x, sl, lab, ll)
        eng._run(sp, 'fb')
        torch.cuda.synchronize()
        dev_cost = float(sp.costs.cpu().numpy().mean())
        assert abs(dev_cost - float(costs.mean())) / float(costs.mean()) < 2e-3, (dev_cost, float(costs.mean()))
        bad = []
        for name in eng.specs:
            r = leaves[name].grad
            if r is None or float(r.abs().max()) < 1e-9:
                continue
            e = l2(eng.grad(name).cpu(), r)
            print('grad %-56s L2-rel %.3e' % (name, e))
            if not e < 5e-3:                      # measured 3e-5 .. 6e-4 on MI355X (round 2)
                bad.append((name, e))
        assert not bad, bad
        eng.setup_optimizer('Adam', 1e-3)
        l0 = eng.train_step(x, lab, ll, sl)
        l1 = [eng.train_step(x, lab, ll, sl) for _ in range(20)][-1]
        assert np.isfinite(l1) and l1 < l0
    finally:
        cfg.TRAIN.WEIGHT_DECAY = old


def test_dsl_kernels(dev):
    g = torch.Generator().manual_seed(2)
    r = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    # softmax
    a = r(37, 5, 96) * 4
    out = torch.empty_like(a, device=dev)
    ops.softmax(a.to(dev), out)
    assert float((out.cpu() - torch.softmax(a, -1)).abs().max()) < 1e-6
    # element-wise: add, relu, relu-mask, fused add+relu (== relu of the bf16-rounded sum), masked accumulate
    ea, eb = r(4, 33, 8, 16).to(BF), r(4, 33, 8, 16).to(BF)
    eo = torch.empty_like(ea, device=dev)
    ops.eltwise(0, ea.to(dev), eb.to(dev), eo)
    assert torch.equal(eo.cpu(), (ea.float() + eb.float()).to(BF))
    ops.eltwise(3, ea.to(dev), eb.to(dev), eo)
    assert torch.equal(eo.cpu(), torch.relu((ea.float() + eb.float()).to(BF)))
    ops.eltwise(2, ea.to(dev), eb.to(dev), eo)
    assert torch.equal(eo.cpu(), torch.where(eb.float() > 0, ea, torch.zeros_like(ea)))
    acc = r(4, 33, 8, 16).to(BF)
    eacc = acc.to(dev).clone()
    ops.eltwise(4, ea.to(dev), eb.to(dev), eacc)
    assert torch.equal(eacc.cpu(), (acc.float() + torch.where(eb.float() > 0, ea.float(), torch.zeros(()))).to(BF))
    # strided pick and its transpose are adjoint; TF SAME geometry for k = 3, s = 2: offset 1 on an even axis, 0 on an odd one
    Nb, W, H, C = 3, 9, 8, 16
    x = r(Nb, W, H, C).to(BF)
    Wo, Ho = 5, 4
    y = torch.empty(Nb, Wo, Ho, C, dtype=BF, device=dev)
    ops.subsample(x.to(dev), y, Nb, W, H, C, Wo, Ho, 2, 2, 0, 1)
    assert torch.equal(y.cpu(), x[:, 0::2, 1::2])
    dy = r(Nb, Wo, Ho, C).to(BF)
    dx = torch.empty(Nb, W, H, C, dtype=BF, device=dev)
    ops.subsample(dy.to(dev), dx, Nb, W, H, C, Wo, Ho, 2, 2, 0, 1, backward=True)
    ref = torch.zeros(Nb, W, H, C, dtype=BF); ref[:, 0::2, 1::2] = dy
    assert torch.equal(dx.cpu(), ref)
    # average pool
    x = r(2, 6, 8, 16).to(BF)
    y = torch.empty(2, 3, 4, 16, dtype=BF, device=dev)
    ops.avgpool(x.to(dev), y, 2, 6, 8, 16, 2, 2)
    want = torch.nn.functional.avg_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert float((y.float().cpu() - want).abs().max()) < 8e-3
    # dropout: the mask is a pure function of (seed, step, index); kept values are scaled by 1 / keep_prob
    n = 1 << 16
    x = torch.ones(n, dtype=BF)
    step = torch.tensor([5.0], dtype=torch.float64, device=dev)
    y1 = torch.empty(n, dtype=BF, device=dev); y2 = torch.empty(n, dtype=BF, device=dev)
    ops.dropout(x.to(dev), y1, 1234, step, 0.5); ops.dropout(x.to(dev), y2, 1234, step, 0.5)
    assert torch.equal(y1, y2) and set(y1.float().cpu().unique().tolist()) == {0.0, 2.0}
    assert abs(float((y1 > 0).float().mean()) - 0.5) < 0.02
    m = plan_exec.dropout_mask((n,), 'drop', 5, 0.5)              # the oracle's mask for a layer named 'drop'
    import zlib
    ops.dropout(x.to(dev), y1, zlib.crc32(b'drop') ^ 0x5bd1e995, step, 0.5)
    assert torch.equal((y1 > 0).cpu(), m > 0)
    step.fill_(6.0)
    ops.dropout(x.to(dev), y2, zlib.crc32(b'drop') ^ 0x5bd1e995, step, 0.5)
    assert not torch.equal(y1, y2)                                   # another step, another mask
    # inference-mode batch norm, forward and backward
    M, C = 1000, 64
    x = r(M, C).to(BF); dyv = r(M, C).to(BF)
    gamma, beta, mean, var = 1 + 0.3 * r(C), 0.2 * r(C), 0.1 * r(C), 1 + 0.5 * torch.rand(C, generator=g)
    xr = x.float().clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    yr = torch.relu((xr - mean) * torch.rsqrt(var + 1e-3) * gr + br)
    yr.backward(dyv.float())
    y = torch.empty(M, C, dtype=BF, device=dev); dx = torch.empty(M, C, dtype=BF, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    ops.bn_infer_fwd(x.to(dev), gamma.to(dev), beta.to(dev), mean.to(dev), var.to(dev), 1e-3, True, y)
    assert float((y.float().cpu() - yr.detach()).abs().max()) < 2e-2
    ops.bn_infer_bwd(x.to(dev), y, dyv.to(dev), gamma.to(dev), mean.to(dev), var.to(dev), dg, db, 1e-3, True, dx)
    # the device masks with ITS (bf16) output; elements whose pre-activation rounds across zero differ — compare in L2
    assert l2(dx.float().cpu(), xr.grad) < 2e-2 and l2(dg.cpu(), gr.grad) < 2e-2 and l2(db.cpu(), br.grad) < 2e-2


def test_bind_batch_and_step_report(dev):
    """The two host-boundary kernels of a training iteration: ocr_bind_batch (device batch -> the engine's fixed input buffers,
    uint8 / 255 exactly as groupBatch's astype(float32) / 255., gen.py:59-65) and ocr_step_report (what train.py:130,139 fetches)."""
    g = torch.Generator().manual_seed(5)
    N, W, Hh, L = 8, 96, 32, 40
    pix = torch.randint(0, 256, (N, W, Hh), dtype=torch.uint8, generator=g)
    sl = torch.randint(1, 24, (N,), dtype=torch.int32, generator=g)
    lab = torch.randint(1, 60, (L,), dtype=torch.int32, generator=g)
    ll = torch.randint(1, 8, (N,), dtype=torch.int32, generator=g)
    for src in (pix, pix.float() / 3.0):
        x = torch.full((N, W, Hh), -1.0, device=dev)
        d_sl = torch.full((N,), -1, dtype=torch.int32, device=dev)
        d_lab = torch.full((N * 10,), -1, dtype=torch.int32, device=dev)
        d_ll = torch.full((N,), -1, dtype=torch.int32, device=dev)
        ops.bind_batch(src.to(dev), x, sl.to(dev), d_sl, lab.to(dev), d_lab, ll.to(dev), d_ll)
        want = (src.numpy().astype(np.float32) / 255.) if src.dtype == torch.uint8 else src.numpy()
        assert np.array_equal(x.cpu().numpy(), want)                      # bit-exact (IEEE division on the device)
        assert torch.equal(d_sl.cpu(), sl) and torch.equal(d_ll.cpu(), ll)
        assert torch.equal(d_lab.cpu()[:L], lab) and bool((d_lab.cpu()[L:] == -1).all())
    # inference form: no labels
    x = torch.empty((N, W, Hh), device=dev); d_sl = torch.zeros((N,), dtype=torch.int32, device=dev)
    ops.bind_batch(pix.to(dev), x, sl.to(dev), d_sl)
    assert torch.equal(d_sl.cpu(), sl)

    costs = torch.rand(64, generator=g) * 30
    sc = torch.arange(ops.optim_scalar_count(), dtype=torch.float64) * 1.5       # the WHOLE optimiser block: the report reads [1], [7] and (round 6) [72]
    sc[72] = 0.0
    words = [torch.zeros(65, dtype=torch.int32, device=dev) for _ in range(3)]
    words[1][-1] = 1                      # the persistent LSTM kernels' time-out mark
    words[2][-1] = -1                     # a block the caller prepared (all ones) that nothing touched: not an error
    addrs = torch.tensor([w[-1:].data_ptr() for w in words], dtype=torch.int64, device=dev)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    ops.step_report(costs.to(dev), sc.to(dev), addrs, out)
    o = out.cpu().numpy()
    assert abs(o[0] - float(costs.double().mean())) < 1e-12 and o[1] == 1.5 and o[2] == 10.5 and o[3] == 2.0
    sc[72] = 1.0                          # the guarded optimiser launch dropped this step: bit 40 of the report's word (the host decides per step)
    ops.step_report(costs.to(dev), sc.to(dev), addrs, out)
    assert out.cpu().numpy()[3] == 2.0 + 2.0 ** 40
    ops.step_report(costs.to(dev), None, None, out)
    o = out.cpu().numpy()
    assert o[1] == 0.0 and o[2] == 0.0 and o[3] == 0.0
