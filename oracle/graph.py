"""ORACLE (test infrastructure): CPU restatement of the reference's training graph.

Follows, line by line:
  graph          lib/networks/LSTM_train.py:22-38  (conv1 .. conv5, reshape, bi_lstm)
  conv_single    lib/networks/network.py:160-191   (conv2d NHWC/HWIO stride 1, bias, optional BN, optional ReLU)
  max_pool       lib/networks/network.py:343-350
  bi_lstm        lib/networks/network.py:97-129    (2 x LSTMCell(num_hids//2), bidirectional_dynamic_rnn, FC, transpose)
  build_loss     lib/networks/network.py:647-664   (mean warp-ctc cost + sum wd*l2_loss(w))
  optimiser      lib/lstm/train.py:73-83           (clip_by_global_norm 10 -> Adam), TF-1.0 formulas
TF / warp-ctc op semantics are restated from their published definitions (they are not vendored in the reference):
batch_norm = contrib defaults (eps 1e-3, biased batch variance, always training mode — network.py:176-178);
LSTMCell gate order (i, j, f, o), forget_bias 1.0, zero initial state, outputs zero / state frozen past the
sequence length, backward direction reversed within the length.

Arithmetic: torch CPU fp32 kernels (conv2d / max_pool2d / matmul) — an implementation independent of the HIP
kernels.  `sim_bf16=True` additionally rounds to bfloat16 at exactly the points where the MI355X path stores
bf16 (activations, GEMM operands), so that the two paths differ only by fp32 summation order.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ctc as octc

BN_EPS = 1e-3
CONV_SPECS = [  # name, kh, kw, cin, cout, padding, bn, relu     (LSTM_train.py:24-34)
    ("conv1", 3, 3, 1, 64, "SAME", False, True),
    ("conv2", 3, 3, 64, 128, "SAME", False, True),
    ("conv3_1", 3, 3, 128, 256, "SAME", False, True),
    ("conv3_2", 3, 3, 256, 256, "SAME", False, True),
    ("conv4_1", 3, 3, 256, 512, "SAME", True, True),
    ("conv4_2", 3, 3, 512, 512, "SAME", True, True),
    ("conv5", 2, 2, 512, 512, "VALID", False, False),
]
POOL_AFTER = {"conv1": (2, 2), "conv2": (2, 2), "conv3_2": (1, 2), "conv4_2": (1, 2)}   # (k over W, k over H)


def q(x, sim):
    """bf16 rounding of a WEIGHT operand (forward only; the gradient passes straight through — the device keeps weight
    gradients in fp32)."""
    if not sim:
        return x
    return x + (x.detach().to(torch.bfloat16).to(torch.float32) - x.detach())


class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of an ACTIVATION as the device stores it: the value is rounded on the way forward and its gradient is
    rounded on the way back (the device keeps every activation gradient as bf16: DESIGN.md section 2)."""

    @staticmethod
    def forward(ctx, x, fwd):
        return x.to(torch.bfloat16).to(torch.float32) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32), None


SIM_GRAD_ROUNDING = True     # tests may switch the backward rounding off to measure what it explains


def qa(x, sim, fwd=True):
    """activation storage point: round the value (if fwd) and, in the backward pass, its gradient"""
    if not sim:
        return x
    if not SIM_GRAD_ROUNDING or not x.requires_grad:
        return x.to(torch.bfloat16).to(torch.float32) if fwd else x
    return _RoundBoth.apply(x, fwd)


def init_params(num_hid=512, nclasses=64, seed=3, specs=CONV_SPECS):
    """Initialisers of the reference: xavier-uniform conv weights, zero biases (network.py:168-169), BN gamma 1 /
    beta 0, LSTM glorot-uniform (TF variable-scope default), FC variance_scaling(0.01, FAN_AVG, normal)
    (network.py:119).  Values only need to be plausible: parity tests load the SAME arrays into both paths."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, kh, kw, ci, co, _, bn, _ in specs:
        lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
        p[name + "/weights"] = (torch.rand(kh, kw, ci, co, generator=g) * 2 - 1) * lim
        p[name + "/biases"] = torch.zeros(co)
        if bn:
            p["%s/%s/gamma" % (name, name)] = torch.ones(co)     # TF scope nesting: name/name/{gamma,beta}
            p["%s/%s/beta" % (name, name)] = torch.zeros(co)
    din = specs[-1][4]
    u = num_hid // 2
    for d in ("fw", "bw"):
        lim = math.sqrt(6.0 / (din + u + 4 * u))
        p["logits/%s/weights" % d] = (torch.rand(din + u, 4 * u, generator=g) * 2 - 1) * lim
        p["logits/%s/biases" % d] = torch.zeros(4 * u)
    std = math.sqrt(1.3 * 0.01 / ((num_hid + nclasses) / 2.0))
    p["logits/weights"] = torch.randn(num_hid, nclasses, generator=g) * std
    p["logits/biases"] = torch.zeros(nclasses)
    return p


def conv_single(x, w, b, padding, sim, first=False):
    """x [N, W, H, Cin] -> [N, W', H', Cout]; w HWIO [kh, kw, Cin, Cout]."""
    xin = x.permute(0, 3, 1, 2)
    wt = w.permute(3, 2, 0, 1)
    if not first:
        wt = q(wt, sim)          # conv1 is computed with fp32 weights on the device path too
    pad = (w.shape[0] // 2, w.shape[1] // 2) if padding == "SAME" else 0
    y = F.conv2d(xin, wt, None, stride=1, padding=pad)
    return y.permute(0, 2, 3, 1) + b


def batch_norm_train(z, gamma, beta, eps=BN_EPS):
    mu = z.mean(dim=(0, 1, 2))
    var = z.var(dim=(0, 1, 2), unbiased=False)
    return (z - mu) * torch.rsqrt(var + eps) * gamma + beta


def max_pool(x, kw, kh):
    y = F.max_pool2d(x.permute(0, 3, 1, 2), kernel_size=(kw, kh), stride=(kw, kh))
    return y.permute(0, 2, 3, 1)


def lstm_direction(x, seq_len, W, b, reverse, sim, forget_bias=1.0, state0=None, return_state=False):
    """x [N, T, D] (already bf16-rounded when sim); W [D+U, 4U] (TF LSTMCell), gate order i, j, f, o.
    state0 = (c0, h0) and return_state exist for the TensorFlow known-answer test of the cell equations
    (tests/test_oracle_graph.py::test_lstm_cell_equations_on_tensorflows_own_test_vector); the graph uses a zero state."""
    N, T, D = x.shape
    U = W.shape[1] // 4
    Wx, Wh = q(W[:D], sim), q(W[D:], sim)
    xproj = x.reshape(N * T, D) @ Wx + b
    xproj = xproj.reshape(N, T, 4 * U)
    h = torch.zeros(N, U)
    c = torch.zeros(N, U)
    if state0 is not None:
        c, h = state0
    outs = [None] * T
    lens = torch.as_tensor(seq_len)
    for s in range(T):
        # per-sample frame index of step s
        t_idx = (lens - 1 - s).clamp(min=0) if reverse else torch.full((N,), s, dtype=torch.long)
        active = (s < lens)
        xs = xproj[torch.arange(N), t_idx]
        z = qa(xs + h @ Wh, sim, fwd=False)              # d loss / d z is stored as bf16 (operand of the weight-gradient GEMMs)
        i, j, f, o = z.split(U, dim=1)
        cn = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
        hn = qa(torch.sigmoid(o) * torch.tanh(cn), sim)
        m = active.unsqueeze(1)
        c = torch.where(m, cn, c)
        h = torch.where(m, hn, h)
        outs[s] = (t_idx, active, hn)
    out = torch.zeros(N, T, U)
    # scatter step outputs to their frames (differentiable)
    rows = []
    for n in range(N):
        frames = [None] * T
        for s in range(T):
            t_idx, active, hn = outs[s]
            if bool(active[n]):
                frames[int(t_idx[n])] = hn[n]
        rows.append(torch.stack([fr if fr is not None else torch.zeros(U) for fr in frames]))
    if return_state:
        return torch.stack(rows), (c, h)
    return torch.stack(rows)


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_tnc, labels, label_len, seq_len):
        costs, grad = octc.ctc_loss_c(logits_tnc.detach().numpy(), labels, label_len, seq_len, blank=0)
        ctx.save_for_backward(torch.from_numpy(grad))
        return torch.from_numpy(costs.copy())

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout.view(1, -1, 1), None, None, None


def forward(params, x, seq_len, sim_bf16=False, specs=CONV_SPECS, pool_after=POOL_AFTER, keep=False):
    """x: [N, W, 32] float32 in [0,1] (gen.py:59-65 layout).  Returns logits [T, N, C] (time-major, network.py:127-128)
    and, if keep, a dict of intermediates keyed by layer name."""
    sim = sim_bf16
    inter = {}
    h = x.unsqueeze(3)                                  # conv_single c_i == 1 -> expand_dims (network.py:165)
    for idx, (name, kh, kw, ci, co, padding, bn, relu) in enumerate(specs):
        z = conv_single(h, params[name + "/weights"], params[name + "/biases"], padding, sim, first=(idx == 0))
        if bn:
            z = qa(z, sim)
            if keep: inter[name + "/pre_bn"] = z
            z = batch_norm_train(z, params["%s/%s/gamma" % (name, name)], params["%s/%s/beta" % (name, name)])
        if relu:
            z = torch.relu(z)
        h = qa(z, sim)
        if keep: inter[name] = h
        if name in pool_after:
            kw_, kh_ = pool_after[name]
            h = qa(max_pool(h, kw_, kh_), sim, fwd=False)        # the pooled map's gradient is a bf16 tensor on the device
            if keep: inter[name + "/pool"] = h
    N, A, B, D = h.shape
    feat = h.reshape(N, A * B, D)                       # reshape_squeeze_layer (network.py:361-368)
    fw = lstm_direction(feat, seq_len, params["logits/fw/weights"], params["logits/fw/biases"], False, sim)
    bw = lstm_direction(feat, seq_len, params["logits/bw/weights"], params["logits/bw/biases"], True, sim)
    hcat = torch.cat([fw, bw], dim=2)                   # network.py:109
    if keep: inter["lstm_out"] = hcat
    T = A * B
    logits = hcat.reshape(N * T, -1) @ q(params["logits/weights"], sim) + params["logits/biases"]
    logits = qa(logits.reshape(N, T, -1).permute(1, 0, 2).contiguous(), sim, fwd=False)   # CTC gradient handed over as bf16
    return (logits, inter) if keep else logits


REGULARISED = lambda name: name.endswith("/weights") and "/fw/" not in name and "/bw/" not in name


def loss_fn(params, x, labels, label_len, seq_len, weight_decay, sim_bf16=False, **kw):
    logits = forward(params, x, seq_len, sim_bf16, **kw)
    costs = _CTC.apply(logits, np.asarray(labels, np.int32), np.asarray(label_len, np.int32), np.asarray(seq_len, np.int32))
    ctc = costs.mean()                                  # tf.reduce_mean (network.py:655)
    reg = 0.0
    if weight_decay > 0:
        for k, v in params.items():
            if REGULARISED(k):
                reg = reg + weight_decay * 0.5 * (v * v).sum()     # tf.nn.l2_loss = sum(w^2)/2 (network.py:636)
    return ctc + reg, ctc, logits


def clip_by_global_norm(grads, clip=10.0):
    norm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    scale = clip / max(norm, clip)
    return {k: g * scale for k, g in grads.items()}, norm


def adam_step(params, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """TF-1.0 AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for k in params:
        m = state.setdefault("m/" + k, torch.zeros_like(params[k]))
        v = state.setdefault("v/" + k, torch.zeros_like(params[k]))
        m.mul_(beta1).add_(grads[k], alpha=1 - beta1)
        v.mul_(beta2).addcmul_(grads[k], grads[k], value=1 - beta2)
        params[k] = params[k] - lr_t * m / (v.sqrt() + eps)
    return params


def momentum_step(params, grads, state, lr, momentum=0.9):
    """TF-1.0 MomentumOptimizer (train.py:76, the reference's fallback solver): acc = momentum * acc + g; w -= lr * acc."""
    for k in params:
        acc = state.setdefault("acc/" + k, torch.zeros_like(params[k]))
        acc.mul_(momentum).add_(grads[k])
        params[k] = params[k] - lr * acc
    return params


def rmsprop_step(params, grads, state, lr, decay=0.9, eps=1e-10):
    """TF-1.0 RMSPropOptimizer with momentum 0 (train.py:75): the `rms` slot starts at ONES; ms = decay * ms + (1 - decay) g^2;
    w -= lr * g / sqrt(ms + eps) (epsilon INSIDE the root)."""
    for k in params:
        ms = state.setdefault("rms/" + k, torch.ones_like(params[k]))
        ms.mul_(decay).addcmul_(grads[k], grads[k], value=1 - decay)
        params[k] = params[k] - lr * grads[k] / (ms + eps).sqrt()
    return params


def train_step(params, state, batch, lr, weight_decay, clip=10.0, sim_bf16=False):
    x, labels, label_len, seq_len = batch
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    total, ctc, logits = loss_fn(leaves, x, labels, label_len, seq_len, weight_decay, sim_bf16)
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    grads, norm = clip_by_global_norm(grads, clip)
    new = adam_step({k: v.detach() for k, v in leaves.items()}, grads, state, lr)
    return new, float(total), float(ctc), norm, logits.detach()
