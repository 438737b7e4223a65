"""Flat parameter / gradient layout of an Engine (pure host logic, no device needed — covered by the CPU test-suite).

    [early rest | early regularised | late regularised | late rest]

* the L2-regularised tensors (conv + FC weights, network.py:170-171,123-124) form ONE contiguous range, whose bounds the
  optimiser kernels take (`ocr_optim_step(..., reg_begin, reg_end, ...)`);
* the tensors of the LATE layers — the last layers of the network in EXECUTION order (`order`: the names of the lowered
  operators as the engine runs them; definition order when not given — the two differ on residual graphs, where a block's
  1x1 projection is defined after its convolutions but lowered before them) that together hold at least `late_fraction` of
  the parameters — form the upper part of the buffer.  Their gradients are complete first in the backward
  pass, so data parallelism all-reduces [late_begin, n_total) as one contiguous bucket while the early layers' backward is
  still running, and [0, late_begin) afterwards (Engine.train_step).
"""
import numpy as np


def execution_order(output_node):
    """Plan nodes reachable from `output_node` in the order the engine executes them: depth-first post-order over DATA edges
    (`add` and `concat` have several; the second input of bi_lstm is the time_step_len slot).  Input slots are not listed."""
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    seen, order = set(), []

    def visit(nd):
        if nd.op == 'input' or id(nd) in seen:
            return
        seen.add(id(nd))
        for i in (nd.inputs if nd.op in ('add', 'concat') else nd.inputs[:1]):
            visit(i)
        order.append(nd)

    visit(output_node)
    return order


def round_up(x, m):
    return (int(x) + m - 1) // m * m


def layer_of(name, layers=None):
    """Owning layer of a variable: the longest layer name that prefixes it ('logits/stack0/fw/weights' -> 'logits/stack0',
    'conv4_1/conv4_1/gamma' -> 'conv4_1'); without a layer list, the first path component."""
    if layers:
        best = None
        for l in layers:
            if name.startswith(l + '/') and (best is None or len(l) > len(best)):
                best = l
        if best is not None:
            return best
    return name.split('/')[0]


class FlatLayout(object):
    def __init__(self, specs, align, late_fraction=0.75, order=None):
        """specs: iterable of objects with .name, .shape, .regularized in definition (= forward) order.
        order: layer names in execution order (only those owning variables matter); every variable must belong to one."""
        specs = list(specs)
        known = list(order) if order is not None else None
        if known is not None:
            layers = []
            for l in known:
                if l not in layers and any(layer_of(s.name, known) == l for s in specs):
                    layers.append(l)
            orphans = [s.name for s in specs if layer_of(s.name, known) not in layers]
            if orphans:
                raise ValueError('variables without a lowered operator: %s' % orphans)
        else:
            layers = []
            for s in specs:
                if layer_of(s.name) not in layers:
                    layers.append(layer_of(s.name))
        owner = {s.name: layer_of(s.name, known) for s in specs}
        self.owner = owner
        size = {l: 0 for l in layers}
        for s in specs:
            size[owner[s.name]] += int(np.prod(s.shape))
        total, acc, split = sum(size.values()), 0, 0
        for i in range(len(layers) - 1, -1, -1):
            acc += size[layers[i]]
            split = i
            if acc >= late_fraction * total:
                break
        late = set(layers[split:]) if split > 0 else set()
        self.split_layer = layers[split] if split > 0 else None     # first late layer; None: no split (everything "early")
        is_late = lambda s: owner[s.name] in late
        self.late_layers = late
        groups = ([s for s in specs if not s.regularized and not is_late(s)], [s for s in specs if s.regularized and not is_late(s)],
                  [s for s in specs if s.regularized and is_late(s)], [s for s in specs if not s.regularized and is_late(s)])
        self.specs, self.offsets = {}, {}
        off, bounds = 0, []
        for grp in groups:
            bounds.append(off)
            for s in grp:
                self.specs[s.name] = s
                self.offsets[s.name] = off
                off += round_up(int(np.prod(s.shape)), align)
        bounds.append(off)
        self.n_total = off
        self.reg_range = (bounds[1], bounds[3])        # [early regularised | late regularised]
        self.late_begin = bounds[2]                    # gradients of [late_begin, n_total) are complete after backward part 1


def init_host_parameters(specs, offsets, n_total, seed):
    """Initial values of every variable, as ONE flat fp32 host tensor in the engine's layout.  Pure host code (torch CPU
    generator): the engine and the CPU golden-vector scripts draw identical parameters from the same seed.
    Initialisers of the reference: xavier-uniform conv weights, zero biases (network.py:168-169), BN gamma 1 / beta 0 /
    moving_mean 0 / moving_variance 1, LSTMCell matrix glorot-uniform (TF variable-scope default), FC
    variance_scaling(0.01, FAN_AVG, truncated normal) (network.py:119)."""
    import math
    import torch
    g = torch.Generator().manual_seed(int(seed))
    host = torch.zeros(n_total, dtype=torch.float32)
    for name, s in specs.items():
        n = int(np.prod(s.shape))
        init = s.init
        if init == 'zeros':
            v = torch.zeros(n)
        elif init == 'ones':
            v = torch.ones(n)
        elif init == 'xavier_uniform':          # tf.contrib.layers.xavier_initializer (network.py:168)
            kh, kw, ci, co = s.shape
            lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
            v = (torch.rand(n, generator=g) * 2 - 1) * lim
        elif init == 'glorot_uniform':          # TF variable-scope default for the LSTMCell matrix
            lim = math.sqrt(6.0 / (s.shape[0] + s.shape[1]))
            v = (torch.rand(n, generator=g) * 2 - 1) * lim
        elif isinstance(init, tuple) and init[0] == 'variance_scaling':   # factor, FAN_AVG, truncated normal (network.py:119)
            std = math.sqrt(1.3 * init[1] / ((s.shape[0] + s.shape[1]) / 2.0))
            v = torch.fmod(torch.randn(n, generator=g), 2.0) * std
        elif isinstance(init, tuple) and init[0] == 'truncated_normal':   # tf.truncated_normal_initializer(stddev) (network.py:144)
            v = torch.fmod(torch.randn(n, generator=g), 2.0) * float(init[1])
        else:
            raise ValueError('unknown initializer %r for %s' % (init, name))
        host[offsets[name]:offsets[name] + n] = v
    return host


def host_parameters(net, seed, align=64):
    """{TF variable name: fp32 tensor} exactly as Engine(net, seed=seed) initialises them — without a GPU."""
    order = [nd.name for nd in execution_order(net.get_output('logits'))]
    lay = FlatLayout(net.param_specs.values(), align, order=order)
    flat = init_host_parameters(lay.specs, lay.offsets, lay.n_total, seed)
    return {name: flat[lay.offsets[name]:lay.offsets[name] + int(np.prod(s.shape))].view(s.shape).clone() for name, s in lay.specs.items()}
