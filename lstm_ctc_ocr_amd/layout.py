"""Flat parameter / gradient layout of an Engine (pure host logic, no device needed — covered by the CPU test-suite).

    [early rest | early regularised | late regularised | late rest]

* the L2-regularised tensors (conv + FC weights, network.py:170-171,123-124) form ONE contiguous range, whose bounds the
  optimiser kernels take (`ocr_optim_step(..., reg_begin, reg_end, ...)`);
* the tensors of the LATE layers — the last layers of the network (in definition order) that together hold at least
  `late_fraction` of the parameters — form the upper part of the buffer.  Their gradients are complete first in the backward
  pass, so data parallelism all-reduces [late_begin, n_total) as one contiguous bucket while the early layers' backward is
  still running, and [0, late_begin) afterwards (Engine.train_step).
"""
import numpy as np


def round_up(x, m):
    return (int(x) + m - 1) // m * m


def layer_of(name):
    return name.split('/')[0]


class FlatLayout(object):
    def __init__(self, specs, align, late_fraction=0.75):
        """specs: iterable of objects with .name, .shape, .regularized in definition (= forward) order."""
        specs = list(specs)
        layers = []
        for s in specs:
            if layer_of(s.name) not in layers:
                layers.append(layer_of(s.name))
        size = {l: 0 for l in layers}
        for s in specs:
            size[layer_of(s.name)] += int(np.prod(s.shape))
        total, acc, split = sum(size.values()), 0, 0
        for i in range(len(layers) - 1, -1, -1):
            acc += size[layers[i]]
            split = i
            if acc >= late_fraction * total:
                break
        late = set(layers[split:]) if split > 0 else set()
        self.split_layer = layers[split] if split > 0 else None     # first late layer; None: no split (everything "early")
        is_late = lambda s: layer_of(s.name) in late
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
