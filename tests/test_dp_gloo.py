"""Multi-GPU path on CPU: two gloo ranks run the data-parallel recipe the engine uses (each rank scales its local
CTC gradient by 1/(local_batch*world), SUM all-reduce of ONE flat gradient buffer, identical clip+Adam afterwards) and
must reproduce the single-process gradient of the global batch.  Uses a BN-free reduced graph, because batch-norm
statistics are (deliberately, like the reference at bs 64/replica) per rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lstm_ctc_ocr_amd import dist as ocr_dist
from oracle import graph as og

SPECS = [("conv1", 3, 3, 1, 64, "SAME", False, True), ("conv2", 3, 3, 64, 32, "SAME", False, True),
         ("conv3", 3, 3, 32, 32, "SAME", False, True), ("conv4", 3, 3, 32, 32, "SAME", False, True),
         ("conv5", 2, 2, 32, 32, "VALID", False, False)]
POOLS = {"conv1": (2, 2), "conv2": (2, 2), "conv3": (1, 2), "conv4": (1, 2)}


def _batch():
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.rand(4, 16, 32).astype(np.float32))
    ll = np.array([1, 2, 1, 1], np.int32)
    labels = rng.randint(1, 15, int(ll.sum())).astype(np.int32)
    return x, labels, ll, [3, 3, 3, 2]


class _Spec(object):
    def __init__(self, name, shape):
        self.name, self.shape, self.regularized = name, tuple(shape), og.REGULARISED(name)


def _layout(params):
    """The engine's flat gradient layout (lstm_ctc_ocr_amd/layout.py) for this reduced graph: [early rest | early regularised |
    late regularised | late rest], late = the last layers in execution order holding >= 75 % of the parameters."""
    from lstm_ctc_ocr_amd.layout import FlatLayout
    order = [s[0] for s in SPECS] + ['logits']
    specs = [_Spec(k, v.shape) for k, v in params.items()]
    specs.sort(key=lambda s: order.index(s.name.split('/')[0]))            # definition order = forward order
    return FlatLayout(specs, 64, order=order)


def _flat_grads(params, x, labels, ll, sl, scale, lay):
    """Gradient of scale * sum(costs) packed into the engine's flat buffer layout."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    logits = og.forward(leaves, x, sl, specs=SPECS, pool_after=POOLS)
    costs = og._CTC.apply(logits, np.asarray(labels, np.int32), np.asarray(ll, np.int32), np.asarray(sl, np.int32))
    (costs.sum() * scale).backward()
    flat = torch.zeros(lay.n_total)
    for k, v in leaves.items():
        flat[lay.offsets[k]:lay.offsets[k] + v.numel()] = v.grad.reshape(-1)
    return flat


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    params = og.init_params(num_hid=32, nclasses=16, seed=5, specs=SPECS)
    x, labels, ll, sl = _batch()
    lo, hi = rank * 2, rank * 2 + 2                                # shard the global batch of 4 over 2 ranks
    off = int(ll[:lo].sum())
    lay = _layout(params)
    g = _flat_grads(params, x[lo:hi], labels[off:off + int(ll[lo:hi].sum())], ll[lo:hi], sl[lo:hi], ocr_dist.loss_scale(2, world), lay)
    # Engine.train_step's exchange: the LATE bucket [late_begin, n_total) first (it overlaps the early layers' backward on the
    # device), then the early bucket [0, late_begin) — two all-reduces over contiguous views of ONE flat buffer
    assert 0 < lay.late_begin < lay.n_total and lay.split_layer is not None
    ocr_dist.allreduce_sum_(g[lay.late_begin:])
    ocr_dist.allreduce_sum_(g[:lay.late_begin])
    assert abs(ocr_dist.mean_scalar(float(rank)) - 0.5) < 1e-12              # the printed loss is the mean over the ranks
    if rank == 0:
        torch.save(g, out)
    assert ocr_dist.rank_seed(3, rank) != ocr_dist.rank_seed(3, rank + 1)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_global_batch_gradient(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'g.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    g2 = torch.load(out)
    params = og.init_params(num_hid=32, nclasses=16, seed=5, specs=SPECS)
    x, labels, ll, sl = _batch()
    lay = _layout(params)
    g1 = _flat_grads(params, x, labels, ll, sl, ocr_dist.loss_scale(4, 1), lay)
    assert g1.shape == g2.shape == (lay.n_total,)
    assert float((g1 - g2).abs().max()) < 1e-5 * max(1.0, float(g1.abs().max()))
    for k, v in params.items():                                               # both buckets really carry gradient
        assert float(g2[lay.offsets[k]:lay.offsets[k] + v.numel()].abs().max()) > 0 or k.endswith('biases')
    # clip + Adam applied to identical reduced gradients stays identical on every rank by construction
    assert ocr_dist.loss_scale(64, 8) == 1.0 / 512
