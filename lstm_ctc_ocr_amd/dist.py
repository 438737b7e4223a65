"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl") on the GPU box,
gloo in the CPU test-suite.  The path has exactly ONE exchange per step — the flat fp32 gradient buffer — so this
module is deliberately tiny."""
import os


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def loss_scale(local_batch, world):
    """d(total loss)/d(cost_n): the reference's loss is the MEAN cost over the batch (network.py:655); with the
    global batch sharded over `world` ranks every rank scales its local gradient by 1/(local_batch*world) so that a
    plain SUM all-reduce yields the global-batch mean gradient."""
    return 1.0 / (float(local_batch) * float(world))


def allreduce_sum_(flat, group=None, force=False):
    import torch.distributed as dist
    fake = os.environ.get('OCR_FAKE_WORLD')
    if fake:                # single-GPU emulation of `fake` ranks holding identical data: the sum is a multiplication
        flat.mul_(float(fake))      # (a real kernel on the current stream, so stream ordering is exercised like RCCL's)
        # ... and, with OCR_FAKE_COMM_CUS=n, n workgroups that HOLD CUs for as long as a ring all-reduce of this range would run
        # (OCR_FAKE_COMM_US = microseconds per 25 MB, scaled by the range; >= 20 us): the doubling kernel alone is a ~10 us elementwise
        # pass and cannot show what RCCL's channel kernels do to the 256-tile convolution grids of the concurrent backward graph
        cus = int(os.environ.get('OCR_FAKE_COMM_CUS', '0') or 0)
        if cus > 0 and flat.is_cuda:
            from . import _native as nat
            us = max(20.0, float(os.environ.get('OCR_FAKE_COMM_US', '250')) * flat.numel() * flat.element_size() / 25e6)
            lds = int(os.environ.get('OCR_FAKE_COMM_LDS_KB', '96')) * 1024      # > half a CU's 160 KB: no convolution tile fits beside it
            nat.call("ocr_occupy_cus", cus, 256, lds, us, nat.stream())
        return flat
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        if flat.is_cuda and dist.get_backend(group) == 'gloo':
            # gloo (tests: several ranks sharing one GPU) is staged through the host on the CURRENT stream, so the ordering against
            # the producing / consuming kernels is the same as with the RCCL call
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def rank_seed(base_seed, rank):
    """Independent, reproducible data stream per rank."""
    return int(base_seed) + 1000003 * int(rank)


def mean_scalar(value, device=None, group=None):
    """Mean of a Python float over the ranks (every rank calls it at the same iteration)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    backend = dist.get_backend(group)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)
