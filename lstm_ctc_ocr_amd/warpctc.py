"""`warpctc_tensorflow.ctc`-shaped operator over warp-ctc's own C ABI as exported by libocrhip.so (include/warpctc_abi.h).

The reference calls `warpctc_tensorflow.ctc(activations=logits, flat_labels=labels, label_lengths=label_len,
input_lengths=time_step_batch)` (lib/networks/network.py:653-654) and differentiates through it.  Here the same call takes a
device tensor of activations [T, N, C] (f32, unnormalised) and host label / length arrays, goes through `compute_ctc_loss`
with warp-ctc's exact prototype (ctcOptions by value, host labels, host costs) and returns the per-sample costs as a tensor
that autograd can differentiate (d cost_n / d activations, scaled by the incoming gradient, like the registered TF gradient).
The training engine does NOT use this entry point (it keeps labels and costs in HBM and never synchronises —
ocr_ctc_loss_train); this one exists so that code written against warp-ctc finds the interface it expects.
"""
import ctypes

import numpy as np
import torch

from . import _native as nat

CTC_CPU, CTC_GPU = 0, 1


class _Where(ctypes.Union):
    _fields_ = [("num_threads", ctypes.c_uint), ("stream", ctypes.c_void_p)]


class ctcOptions(ctypes.Structure):
    """struct ctcOptions of warp-ctc's ctc.h: {ctcComputeLocation loc; union {unsigned num_threads; stream}; int blank_label}"""
    _anonymous_ = ("where",)
    _fields_ = [("loc", ctypes.c_int), ("where", _Where), ("blank_label", ctypes.c_int)]


_bound = False


def _lib():
    global _bound
    lib = nat.lib()
    if not _bound:
        ip = ctypes.POINTER(ctypes.c_int)
        lib.compute_ctc_loss.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ip, ip, ip, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctcOptions]
        lib.compute_ctc_loss.restype = ctypes.c_int
        lib.get_workspace_size.argtypes = [ip, ip, ctypes.c_int, ctypes.c_int, ctcOptions, ctypes.POINTER(ctypes.c_size_t)]
        lib.get_workspace_size.restype = ctypes.c_int
        lib.ctcGetStatusString.argtypes = [ctypes.c_int]
        lib.ctcGetStatusString.restype = ctypes.c_char_p
        lib.get_warpctc_version.restype = ctypes.c_int
        _bound = True
    return lib


def _check(status):
    if status != 0:
        raise nat.NativeError("warp-ctc ABI call failed: status %d (%s)" % (status, _lib().ctcGetStatusString(status).decode()))


def _iarr(a):
    a = np.ascontiguousarray(a.cpu().numpy() if torch.is_tensor(a) else a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def compute(activations, flat_labels, label_lengths, input_lengths, blank_label=0, want_grad=True, loc=CTC_GPU):
    """-> (costs float32 [N] on the host, gradients [T, N, C] on the device or None)."""
    if not (torch.is_tensor(activations) and activations.is_cuda and activations.dtype == torch.float32):
        raise nat.NativeError("activations must be a float32 device tensor [T, N, C] (there is no CPU path)")
    acts = activations.contiguous()
    T, N, C = acts.shape
    lab, plab = _iarr(flat_labels)
    ll, pll = _iarr(label_lengths)
    il, pil = _iarr(input_lengths)
    if int(il.max()) != T:
        raise ValueError("warp-ctc addresses activations with maxT = max(input_lengths) = %d, tensor has T = %d" % (int(il.max()), T))
    opt = ctcOptions()
    opt.loc, opt.blank_label = loc, int(blank_label)
    opt.stream = torch.cuda.current_stream(acts.device).cuda_stream
    lib = _lib()
    size = ctypes.c_size_t(0)
    _check(lib.get_workspace_size(pll, pil, C, N, opt, ctypes.byref(size)))
    ws = torch.empty(size.value, dtype=torch.uint8, device=acts.device)
    grads = torch.empty_like(acts) if want_grad else None
    costs = np.zeros(N, np.float32)
    _check(lib.compute_ctc_loss(acts.data_ptr(), grads.data_ptr() if want_grad else None, plab, pll, pil, C, N,
                                costs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ws.data_ptr(), opt))
    return torch.from_numpy(costs), grads


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, activations, flat_labels, label_lengths, input_lengths, blank_label):
        costs, grads = compute(activations.detach(), flat_labels, label_lengths, input_lengths, blank_label, want_grad=True)
        ctx.save_for_backward(grads)
        return costs.to(activations.device)

    @staticmethod
    def backward(ctx, grad_costs):
        (grads,) = ctx.saved_tensors
        return grads * grad_costs.view(1, -1, 1), None, None, None, None


def ctc(activations, flat_labels, label_lengths, input_lengths, blank_label=0):
    """warpctc_tensorflow.ctc: per-sample costs [N] (device tensor, differentiable w.r.t. activations)."""
    return _CTC.apply(activations, flat_labels, label_lengths, input_lengths, blank_label)
