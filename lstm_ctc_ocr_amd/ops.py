"""Thin tensor-level wrappers over the C ABI (include/ocr_hip.h).

torch is used for allocation and stream handles only; every function enqueues hand-written HIP kernels on torch's
current stream and returns device tensors.  No function here has a CPU or eager fallback.
"""
import ctypes

import torch

from . import _native as nat
from ._native import EPI_ACCUM, EPI_BIAS, EPI_MASK, EPI_OUT_F32, EPI_RELU, EPI_ROWSWAP, call, ptr

BF16 = torch.bfloat16
F32 = torch.float32


def _st():
    return nat.stream()


def _dev(t):
    if not t.is_cuda:
        raise nat.NativeError("hot-path operators take device tensors only (got a CPU tensor)")
    return t


# ----------------------------------------------------------------------------------------------- CTC
def ctc_workspace_bytes(max_label_len, max_time, minibatch):
    sz = ctypes.c_size_t(0)
    call("ocr_ctc_workspace_size", max_label_len, max_time, minibatch, ctypes.byref(sz))
    return sz.value


def ctc_loss(acts, flat_labels, label_lengths, input_lengths, max_label_len, blank=0, want_grad=True,
             workspace=None, costs=None, grads=None):
    """warp-ctc shaped: acts f32 [T, N, C] unnormalised; labels/lengths int32 device tensors."""
    T, N, C = acts.shape
    _dev(acts)
    if workspace is None:
        workspace = torch.empty(ctc_workspace_bytes(max_label_len, T, N), dtype=torch.uint8, device=acts.device)
    if costs is None:
        costs = torch.empty(N, dtype=F32, device=acts.device)
    if want_grad and grads is None:
        grads = torch.empty_like(acts)
    call("ocr_ctc_loss", ptr(acts), ptr(grads) if want_grad else None, ptr(flat_labels), ptr(label_lengths),
         ptr(input_lengths), C, N, T, max_label_len, blank, ptr(costs), ptr(workspace), _st())
    return costs, (grads if want_grad else None)


def ctc_train_supported(C, T, max_label_len):
    return bool(nat.lib().ocr_ctc_train_supported(C, T, max_label_len))


def ctc_loss_train(acts, grad_ntc_bf16, scale, flat_labels, label_lengths, input_lengths, max_label_len, costs, blank=0):
    """costs + scale * d cost / d acts written as bf16 [N, T, C] in one launch."""
    T, N, C = acts.shape
    call("ocr_ctc_loss_train", ptr(_dev(acts)), ptr(grad_ntc_bf16), float(scale), ptr(flat_labels), ptr(label_lengths),
         ptr(input_lengths), C, N, T, max_label_len, blank, ptr(costs), _st())
    return costs


def ctc_greedy_decode(acts, input_lengths, blank=0, pad_value=0):
    T, N, C = acts.shape
    out = torch.empty((N, T), dtype=torch.int32, device=acts.device)
    lens = torch.empty(N, dtype=torch.int32, device=acts.device)
    call("ocr_ctc_greedy_decode", ptr(_dev(acts)), ptr(input_lengths), C, N, T, blank, pad_value, ptr(out), ptr(lens), _st())
    return out, lens


def ctc_beam_decode(acts, input_lengths, beam_width=100, merge_repeated=True, pad_value=0, workspace=None):
    """tf.nn.ctc_beam_search_decoder semantics (blank = C-1).  Returns (dense [N, T] int32, lengths, neg_log_prob)."""
    T, N, C = acts.shape
    sz = ctypes.c_size_t(0)
    call("ocr_ctc_beam_workspace_size", C, N, T, beam_width, ctypes.byref(sz))
    if workspace is None or workspace.numel() < sz.value:
        workspace = torch.empty(sz.value, dtype=torch.uint8, device=acts.device)
    out = torch.empty((N, T), dtype=torch.int32, device=acts.device)
    lens = torch.empty(N, dtype=torch.int32, device=acts.device)
    nlp = torch.empty(N, dtype=F32, device=acts.device)
    call("ocr_ctc_beam_decode", ptr(_dev(acts)), ptr(input_lengths), C, N, T, beam_width, int(merge_repeated), pad_value,
         ptr(out), ptr(lens), ptr(nlp), ptr(workspace), workspace.numel(), _st())
    return out, lens, nlp


# ----------------------------------------------------------------------------------------------- GEMMs
def gemm_nt(P, Q, out=None, *, M=None, N=None, K=None, ldp=None, ldq=None, ldo=None, bias=None, relu=False,
            out_f32=False, mask=None, accumulate=False, splits=1, row_group=0, row_skip=0, rowswap=None):
    """out[m][n] = sum_k P[m][k] Q[n][k]; P, Q bf16 (K contiguous)."""
    M = P.shape[0] if M is None else M
    K = P.shape[1] if K is None else K
    N = Q.shape[0] if N is None else N
    ldp = P.stride(0) if ldp is None else ldp
    ldq = Q.stride(0) if ldq is None else ldq
    if out is None:
        out = torch.empty((M, N), dtype=F32 if out_f32 else BF16, device=P.device)
    ldo = out.stride(0) if ldo is None else ldo
    flags = 0
    if bias is not None: flags |= EPI_BIAS
    if relu: flags |= EPI_RELU
    if out.dtype == F32: flags |= EPI_OUT_F32
    if mask is not None: flags |= EPI_MASK
    if accumulate: flags |= EPI_ACCUM
    si = so = 0
    if rowswap is not None:
        flags |= EPI_ROWSWAP
        si, so = rowswap
    call("ocr_gemm_nt_bf16", ptr(_dev(P)), ldp, ptr(Q), ldq, ptr(out), ldo, M, N, K, ptr(bias), ptr(mask),
         mask.stride(0) if mask is not None else 0, flags, splits, row_group, row_skip, si, so, _st())
    return out


CONV_KERNEL_NAMES = ("gemm", "conv_halo", "conv_k2/A", "conv_k2/D", "conv_k3/A", "conv_k3/D", "conv_k3w/A", "conv_k3w/D", "conv_ws")


def conv3x3_kernel_choice(Nb, W, H, Cin, Cout, *, bias=True, relu=True, mask=False, accumulate=False, pool=(0, 0)):
    """Name of the kernel family ocr_conv3x3_bf16 / ocr_conv3x3_relu_pool_bf16 take for this shape (host-only query, no GPU needed)."""
    flags = (nat.EPI_BIAS if bias else 0) | (nat.EPI_RELU if relu else 0) | (nat.EPI_MASK if mask else 0) | (nat.EPI_ACCUM if accumulate else 0)
    rc = nat.lib().ocr_conv3x3_kernel_choice(int(Nb), int(W), int(H), int(Cin), int(Cout), flags, int(pool[0]), int(pool[1]))
    if rc < 0:
        raise nat.NativeError("ocr_conv3x3_kernel_choice failed: status %d (%s)" % (-rc, nat.status_string(-rc)))
    return CONV_KERNEL_NAMES[rc]


def conv3x3_accum_supported(Nb, W, H, Cin, Cout):
    return bool(nat.lib().ocr_conv3x3_accum_supported(Nb, W, H, Cin, Cout))


def conv3x3(x, wpack, out=None, *, bias=None, relu=False, mask=None, accumulate=False):
    """x bf16 [Nb, W, H, Cin]; wpack bf16 [Cout, 3, 3, Cin] -> bf16 [Nb, W, H, Cout]  (accumulate: out += result)."""
    Nb, W, H, Cin = x.shape
    Cout = wpack.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty((Nb, W, H, Cout), dtype=BF16, device=x.device)
    flags = ((EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0) | (EPI_MASK if mask is not None else 0) |
             (EPI_ACCUM if accumulate else 0))
    call("ocr_conv3x3_bf16", ptr(_dev(x)), ptr(wpack), ptr(out), Nb, W, H, Cin, Cout, ptr(bias), ptr(mask), flags, _st())
    return out


def conv3x3_pool_supported(Nb, W, H, Cin, Cout, kw, kh):
    return bool(nat.lib().ocr_conv3x3_pool_supported(Nb, W, H, Cin, Cout, kw, kh))


def conv3x3_relu_pool(x, wpack, out, pooled, bias, kw, kh):
    """out = relu(conv3x3(x) + bias) and pooled = max_pool(out, window (kw along W, kh along H)) from one launch."""
    Nb, W, H, Cin = x.shape
    call("ocr_conv3x3_relu_pool_bf16", ptr(_dev(x)), ptr(wpack), ptr(out), ptr(pooled), Nb, W, H, Cin, wpack.shape[0], ptr(bias),
         kw, kh, _st())
    return out, pooled


def gemm_tn(A, B, out, *, Mk=None, I=None, J=None, lda=None, ldb=None, ldo=None, row_group=0, row_skip=0,
            a_row_off=0, scale=1.0, splits=0, colsum=None):
    """out[I][J] (f32) += scale * A^T B;  colsum[J] += scale * column sums of B (bias gradient) when given."""
    Mk = B.shape[0] if Mk is None else Mk
    I = A.shape[1] if I is None else I
    J = B.shape[1] if J is None else J
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldo = out.stride(0) if ldo is None else ldo
    call("ocr_gemm_tn_bf16", ptr(_dev(A)), lda, ptr(B), ldb, ptr(out), ldo, Mk, I, J, row_group, row_skip, a_row_off,
         float(scale), splits, ptr(colsum), _st())
    return out


def gemm_tn_batched(A, lda, strideA, B, ldb, strideB, out, ldo, strideOut, Mk, I, J, nbatch, colsum=None, strideColsum=0,
                    scale=1.0, splits=0):
    """nbatch independent out_b[I][J] += scale * A_b^T B_b in one launch (element strides between the problems)."""
    call("ocr_gemm_tn_batched_bf16", ptr(_dev(A)), lda, strideA, ptr(B), ldb, strideB, ptr(out), ldo, strideOut, Mk, I, J, nbatch,
         float(scale), splits, ptr(colsum), strideColsum, _st())
    return out


class TnJob(ctypes.Structure):
    """ocr_tn_job (include/ocr_hip.h): one batched product out_b[I][J] += scale * A_b^T B_b (+ column sums of B_b) of ocr_gemm_tn_jobs_bf16."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_long), ("strideA", ctypes.c_long),
                ("B", ctypes.c_void_p), ("ldb", ctypes.c_long), ("strideB", ctypes.c_long),
                ("out", ctypes.c_void_p), ("ldo", ctypes.c_long), ("strideOut", ctypes.c_long),
                ("colsum", ctypes.c_void_p), ("strideColsum", ctypes.c_long),
                ("Mk", ctypes.c_int), ("I", ctypes.c_int), ("J", ctypes.c_int), ("nbatch", ctypes.c_int),
                ("row_group", ctypes.c_int), ("row_skip", ctypes.c_int), ("scale", ctypes.c_float), ("reserved", ctypes.c_int)]


def tn_job(A, lda, B, ldb, out, ldo, Mk, I, J, *, nbatch=1, strideA=0, strideB=0, strideOut=0, colsum=None, strideColsum=0, row_group=0,
           row_skip=0, scale=1.0):
    """Descriptor of a plain weight-gradient product for gemm_tn_jobs (the tensors must stay alive and untouched until the launch)."""
    return TnJob(ptr(_dev(A)), lda, strideA, ptr(_dev(B)), ldb, strideB, ptr(out), ldo, strideOut, ptr(colsum), strideColsum,
                 Mk, I, J, nbatch, row_group, row_skip, float(scale), 0)


def gemm_tn_jobs_supported(jobs):
    arr = (TnJob * len(jobs))(*jobs)
    return bool(nat.lib().ocr_gemm_tn_jobs_supported(ctypes.cast(arr, ctypes.c_void_p), len(jobs)))


def gemm_tn_jobs(jobs):
    """1 or 2 products in ONE launch of the ping-pong kernel (no split over the contraction, no atomics)."""
    arr = (TnJob * len(jobs))(*jobs)
    call("ocr_gemm_tn_jobs_bf16", ctypes.cast(arr, ctypes.c_void_p), len(jobs), _st())


def lstm_xh(x2d, hout, seq_len, xh, Nb, T, D, U, ndir=2):
    call("ocr_lstm_xh", ptr(_dev(x2d)), ptr(hout), ptr(seq_len), ptr(xh), Nb, T, D, U, ndir, _st())
    return xh


def conv3x3_wgrad_workspace_bytes(Nb, W, H, Cin, Cout):
    """Scratch bytes of the slab (atomic-free, deterministic) weight-gradient kernel; 0 when the shape is not covered."""
    sz = ctypes.c_size_t(0)
    call("ocr_conv3x3_wgrad_workspace_size", Nb, W, H, Cin, Cout, ctypes.byref(sz))
    return sz.value


def conv3x3_wgrad(x, dy, dw, splits=0, dbias=None, workspace=None):
    """dw [3,3,Cin,Cout] f32 += weight gradient; dbias += column sums of dy.  With `workspace` (uint8 tensor of at least
    conv3x3_wgrad_workspace_bytes) the nine-tap slab kernel runs where it applies."""
    Nb, W, H, Cin = x.shape
    Cout = dy.shape[-1]
    if workspace is not None:
        call("ocr_conv3x3_wgrad_ws_bf16", ptr(_dev(x)), ptr(dy), ptr(dw), ptr(dbias), Nb, W, H, Cin, Cout, splits,
             ptr(workspace), workspace.numel(), _st())
    else:
        call("ocr_conv3x3_wgrad_bf16", ptr(_dev(x)), ptr(dy), ptr(dw), ptr(dbias), Nb, W, H, Cin, Cout, splits, _st())
    return dw


def conv3x3_wgrad_deferred(x, dy, dw, dbias, workspace):
    """The slab kernel only; the reduction dw += sum of the slabs (and dbias) is left to wgrad9_reduce_jobs at the end of the backward
    pass.  Returns (job, nblocks): `job` = the 64 bytes describing the pending reduction (struct W9ReduceJob) — or None when the shape
    is not covered and the whole weight gradient was computed at once.  `workspace` must not be touched until the jobs have run."""
    Nb, W, H, Cin = x.shape
    Cout = dy.shape[-1]
    job = (ctypes.c_ubyte * 64)()
    nblk, deferred = ctypes.c_int(0), ctypes.c_int(0)
    call("ocr_conv3x3_wgrad_defer_bf16", ptr(_dev(x)), ptr(dy), ptr(dw), ptr(dbias), Nb, W, H, Cin, Cout, ptr(workspace),
         workspace.numel(), ctypes.cast(job, ctypes.c_void_p), ctypes.byref(nblk), ctypes.byref(deferred), _st())
    return (bytes(job), nblk.value) if deferred.value else (None, 0)


def wgrad9_reduce_jobs(table, njobs, total_blocks):
    call("ocr_wgrad9_reduce_jobs", ptr(_dev(table)), njobs, total_blocks, _st())


# ----------------------------------------------------------------------------------------------- HBM-bound layers
def conv1_fwd(x, w, bias, relu=True, out=None):
    Nb, W, H = x.shape
    Cout = w.shape[-1]
    if out is None:
        out = torch.empty((Nb, W, H, Cout), dtype=BF16, device=x.device)
    call("ocr_conv1_fwd", ptr(_dev(x)), ptr(w), ptr(bias), ptr(out), Nb, W, H, Cout, int(relu), _st())
    return out


def conv1_wgrad(x, dz, dw, db):
    Nb, W, H = x.shape
    call("ocr_conv1_wgrad", ptr(_dev(x)), ptr(dz), ptr(dw), ptr(db), Nb, W, H, dz.shape[-1], _st())


def eltwise(op, a, b, out):
    """op 0: out = a + b ; 1: out = relu(a) ; 2: out = (b > 0) ? a : 0 ; 3: out = relu(a + b) ; 4: out += (b > 0) ? a : 0
    (bf16, numel % 8 == 0)"""
    call("ocr_eltwise_bf16", op, ptr(_dev(a)), ptr(b), ptr(out), a.numel(), _st())
    return out


def conv1_pool_fwd(x, w, bias, out=None, zero=None, codes=None, ones=None):
    """zero (fp32 tensor, numel % 4 == 0): cleared by the same launch (the step's flat gradient buffer).  codes (int32 [Nb * W/2 * H/2, 8]):
    receives the pool routing + ReLU bits for conv1_pool_bwd(codes=...).  ones (int32 tensor, numel % 4 == 0): set to 0xFFFFFFFF by the
    same launch (the hand-off blocks of the step's persistent LSTM launches, lstm_*_seq(prepared=True))."""
    Nb, W, H = x.shape
    Cout = w.shape[-1]
    if out is None:
        out = torch.empty((Nb, W // 2, H // 2, Cout), dtype=BF16, device=x.device)
    if zero is not None or codes is not None or ones is not None:
        call("ocr_conv1_pool_fwd_train", ptr(_dev(x)), ptr(w), ptr(bias), ptr(out), Nb, W, H, Cout, ptr(codes),
             ptr(_dev(zero)) if zero is not None else None, 0 if zero is None else zero.numel(),
             ptr(_dev(ones)) if ones is not None else None, 0 if ones is None else ones.numel(), _st())
    else:
        call("ocr_conv1_pool_fwd", ptr(_dev(x)), ptr(w), ptr(bias), ptr(out), Nb, W, H, Cout, _st())
    return out


def conv1_pool_bwd(x, w, bias, dp, dw, db, codes=None):
    """codes: what conv1_pool_fwd(codes=...) saved — the gradient is routed with them (bit-identical) instead of recomputing the windows."""
    Nb, W, H = x.shape
    if codes is not None:
        call("ocr_conv1_pool_bwd_codes", ptr(_dev(x)), ptr(w), ptr(bias), ptr(dp), ptr(dw), ptr(db), Nb, W, H, w.shape[-1], ptr(codes), _st())
    else:
        call("ocr_conv1_pool_bwd", ptr(_dev(x)), ptr(w), ptr(bias), ptr(dp), ptr(dw), ptr(db), Nb, W, H, w.shape[-1], _st())


def conv1_pool_bwd_slab_rows(Nb, W, H):
    return int(nat.lib().ocr_conv1_pool_bwd_slab_rows(int(Nb), int(W), int(H)))


def conv1_pool_bwd_slab(x, w, bias, dp, slab, codes=None):
    """No atomics: block b's partial sums {dW [9][64] | db [64]} go to slab[b] (fp32 [rows, 640]); wgrad9_reduce_jobs adds the rows."""
    Nb, W, H = x.shape
    assert slab.dtype == torch.float32 and tuple(slab.shape) == (conv1_pool_bwd_slab_rows(Nb, W, H), 640)
    call("ocr_conv1_pool_bwd_slab", ptr(_dev(x)), ptr(w), ptr(bias), ptr(dp), Nb, W, H, w.shape[-1], ptr(codes), ptr(slab), _st())


def maxpool_fwd(x, kw, kh, out=None):
    Nb, W, H, C = x.shape
    if out is None:
        out = torch.empty((Nb, W // kw, H // kh, C), dtype=BF16, device=x.device)
    call("ocr_maxpool_fwd", ptr(_dev(x)), ptr(out), Nb, W, H, C, kw, kh, _st())
    return out


def maxpool_bwd(x, dy, kw, kh, relu_mask, out=None):
    Nb, W, H, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("ocr_maxpool_bwd", ptr(_dev(x)), ptr(dy), ptr(out), Nb, W, H, C, kw, kh, int(relu_mask), _st())
    return out


def bn_workspace(M, C, device):
    """Scratch block for bn_train_fwd / bn_train_bwd on [M, C] (ocr_bn_workspace_bytes)."""
    return torch.empty(int(nat.lib().ocr_bn_workspace_bytes(int(M), int(C))), dtype=torch.uint8, device=device)


def bn_train_fwd(x2d, gamma, beta, eps, relu, workspace, out=None, save_mean=None, save_rstd=None, residual=None, partial_rows=0,
                 pooled=None):
    """residual (bf16 [M, C]): out = [relu](bf16(bn(x)) + residual) — batch norm, add and relu of a residual block in one apply pass.
    partial_rows > 0: the workspace already holds that many partial statistics rows from the producing convolution (conv3x3_stats);
    pooled (bf16 [M / 2, C]): the 1 x 2 max-pool over row pairs that follows the layer, written by the apply pass."""
    M, C = x2d.shape
    if out is None: out = torch.empty_like(x2d)
    if save_mean is None: save_mean = torch.empty(C, dtype=F32, device=x2d.device)
    if save_rstd is None: save_rstd = torch.empty(C, dtype=F32, device=x2d.device)
    if partial_rows or pooled is not None:
        call("ocr_bn_train_fwd2", ptr(_dev(x2d)), ptr(out), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_rstd), M, C,
             float(eps), int(relu), ptr(workspace), ptr(residual), int(partial_rows), ptr(pooled), _st())
    else:
        call("ocr_bn_train_fwd", ptr(_dev(x2d)), ptr(out), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_rstd), M, C,
             float(eps), int(relu), ptr(workspace), ptr(residual), _st())
    return out, save_mean, save_rstd


def bn_train_bwd(x2d, y2d, dy2d, gamma, save_mean, save_rstd, dgamma, dbeta, relu, workspace, out=None, pooled_dy=False, partial_rows=0):
    """pooled_dy: dy2d is the gradient of the 1 x 2 max-pool behind the layer ([M / 2, C]); the passes route it themselves.
    partial_rows > 0: dy2d is already ReLU-masked and the workspace holds that many partial rows from conv3x3_dgrad_bnbwd."""
    M, C = x2d.shape
    if out is None: out = torch.empty_like(x2d)
    if pooled_dy or partial_rows:
        assert tuple(dy2d.shape) == ((M // 2, C) if pooled_dy else (M, C))
        call("ocr_bn_train_bwd2", ptr(_dev(x2d)), ptr(y2d), ptr(dy2d), ptr(out), ptr(gamma), ptr(save_mean), ptr(save_rstd),
             ptr(dgamma), ptr(dbeta), M, C, int(relu), ptr(workspace), int(bool(pooled_dy)), int(partial_rows), _st())
    else:
        call("ocr_bn_train_bwd", ptr(_dev(x2d)), ptr(y2d), ptr(dy2d), ptr(out), ptr(gamma), ptr(save_mean), ptr(save_rstd),
             ptr(dgamma), ptr(dbeta), M, C, int(relu), ptr(workspace), _st())
    return out


def conv3x3_stats_rows(Nb, W, H, Cin, Cout, *, bias=True, relu=False):
    """Partial-statistics rows the convolution's epilogue would write for this shape (Nb*W*H / 256), 0 where no kernel with that epilogue
    takes it (host-only query)."""
    flags = (EPI_BIAS if bias else 0) | (EPI_RELU if relu else 0)
    return int(nat.lib().ocr_conv3x3_stats_rows(int(Nb), int(W), int(H), int(Cin), int(Cout), flags))


def conv3x3_bnbwd_rows(Nb, W, H, Cin, Cout):
    """Partial rows conv3x3_dgrad_bnbwd writes for a data gradient of this shape (x = dy [Nb, W, H, Cin] -> dx [.., Cout]); 0: not covered."""
    return int(nat.lib().ocr_conv3x3_bnbwd_rows(int(Nb), int(W), int(H), int(Cin), int(Cout)))


def conv3x3_dgrad_bnbwd(dy, wdgrad, out, mask_y, z, mean, rstd, partials):
    """out = (mask_y > 0) ? conv3x3(dy, wdgrad) : 0 and partials[rows][2][Cout] = per-tile (sum out, sum out * (z - mean) * rstd): the data
    gradient into a batch-norm + ReLU layer together with that layer's batch-norm backward sums."""
    Nb, W, H, Cin = dy.shape
    Cout = wdgrad.shape[0]
    call("ocr_conv3x3_dgrad_bnbwd_bf16", ptr(_dev(dy)), ptr(wdgrad), ptr(out), Nb, W, H, Cin, Cout, ptr(mask_y), ptr(z), ptr(mean), ptr(rstd),
         ptr(partials), _st())
    return out


def conv3x3_stats(x, wpack, out, partials, *, bias=None, relu=False):
    """out = conv3x3(x) (+ bias, optional relu) and partials[rows][2][Cout] (fp32) = per-256-pixel-tile sums / sums of squares of out."""
    Nb, W, H, Cin = x.shape
    Cout = wpack.shape[0]
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0)
    call("ocr_conv3x3_bf16_stats", ptr(_dev(x)), ptr(wpack), ptr(out), Nb, W, H, Cin, Cout, ptr(bias), flags, ptr(partials), _st())
    return out


def bn_infer_fwd(x2d, gamma, beta, mean, var, eps, relu, out):
    M, C = x2d.shape
    call("ocr_bn_infer_fwd", ptr(_dev(x2d)), ptr(out), ptr(gamma), ptr(beta), ptr(mean), ptr(var), M, C, float(eps), int(relu), _st())
    return out


def bn_infer_bwd(x2d, y2d, dy2d, gamma, mean, var, dgamma, dbeta, eps, relu, out):
    M, C = x2d.shape
    call("ocr_bn_infer_bwd", ptr(_dev(x2d)), ptr(y2d), ptr(dy2d), ptr(out), ptr(gamma), ptr(mean), ptr(var), ptr(dgamma), ptr(dbeta),
         M, C, float(eps), int(relu), _st())
    return out


def dropout(src, dst, seed, step_counter, keep_prob):
    """dst = keep ? src / keep_prob : 0 with keep = hash(seed, *step_counter, index) < keep_prob (forward and backward alike)."""
    call("ocr_dropout_bf16", ptr(_dev(src)), ptr(dst), src.numel(), int(seed) & 0xffffffff, ptr(step_counter), float(keep_prob), _st())
    return dst


def avgpool(src, dst, Nb, W, H, C, kw, kh, backward=False):
    call("ocr_avgpool_bf16", ptr(_dev(src)), ptr(dst), Nb, W, H, C, kw, kh, int(backward), _st())
    return dst


def subsample(src, dst, Nb, W, H, C, Wo, Ho, sw, sh, ow, oh, backward=False):
    call("ocr_subsample_bf16", ptr(_dev(src)), ptr(dst), Nb, W, H, C, Wo, Ho, sw, sh, ow, oh, int(backward), _st())
    return dst


def softmax(src_f32, dst_f32):
    C = src_f32.shape[-1]
    call("ocr_softmax_f32", ptr(_dev(src_f32)), ptr(dst_f32), src_f32.numel() // C, C, _st())
    return dst_f32


def colsum(a2d, out, M=None, C=None, lda=None):
    M = a2d.shape[0] if M is None else M
    C = a2d.shape[1] if C is None else C
    lda = a2d.stride(0) if lda is None else lda
    call("ocr_colsum_bf16", ptr(_dev(a2d)), ptr(out), M, C, lda, _st())
    return out


def pack_transpose(w2d, out, lstm_units=0, R=None, Cc=None, ldin=None):
    R = w2d.shape[0] if R is None else R
    Cc = w2d.shape[1] if Cc is None else Cc
    ldin = w2d.stride(0) if ldin is None else ldin
    call("ocr_pack_transpose", ptr(_dev(w2d)), ptr(out), R, Cc, ldin, lstm_units, _st())
    return out


def pack_conv_dgrad(w, out):
    kh, kw, cin, cout = w.shape
    call("ocr_pack_conv_dgrad", ptr(_dev(w)), ptr(out), cin, cout, _st())
    return out


def pack_jobs(table, njobs, total_blocks):
    call("ocr_pack_jobs", ptr(_dev(table)), njobs, total_blocks, _st())


def cast_bf16(src, dst):
    call("ocr_cast_f32_bf16", ptr(_dev(src)), ptr(dst), src.numel(), _st())
    return dst


def u8_to_unit_f32(src_u8, dst_f32):
    """dst = src / 255 (fp32, correctly rounded) — device-side form of groupBatch's `astype(float32) / 255.`"""
    call("ocr_u8_to_unit_f32", ptr(_dev(src_u8)), ptr(dst_f32), src_u8.numel(), _st())
    return dst_f32


def bind_batch(pixels, x, seq_len, seq_len_dst, labels=None, labels_dst=None, labels_len=None, labels_len_dst=None):
    """One launch: device pixels (uint8 -> / 255, or fp32) -> x, plus the int32 vectors into their fixed buffers."""
    is_u8 = pixels.dtype == torch.uint8
    assert pixels.is_contiguous() and pixels.numel() == x.numel() and (is_u8 or pixels.dtype == torch.float32)
    for t in (seq_len, labels, labels_len):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous())
    nl = 0 if labels is None else labels.numel()
    call("ocr_bind_batch", ptr(_dev(pixels)), int(is_u8), ptr(x), pixels.numel(), ptr(_dev(seq_len)), ptr(seq_len_dst), seq_len.numel(),
         ptr(_dev(labels)) if nl else None, ptr(labels_dst) if nl else None, nl,
         ptr(_dev(labels_len)) if labels_len is not None else None, ptr(labels_len_dst) if labels_len is not None else None,
         0 if labels_len is None else labels_len.numel(), _st())


def captcha_synth(params, n_images, words_per_image, atlas, stamp, out, W, stream=None, max_glyphs=None, canvas_cap=1024, width_cap=600,
                  out_h=32):
    """GPU-side captcha synthesis (csrc/captcha_synth.hip): params int32 [n_images * words_per_image (+ more)] device records of
    utils.synth.draw_params -> out uint8 [n_images, W, out_h].  stream: a torch stream (default: the current one)."""
    assert params.dtype == torch.int32 and atlas.dtype == torch.uint8 and stamp.dtype == torch.int32 and out.dtype == torch.uint8
    assert params.numel() >= n_images * words_per_image and out.numel() >= n_images * W * out_h
    if max_glyphs is None:
        max_glyphs = (words_per_image - 92) // 20
    st = _st() if stream is None else stream.cuda_stream
    call("ocr_captcha_synth", ptr(_dev(params)), n_images, words_per_image, max_glyphs, ptr(_dev(atlas)), ptr(_dev(stamp)), stamp.numel() // 2,
         ptr(_dev(out)), W, out_h, canvas_cap, width_cap, st)
    return out


def step_report(costs, scalars, word_addrs, out):
    """out[0..3] (double) = mean cost, scalars[1], scalars[7], bit mask of the error words that read 1 (+ 2^40: this step's update was dropped) — ocr_step_report."""
    nw = 0 if word_addrs is None else word_addrs.numel()
    if scalars is not None and scalars.numel() < optim_scalar_count():
        raise ValueError('step_report: `scalars` must be the whole optimiser block (%d doubles; the report reads entry 72), got %d'
                         % (optim_scalar_count(), scalars.numel()))
    call("ocr_step_report", ptr(_dev(costs)), costs.numel(), ptr(scalars), ptr(word_addrs) if nw else None, nw, ptr(out), _st())
    return out


def cast2d_bf16(src, ldin, dst, ldout, rows, cols):
    call("ocr_cast2d_f32_bf16", ptr(_dev(src)), ldin, ptr(dst), ldout, rows, cols, _st())
    return dst


def tnc_to_ntc_bf16(src_tnc, dst_ntc, scale):
    T, N, C = src_tnc.shape
    call("ocr_tnc_to_ntc_bf16", ptr(_dev(src_tnc)), ptr(dst_ntc), T, N, C, float(scale), _st())
    return dst_ntc


def conv5_col2im(col, dx, Nb, W, HC):
    call("ocr_conv5_col2im", ptr(_dev(col)), ptr(dx), Nb, W, HC, _st())
    return dx


# ----------------------------------------------------------------------------------------------- LSTM
def lstm_fwd_step(xproj, whT, seq_len, hout, gates, cell, Nb, T, U, step, forget_bias=1.0, ndir=2):
    call("ocr_lstm_fwd_step", ptr(_dev(xproj)), ptr(whT), ptr(seq_len), ptr(hout), ptr(gates), ptr(cell), Nb, T, U, step,
         float(forget_bias), ndir, _st())


def lstm_bwd_step(wh, ldw, w_dir_stride, seq_len, dhout, gates, cell, dz, dc_state, Nb, T, U, step, ndir=2):
    call("ocr_lstm_bwd_step", ptr(_dev(wh)), ldw, w_dir_stride, ptr(seq_len), ptr(dhout), ptr(gates), ptr(cell), ptr(dz),
         ptr(dc_state), Nb, T, U, step, ndir, _st())


def set_lstm_ksplit(waves):
    """Waves per workgroup of the persistent LSTM kernels (4: default since round 4; 1: the one-wave kernels).  OCR_LSTM_KSPLIT wins."""
    call("ocr_set_lstm_ksplit", int(waves))


def lstm_seq_test_skew(units, at=-1):
    """Test hook: unit block 0 of every group of the persistent LSTM launches that follow sleeps units x 64 clocks at iteration `at` (< 0: every
    iteration; units 0: off)."""
    call("ocr_lstm_seq_test_skew", int(units), int(at))


def lstm_seq_supported(Nb, U):
    return bool(nat.lib().ocr_lstm_seq_supported(Nb, U))


def lstm_seq_sync_words(Nb, U):
    """int32 words of the persistent kernels' scratch block (group counters | hand-off ring | tail with the error word LAST)."""
    return int(nat.lib().ocr_lstm_seq_sync_words(Nb, U))


LSTM_PREPARED = 1       # OCR_LSTM_PREPARED: every word of `sync` was set to 0xFFFFFFFF earlier on this stream; the call skips its own fill launch


def lstm_fwd_seq(xproj, whT, seq_len, hout, gates, cell, Nb, T, U, sync, forget_bias=1.0, prepared=False):
    call("ocr_lstm_fwd_seq2", ptr(_dev(xproj)), ptr(whT), ptr(seq_len), ptr(hout),
         ptr(gates), ptr(cell), Nb, T, U, float(forget_bias), ptr(sync), LSTM_PREPARED if prepared else 0, _st())


def lstm_fwd_seq_x_supported(Nb, U, D):
    return bool(nat.lib().ocr_lstm_fwd_seq_x_supported(Nb, U, D))


def lstm_fwd_seq_x(x, wxT, bias, whT, seq_len, hout, gates, cell, Nb, T, U, sync, forget_bias=1.0, prepared=False):
    """The forward recurrence with the input projection inside: x bf16 [Nb * T, D]; wxT [2 * 4U, D] / bias [2 * 4U] as gemm_nt takes them."""
    D = x.shape[-1]
    call("ocr_lstm_fwd_seq_x", ptr(_dev(x)), ptr(wxT), ptr(bias), D, ptr(whT), ptr(seq_len), ptr(hout), ptr(gates), ptr(cell),
         Nb, T, U, float(forget_bias), ptr(sync), LSTM_PREPARED if prepared else 0, _st())


def lstm_bwd_seq(wh, ldw, w_dir_stride, seq_len, dhout, gates, cell, dz, Nb, T, U, sync, prepared=False):
    call("ocr_lstm_bwd_seq2", ptr(_dev(wh)), ldw, w_dir_stride, ptr(seq_len),
         ptr(dhout), ptr(gates), ptr(cell), ptr(dz), Nb, T, U, ptr(sync), LSTM_PREPARED if prepared else 0, _st())


def lstm_hprev(hout, seq_len, hprev, Nb, T, U, ndir=2):
    call("ocr_lstm_hprev", ptr(_dev(hout)), ptr(seq_len), ptr(hprev), Nb, T, U, ndir, _st())


def lstm_pack_bias(b_fw, b_bw, out, U, ndir=2):
    call("ocr_lstm_pack_bias", ptr(_dev(b_fw)), ptr(b_bw), ptr(out), U, ndir, _st())


# ----------------------------------------------------------------------------------------------- optimiser
SOLVERS = {"Adam": 0, "Momentum": 1, "RMS": 2}


def optim_scalar_count():
    return int(nat.lib().ocr_optim_scalar_count())


def optim_init(scalars, lr):
    call("ocr_optim_init", ptr(_dev(scalars)), float(lr), _st())


def optim_set_lr(scalars, lr, multiply=False):
    call("ocr_optim_set_lr", ptr(_dev(scalars)), float(lr), int(multiply), _st())


def optim_step(params, grads, state1, state2, reg_range, weight_decay, clip_norm, solver, beta1, beta2, eps, scalars, guard=None, drop_flag=None):
    """reg_range = (begin, end) of the L2-regularised tensors inside the flat buffer.  guard: device int64 tensor of addresses of int words
    that read 1 when a kernel of the step reported invalid results — the update is then dropped on the device; drop_flag: a device float, > 0 =>
    drop (data parallel: the all-reduced guard_flag word, so every rank drops the same step) — ocr_optim_step_guarded2."""
    call("ocr_optim_step_guarded2", ptr(_dev(params)), ptr(grads), ptr(state1), ptr(state2), params.numel(), int(reg_range[0]), int(reg_range[1]),
         float(weight_decay), float(clip_norm), solver, float(beta1), float(beta2), float(eps), ptr(scalars),
         ptr(guard) if guard is not None else None, 0 if guard is None else int(guard.numel()),
         ptr(drop_flag) if drop_flag is not None else None, _st())


def guard_flag(guard, out):
    """out[0] (device float) := 1.0 if any of the int words whose addresses `guard` holds reads 1, else 0.0 (ocr_guard_flag)."""
    call("ocr_guard_flag", ptr(guard) if guard is not None else None, 0 if guard is None else int(guard.numel()), ptr(_dev(out)), _st())
