"""Plan executor: lowers a Network plan onto the gfx950 kernels of libocrhip.so.

What the reference does with a TF session (`sess.run([loss, train_op], feed_dict)` — lib/lstm/train.py:118-130)
is done here as:
  * ONE flat fp32 parameter buffer in HBM (TF variable names and layouts kept, so checkpoints stay portable),
    ordered [L2-regularised tensors | rest]; flat gradient / Adam-moment buffers of the same layout — the
    optimiser is two grid-stride kernels and the data-parallel exchange is one RCCL all-reduce of one buffer;
  * bf16 operand copies of the weights in the K-contiguous layouts the MFMA kernels want, refreshed by small
    pack kernels after every update;
  * per input shape (N, W) a statically allocated set of activation / gradient buffers and a captured hipGraph
    of the whole forward+backward (and of the optimiser), replayed each step — no allocator, no host scalars,
    no per-op launch cost on the hot loop;
  * layers run in plan order forward and in reverse order backward (the graph is a chain); ReLU backward is
    fused into the consumer's gradient kernel (dgrad epilogue / pool backward / BN backward).
Arithmetic contract: bf16 storage and MFMA operands, fp32 accumulation, fp32 gate / cell / CTC / BN-statistics /
optimiser math.  oracle/graph.py(sim_bf16=True) rounds at the same points.
"""
import math

import os
import time

import numpy as np
import torch

from . import dist as ocr_dist
from . import ops
from ._native import NativeError
from .layout import FlatLayout, execution_order, init_host_parameters
from ._native import call as nat_call

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
BN_EPS = 1e-3            # tf.contrib.layers.batch_norm default (reference network.py:176-178)
ALIGN = 64               # parameter offsets are multiples of 64 elements (256 B)


def _round_up(x, m):
    return (x + m - 1) // m * m


# ====================================================================================================== ops
def _take_ring(sp, key):
    """True once per fill: the hand-off block `key` still holds the all-ones pattern the conv1 + pool launch of THIS forward pass wrote.
    The persistent LSTM launch that asks consumes it — a second launch on the same block (a repeated backward pass without a new
    training forward) gets False and fills the block itself."""
    ready = getattr(sp, 'rings_ready', None)
    if not ready or key not in ready:
        return False
    ready.discard(key)
    return True


class _Op(object):
    mask_in_consumer = False      # True: output has a fused ReLU whose backward the consumer must apply

    def __init__(self, eng, node, prev):
        self.eng, self.node, self.prev, self.name = eng, node, prev, node.name
        self.key = node.name          # buffer key; made unique by Engine._lower (the reference reuses layer names)
        self.inputs = [prev]
        self.consumers = 0            # how many ops read this op's output (residual graphs: > 1)

    def grad_owner(self):             # the op whose dy buffer really receives gradients for this output (views forward it)
        return self

    def out_shape(self, in_shape):
        raise NotImplementedError

    def alloc(self, sp, in_shape):
        pass

    def refresh(self):            # re-pack work that is not expressible as a pack job
        pass

    def pack_jobs(self):          # bf16 operand re-packs after a parameter update (executed by ONE launch)
        return []

    def shadow_params(self):      # variables whose plain bf16 copy (same layout, Engine.shadow) this op reads
        return []

    def fwd(self, sp):
        pass

    def bwd(self, sp):
        pass

    # gradient w.r.t. this op's output (written by the consumer)
    def y(self, sp):
        return sp.buf[self.key + '/y']

    def dy(self, sp):
        return sp.buf[self.key + '/dy']


class _InputOp(_Op):
    def __init__(self, eng):
        self.eng, self.name, self.prev, self.node = eng, 'data', None, None
        self.key = 'data'
        self.inputs, self.consumers = [], 0

    def y(self, sp):
        return sp.x

    def dy(self, sp):
        return None


class _ConvOp(_Op):
    def __init__(self, eng, node, prev):
        super(_ConvOp, self).__init__(eng, node, prev)
        a = node.attrs
        self.kh, self.kw, self.co, self.ci = a['k_h'], a['k_w'], a['c_o'], a['c_i']
        self.bn, self.relu, self.biased, self.padding = a['bn'], a['relu'], a['biased'], a['padding']
        # strides: the op itself always runs at stride 1; Engine._lower appends a _SubsampleOp (a strided convolution is the
        # stride-1 convolution evaluated at every s-th position)
        if self.ci == 1:
            if (self.kh, self.kw, self.padding) != (3, 3, 'SAME') or self.co != 64 or self.bn or not self.biased:
                raise NotImplementedError('%s: the single-channel input conv is lowered for 3x3 SAME, 64 filters' % self.name)
            self.kind = 'c1'
        elif (self.kh, self.kw, self.padding) == (3, 3, 'SAME'):
            if self.ci % 32 or self.co % 8:
                raise NotImplementedError('%s: 3x3 conv needs C_in %% 32 == 0 and C_out %% 8 == 0' % self.name)
            self.kind = '3x3'
        elif (self.kh, self.kw) == (1, 1):
            if self.ci % 8 or self.co % 8:
                raise NotImplementedError('%s: 1x1 conv needs C_in %% 8 == 0 and C_out %% 8 == 0' % self.name)
            self.kind = '1x1'           # a plain GEMM over pixels
        elif self.padding == 'VALID':
            self.kind = 'full'          # kernel spans the whole feature axis -> plain GEMM over overlapping rows
        else:
            raise NotImplementedError('%s: %dx%d %s convolution is not lowered' % (self.name, self.kh, self.kw, self.padding))
        self.mask_in_consumer = self.relu and not self.bn
        dev = eng.device
        K = self.kh * self.kw * self.ci
        if self.kind != 'c1':
            self.wpack = torch.empty((self.co, K), dtype=BF16, device=dev)
            if self.kind == '3x3' and prev.dy_needed():
                self.wdgrad = torch.empty((self.ci, 9 * self.co), dtype=BF16, device=dev)

    def dy_needed(self):
        return True

    def out_shape(self, s):
        N, W, H = s[0], s[1], s[2]
        if self.kind == 'full':
            if self.kw != H:
                raise NotImplementedError('%s: VALID conv is lowered only when k_w equals the feature-axis size '
                                          '(got k_w=%d, H=%d)' % (self.name, self.kw, H))
            return (N, W - self.kh + 1, 1, self.co)
        return (N, W, H, self.co)

    fused_pool = None          # set by Engine._lower: conv1 + ReLU + 2x2 max-pool run as one kernel, no full-res activation
    tail_into = None           # set by Engine._lower: (add op, the add's other input) — this conv's batch-norm apply pass also does the
                               # residual add + relu and writes the add's output (the add's forward is then a no-op)
    mask_from = None           # set by Engine._lower: the fused add + relu that is this (BN, no ReLU) conv's only consumer: its batch-norm
                               # backward reads the add's gradient and applies the add's ReLU mask itself (no masked copy pass)
    bn_pool = None             # set by Engine._lower: the 1 x 2 max-pool that is this batch-norm layer's only consumer (written by the BN apply
                               # pass; its gradient is routed inside the BN backward passes)
    pool_after = None          # set by Engine._lower: the max-pool that follows this 3x3 conv + ReLU; where the shape allows, the conv's
                               # epilogue writes the pooled tensor too (the full-resolution output is still kept for the backward pass)

    def alloc(self, sp, s):
        o = self.out_shape(s)
        dev = self.eng.device
        sp.shape[self.key] = (s, o)
        if self.fused_pool is not None:
            if self.co == 64 and self.eng.defer_w9 and os.environ.get('OCR_CONV1_SLAB', '1') != '0':
                # the backward kernel's per-block partial sums (no atomics); two jobs of the merged slab reduction add them
                sp.buf[self.key + '/slab'] = torch.empty((ops.conv1_pool_bwd_slab_rows(s[0], s[1], s[2]), 640), dtype=F32, device=dev)
            if self.co == 64 and os.environ.get('OCR_CONV1_CODES', '1') != '0':
                # pool routing + ReLU bits saved by the training forward pass (4 bits per pooled output) for the backward pass
                sp.buf[self.key + '/codes'] = torch.empty((s[0] * (s[1] // 2) * (s[2] // 2), 8), dtype=torch.int32, device=dev)
            return
        sp.buf[self.key + '/y'] = torch.empty(o, dtype=BF16, device=dev)
        sp.buf[self.key + '/dy'] = torch.empty(o, dtype=BF16, device=dev)
        if self.bn:
            sp.buf[self.key + '/z'] = torch.empty(o, dtype=BF16, device=dev)
            sp.buf[self.key + '/dz'] = torch.empty(o, dtype=BF16, device=dev)
            sp.buf[self.key + '/mean'] = torch.empty(self.co, dtype=F32, device=dev)
            sp.buf[self.key + '/rstd'] = torch.empty(self.co, dtype=F32, device=dev)
            sp.buf[self.key + '/bnws'] = ops.bn_workspace(o[0] * o[1] * o[2], self.co, dev)
            # batch-norm statistics from the convolution's own epilogue where a plane-layout kernel takes the shape (partial rows into the
            # BN workspace: it must hold them), else a statistics pass over z
            rows = 0
            if self.kind == '3x3' and self.eng.fuse_bn_stats:
                rows = ops.conv3x3_stats_rows(s[0], s[1], s[2], self.ci, self.co, bias=self.biased)
                if rows * 2 * self.co * 4 > sp.buf[self.key + '/bnws'].numel() - 2 * self.co * 8:
                    rows = 0
            sp.bn_stat_rows = getattr(sp, 'bn_stat_rows', {})
            sp.bn_stat_rows[self.key] = rows
        if self.kind == 'full':
            N, W, H, C = s
            sp.buf[self.key + '/col'] = torch.empty((N * o[1], self.kh * H * C), dtype=BF16, device=dev)
        if self.pool_after is not None:
            p = self.pool_after
            sp.fused_pools = getattr(sp, 'fused_pools', set())
            if ops.conv3x3_pool_supported(s[0], s[1], s[2], self.ci, self.co, p.kw_t, p.kh_f):
                sp.fused_pools.add(p.key)
        if self.kind == '3x3' and hasattr(self, 'wdgrad') and self.eng.fuse_bn_stats and os.environ.get('OCR_FUSE_BN_BWD', '1') != '0':
            # the producer is a batch-norm + ReLU convolution feeding only this one: its batch-norm BACKWARD sums (and its ReLU mask) are
            # taken by this layer's data-gradient kernel (conv_k3's write-out), where a plane-layout kernel takes the shape
            p = self.prev
            rows = 0
            if (isinstance(p, _ConvOp) and p.bn and p.relu and p.consumers == 1 and p.tail_into is None and p.mask_from is None
                    and p.bn_pool is None and p.kind != 'c1'):
                rows = ops.conv3x3_bnbwd_rows(s[0], s[1], s[2], self.co, self.ci)
                if rows * 2 * self.ci * 4 > sp.buf[p.key + '/bnws'].numel() - 2 * self.ci * 8:
                    rows = 0
            sp.bn_bwd_rows = getattr(sp, 'bn_bwd_rows', {})
            if isinstance(p, _ConvOp):
                # the producer's backward sums come EITHER from this layer's data-gradient write-out (partial rows) OR from its own passes
                # routing a pooled gradient — never both: ocr_bn_train_bwd2 takes one form per launch
                assert p.bn_pool is None or not rows, (p.key, rows)
                sp.bn_bwd_rows[p.key] = rows
        if self.kind == '3x3':      # scratch of the slab weight-gradient kernel
            need = ops.conv3x3_wgrad_workspace_bytes(s[0], s[1], s[2], self.ci, self.co)
            have = sp.buf.get('wgrad_ws')
            if need and self.eng.defer_w9:
                # one buffer PER LAYER: the slabs stay until the end of the backward pass, where ONE launch reduces all layers'
                # (Engine._flush_w9) — ~20 MB per layer of 288 GB against four dependent launches less per step
                # ... shared by the plans of an engine (plans run one at a time on one stream and no slab outlives its backward body): a
                # variable-width run makes one plan per padded width — a narrower plan takes a view of the widest workspace allocated so
                # far instead of slabs of its own (9-61 plans x ~0.4 GB for the deep net otherwise)
                pool = self.eng.w9_ws_pool
                have_ws = pool.get(self.key)
                if have_ws is None or have_ws.numel() < need:
                    # sized ONCE for the widest batch this engine is expected to see (plans created in ascending width order — the usual
                    # case for variable-width batches — each found the pool too small and allocated a block of their own): the slab bytes of a
                    # layer grow with the split count, which is capped, so the need at OCR_MAX_WIDTH (default 320 = configs[3]'s widest) bounds them
                    wmax = max(int(os.environ.get('OCR_MAX_WIDTH', '320')), sp.W)
                    scale = max(1, wmax // sp.W + (1 if wmax % sp.W else 0))
                    big = need
                    if scale > 1:
                        big = max(need, ops.conv3x3_wgrad_workspace_bytes(s[0], s[1] * scale, s[2], self.ci, self.co))
                    have_ws = pool[self.key] = torch.empty(big, dtype=torch.uint8, device=dev)      # (older plans keep the block their graphs captured)
                sp.buf[self.key + '/w9ws'] = have_ws[:need]
            elif need and (have is None or have.numel() < need):     # one buffer per plan, shared by all layers (one stream)
                sp.buf['wgrad_ws'] = torch.empty(need, dtype=torch.uint8, device=dev)

    def shadow_params(self):
        return [self.name + '/weights'] if self.kind in ('1x1', 'full') else []

    def pack_jobs(self):
        if self.kind == 'c1':
            return []
        w = self.eng.param(self.name + '/weights')
        K = self.kh * self.kw * self.ci
        jobs = [dict(type=0, R=K, Cc=self.co, ldin=self.co, src=w, dst=self.wpack)]
        if hasattr(self, 'wdgrad'):
            jobs.append(dict(type=1, R=self.ci, Cc=self.co, src=w, dst=self.wdgrad, n=w.numel()))
        return jobs

    def fwd(self, sp):
        e = self.eng
        x = self.prev.y(sp)
        s, o = sp.shape[self.key]
        bias = e.param(self.name + '/biases') if self.biased else None
        if self.fused_pool is not None:
            zero, e._zero_pending = (e.grads if e._zero_pending else None), False     # the gradient buffer's clear rides on this launch
            codes = sp.buf.get(self.key + '/codes') if e.training else None
            ones, e._ones_pending = e._ones_pending, None                              # ... and so does the fill of the LSTM hand-off blocks
            ops.conv1_pool_fwd(x, e.param(self.name + '/weights'), bias, out=self.fused_pool.y(sp), zero=zero, codes=codes, ones=ones)
            # every hand-off block of the plan is armed now; each persistent launch consumes (and thereby dirties) its own block ONCE
            sp.rings_ready = set(getattr(sp, 'lstm_sync_keys', ())) if ones is not None else set()
            sp.rings_armed = ones is not None         # (what the last forward pass did; rings_ready is consumed by the launches)
            return
        y = self.y(sp)
        if self.kind == 'c1':
            ops.conv1_fwd(x, e.param(self.name + '/weights'), bias, relu=self.relu, out=y)
            return
        tgt = sp.buf[self.key + '/z'] if self.bn else y
        relu_now = self.relu and not self.bn
        if self.kind == '3x3' and self.pool_after is not None and self.pool_after.key in sp.fused_pools:
            p = self.pool_after
            ops.conv3x3_relu_pool(x, self.wpack.view(self.co, 3, 3, self.ci), y, p.y(sp), bias, p.kw_t, p.kh_f)
        elif self.kind == '3x3' and self.bn and sp.bn_stat_rows[self.key]:
            ops.conv3x3_stats(x, self.wpack.view(self.co, 3, 3, self.ci), tgt, sp.buf[self.key + '/bnws'], bias=bias)
        elif self.kind == '3x3':
            ops.conv3x3(x, self.wpack.view(self.co, 3, 3, self.ci), out=tgt, bias=bias, relu=relu_now)
        elif self.kind == '1x1':
            Mx = s[0] * s[1] * s[2]
            ops.gemm_nt(x.view(Mx, self.ci), self.wpack, out=tgt.view(Mx, self.co), bias=bias, relu=relu_now)
        else:
            N, W, H, C = s
            Wo = o[1]
            ops.gemm_nt(x, self.wpack, out=tgt.view(N * Wo, self.co), M=N * Wo, N=self.co, K=self.kh * H * C,
                        ldp=H * C, ldq=self.kh * H * C, bias=bias, relu=relu_now, row_group=Wo, row_skip=self.kh - 1)
        if self.bn:
            M = o[0] * o[1] * o[2]
            res, relu = None, self.relu
            if self.tail_into is not None:          # y_add = relu(bf16(bn(z)) + other input): written straight into the add's buffer
                add, other = self.tail_into
                y, res, relu = add.y(sp), other.y(sp).view(M, self.co), True
            pooled = self.bn_pool.y(sp).view(M // 2, self.co) if self.bn_pool is not None else None
            ops.bn_train_fwd(tgt.view(M, self.co), e.param('%s/%s/gamma' % (self.name, self.name)),
                             e.param('%s/%s/beta' % (self.name, self.name)), BN_EPS, relu,
                             sp.buf[self.key + '/bnws'], out=y.view(M, self.co),
                             save_mean=sp.buf[self.key + '/mean'], save_rstd=sp.buf[self.key + '/rstd'], residual=res,
                             partial_rows=sp.bn_stat_rows[self.key], pooled=pooled)

    def bwd(self, sp):
        e = self.eng
        if self.fused_pool is not None:     # pool routing + ReLU mask + weight gradient in one recomputing pass
            slab = sp.buf.get(self.key + '/slab')
            if slab is not None:
                ops.conv1_pool_bwd_slab(self.prev.y(sp), e.param(self.name + '/weights'), e.param(self.name + '/biases'), self.fused_pool.dy(sp),
                                        slab, codes=sp.buf.get(self.key + '/codes'))
                S = slab.shape[0]
                for dst, off, n4 in ((e.grad(self.name + '/weights'), 0, 144), (e.grad(self.name + '/biases'), 576, 16)):
                    job = np.zeros(1, dtype=e.W9_JOB_DTYPE)
                    job['dw'], job['part'], job['n4'], job['slab4'] = dst.data_ptr(), slab.data_ptr() + 4 * off, n4, 160
                    job['S'], job['rows'], job['Cout'] = S, 8, 64
                    sp.w9_pending.append((job.tobytes(), (n4 + 31) // 32))          # 256 / rows = 32 float4 columns per block
                return
            ops.conv1_pool_bwd(self.prev.y(sp), e.param(self.name + '/weights'), e.param(self.name + '/biases'),
                               self.fused_pool.dy(sp), e.grad(self.name + '/weights'), e.grad(self.name + '/biases'),
                               codes=sp.buf.get(self.key + '/codes'))
            return
        s, o = sp.shape[self.key]
        x = self.prev.y(sp)
        dy = self.dy(sp)                    # already ReLU-masked by the consumer unless this layer has BN
        M = o[0] * o[1] * o[2]
        dz = dy
        if self.bn:
            dz = sp.buf[self.key + '/dz']
            ymask, relu = self.y(sp), self.relu
            if self.mask_from is not None:          # gradient and ReLU mask straight from the residual add + relu behind this layer
                dy, ymask, relu = self.mask_from.dy(sp), self.mask_from.y(sp), True
            dy2 = dy.view(M, self.co)
            if self.bn_pool is not None:            # the pooled gradient, routed by the passes themselves
                dy2 = self.bn_pool.dy(sp).view(M // 2, self.co)
            ops.bn_train_bwd(sp.buf[self.key + '/z'].view(M, self.co), ymask.view(M, self.co), dy2,
                             e.param('%s/%s/gamma' % (self.name, self.name)), sp.buf[self.key + '/mean'],
                             sp.buf[self.key + '/rstd'], e.grad('%s/%s/gamma' % (self.name, self.name)),
                             e.grad('%s/%s/beta' % (self.name, self.name)), relu, sp.buf[self.key + '/bnws'],
                             out=dz.view(M, self.co), pooled_dy=self.bn_pool is not None,
                             partial_rows=getattr(sp, 'bn_bwd_rows', {}).get(self.key, 0))
        dw = e.grad(self.name + '/weights')
        db = e.grad(self.name + '/biases') if self.biased else None
        if self.kind == 'c1':
            ops.conv1_wgrad(x, dz, dw, db)
            return
        pmask = self.prev.y(sp) if self.prev.mask_in_consumer else None
        if self.kind == '3x3':
            own_ws = sp.buf.get(self.key + '/w9ws')
            if own_ws is not None:      # slab kernel now, the reduction with the other layers' at the end of the pass
                job, nblk = ops.conv3x3_wgrad_deferred(x, dz, dw, db, own_ws)
                if job is not None:
                    sp.w9_pending.append((job, nblk))
                    sp.w9_pending_bytes = getattr(sp, 'w9_pending_bytes', 0) + own_ws.numel()
                    if e.w9_flush_bytes and sp.w9_pending_bytes >= e.w9_flush_bytes:
                        e._flush_w9(sp)         # (knob OCR_W9_FLUSH_MB: reduce while the slabs are still in the Infinity Cache)
            else:
                ops.conv3x3_wgrad(x, dz, dw, dbias=db, workspace=sp.buf.get('wgrad_ws'))   # bias gradient rides on the same pass
            # a producer with several consumers (residual graphs): where the halo kernel runs, its epilogue adds to what was already
            # delivered instead of writing a scratch tensor that a second pass adds
            pdy, finish, acc = e.grad_dst_acc(sp, self.prev, ops.conv3x3_accum_supported(o[0], o[1], o[2], self.co, self.ci))
            if pdy is not None:
                p = self.prev
                if getattr(sp, 'bn_bwd_rows', {}).get(p.key, 0) and not acc:
                    ops.conv3x3_dgrad_bnbwd(dz, self.wdgrad.view(self.ci, 3, 3, self.co), pdy, p.y(sp), sp.buf[p.key + '/z'],
                                            sp.buf[p.key + '/mean'], sp.buf[p.key + '/rstd'], sp.buf[p.key + '/bnws'])
                else:
                    ops.conv3x3(dz, self.wdgrad.view(self.ci, 3, 3, self.co), out=pdy, mask=pmask, accumulate=acc)
                finish()
            return
        pdy, finish = e.grad_dst(sp, self.prev)
        if self.kind == '1x1':
            ops.gemm_tn(x.view(M, self.ci), dz.view(M, self.co), dw.view(self.ci, self.co), colsum=db)
            if pdy is not None:
                wsh = e.shadow(self.name + '/weights').view(self.ci, self.co)      # Q[n = ci][k = co]
                ops.gemm_nt(dz.view(M, self.co), wsh, out=pdy.view(M, self.ci),
                            mask=None if pmask is None else pmask.view(M, self.ci))
                finish()
        else:
            N, W, H, C = s
            Wo = o[1]
            K = self.kh * H * C
            dz2, dw2 = dz.view(M, self.co), dw.view(K, self.co)
            e.tn_push(sp, ops.tn_job(x, H * C, dz2, self.co, dw2, self.co, M, K, self.co, row_group=Wo, row_skip=self.kh - 1, colsum=db),
                      (x, dz2, dw2, db),
                      lambda: ops.gemm_tn(x, dz2, dw2, Mk=M, I=K, J=self.co, lda=H * C, ldb=self.co,
                                          ldo=self.co, row_group=Wo, row_skip=self.kh - 1, colsum=db))
            if pdy is not None:
                if pmask is not None or self.kh != 2:
                    raise NotImplementedError('%s: data gradient of a full-height VALID conv is lowered for k_h = 2 '
                                              'behind a non-ReLU producer' % self.name)
                col = sp.buf[self.key + '/col']
                wsh = e.shadow(self.name + '/weights').view(K, self.co)      # [K][co] bf16, k = co contiguous
                ops.gemm_nt(dz.view(M, self.co), wsh, out=col, M=M, N=K, K=self.co)
                ops.conv5_col2im(col, pdy, N, W, H * C)
                finish()


class _FcOp(_ConvOp):
    """Network.fc (network.py:415-447) on a row tensor [N, T, d]: y = relu?(x W + b) over the last axis = the GEMMs of a 1 x 1 convolution
    (`weights [dim, num_out]` has the memory layout of a [1, 1, dim, num_out] filter), so forward, data gradient, weight gradient, bf16 shadow and
    re-pack are _ConvOp's '1x1' paths on the flattened rows."""

    def __init__(self, eng, node, prev):
        super(_FcOp, self).__init__(eng, node, prev)
        assert self.kind == '1x1'

    def out_shape(self, s):
        return tuple(s[:-1]) + (self.co,)

    def alloc(self, sp, s):
        o = self.out_shape(s)
        rows = int(np.prod(s[:-1]))
        sp.shape[self.key] = ((rows, 1, 1, s[-1]), (rows, 1, 1, self.co))      # what the 1x1 paths multiply out
        sp.buf[self.key + '/y'] = torch.empty(o, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(o, dtype=BF16, device=self.eng.device)


class _PoolOp(_Op):
    def __init__(self, eng, node, prev):
        super(_PoolOp, self).__init__(eng, node, prev)
        a = node.attrs
        # reference argument order is (k_h, k_w, s_h, s_w) with TF "height" = our time axis W
        self.kw_t, self.kh_f = a['k_h'], a['k_w']
        if (a['s_h'], a['s_w']) != (a['k_h'], a['k_w']) or self.kw_t not in (1, 2) or self.kh_f not in (1, 2):
            raise NotImplementedError('%s: max-pool is lowered for window == stride in {1,2}' % self.name)

    def dy_needed(self):
        return True

    def out_shape(self, s):
        N, W, H, C = s
        if W % self.kw_t or H % self.kh_f:
            raise NotImplementedError('%s: pooled axes must divide evenly (W=%d, H=%d)' % (self.name, W, H))
        return (N, W // self.kw_t, H // self.kh_f, C)

    def alloc(self, sp, s):
        o = self.out_shape(s)
        sp.shape[self.key] = (s, o)
        sp.buf[self.key + '/y'] = torch.empty(o, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(o, dtype=BF16, device=self.eng.device)

    fused_into = None          # the conv1 op that computes this pool's output itself
    bn_fused_into = None       # the batch-norm conv whose apply pass writes this pool's output and whose backward passes route its gradient

    def fwd(self, sp):
        if self.bn_fused_into is not None:
            return
        if self.fused_into is None and self.key not in getattr(sp, 'fused_pools', ()):      # else: written by the producing conv's epilogue
            ops.maxpool_fwd(self.prev.y(sp), self.kw_t, self.kh_f, out=self.y(sp))

    def bwd(self, sp):
        if self.fused_into is not None or self.bn_fused_into is not None:
            return
        pdy, finish = self.eng.grad_dst(sp, self.prev)
        if pdy is not None:
            ops.maxpool_bwd(self.prev.y(sp), self.dy(sp), self.kw_t, self.kh_f, self.prev.mask_in_consumer, out=pdy)
            finish()


class _ViewOp(_Op):
    """reshape_squeeze_layer / dropout(identity): no data movement, shares the producer's buffers."""

    def __init__(self, eng, node, prev):
        super(_ViewOp, self).__init__(eng, node, prev)
        self.mask_in_consumer = prev.mask_in_consumer

    def dy_needed(self):
        return True

    def out_shape(self, s):
        if self.node.op == 'reshape_squeeze':
            N, A, B, C = s
            return (N, A * B, C)
        return s

    def alloc(self, sp, s):
        sp.shape[self.key] = (s, self.out_shape(s))

    def y(self, sp):
        return self.prev.y(sp).view(sp.shape[self.key][1])

    def dy(self, sp):
        d = self.prev.dy(sp)
        return None if d is None else d.view(sp.shape[self.key][1])

    def grad_owner(self):
        return self.prev.grad_owner()


class _AddOp(_Op):
    """Network.add (network.py:461-463): element-wise sum of two feature maps (residual connections)."""

    def dy_needed(self):
        return True

    def out_shape(self, s):
        return s

    def alloc(self, sp, s):
        sp.shape[self.key] = (s, s)
        sp.buf[self.key + '/y'] = torch.empty(s, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(s, dtype=BF16, device=self.eng.device)

    relu = False               # True: the Network.relu that follows (its only consumer) is computed here, y = relu(a + b)

    fwd_by = None              # the producing conv whose batch-norm apply pass writes this op's output (Engine._lower)

    def fwd(self, sp):
        if self.fwd_by is None:
            ops.eltwise(3 if self.relu else 0, self.inputs[0].y(sp), self.inputs[1].y(sp), self.y(sp))

    def bwd(self, sp):
        for p in self.inputs:
            if getattr(p, 'mask_from', None) is self:
                continue                            # that producer's batch-norm backward reads dy / y of this op directly
            self.eng.deliver(sp, p, self.dy(sp), mask=self.y(sp) if self.relu else None)


class _ReluOp(_Op):
    """Network.relu (network.py:340-341) as a stand-alone layer (after a residual add)."""

    def dy_needed(self):
        return True

    def out_shape(self, s):
        return s

    fused_into = None          # the _AddOp that computes relu(a + b) itself: this op is then a view of it (no buffers, no launches)

    def alloc(self, sp, s):
        sp.shape[self.key] = (s, s)
        if self.fused_into is None:
            sp.buf[self.key + '/y'] = torch.empty(s, dtype=BF16, device=self.eng.device)
            sp.buf[self.key + '/dy'] = torch.empty(s, dtype=BF16, device=self.eng.device)

    def y(self, sp):
        return self.fused_into.y(sp) if self.fused_into is not None else sp.buf[self.key + '/y']

    def dy(self, sp):
        return self.fused_into.dy(sp) if self.fused_into is not None else sp.buf[self.key + '/dy']

    def grad_owner(self):
        return self.fused_into.grad_owner() if self.fused_into is not None else self

    def fwd(self, sp):
        if self.fused_into is None:
            ops.eltwise(1, self.prev.y(sp), None, self.y(sp))

    def bwd(self, sp):
        if self.fused_into is None:
            self.eng.deliver(sp, self.prev, self.dy(sp), mask=self.y(sp))


class _SubsampleOp(_Op):
    """Second half of a strided convolution (Network.conv / conv_single with s_h, s_w > 1, network.py:193-216): picks every
    s-th position of the stride-1 result.  TF SAME: out = ceil(in / s), pad_before = max((out-1) s + k - in, 0) // 2, so output
    o is the stride-1 output at o s + (k-1)//2 - pad_before."""

    def __init__(self, eng, node, prev):
        super(_SubsampleOp, self).__init__(eng, node, prev)
        a = node.attrs
        self.sw, self.sh, self.kw, self.kh = a['s_h'], a['s_w'], a['k_h'], a['k_w']      # reference "height" = our time axis W
        if a['padding'] != 'SAME':
            raise NotImplementedError('%s: strided convolution is lowered for SAME padding' % self.name)

    def dy_needed(self):
        return True

    @staticmethod
    def _geom(n, k, s):
        out = -(-n // s)
        before = max((out - 1) * s + k - n, 0) // 2
        return out, (k - 1) // 2 - before

    def out_shape(self, s):
        N, W, H, C = s
        Wo, self.ow = self._geom(W, self.kw, self.sw)
        Ho, self.oh = self._geom(H, self.kh, self.sh)
        return (N, Wo, Ho, C)

    def alloc(self, sp, s):
        o = self.out_shape(s)
        sp.shape[self.key] = (s, o)
        sp.buf[self.key + '/y'] = torch.empty(o, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(o, dtype=BF16, device=self.eng.device)

    def fwd(self, sp):
        (N, W, H, C), (_, Wo, Ho, _) = sp.shape[self.key]
        ops.subsample(self.prev.y(sp), self.y(sp), N, W, H, C, Wo, Ho, self.sw, self.sh, self.ow, self.oh)

    def bwd(self, sp):
        (N, W, H, C), (_, Wo, Ho, _) = sp.shape[self.key]
        pdy, finish = self.eng.grad_dst(sp, self.prev)
        if pdy is not None:
            ops.subsample(self.dy(sp), pdy, N, W, H, C, Wo, Ho, self.sw, self.sh, self.ow, self.oh, backward=True)
            if self.prev.mask_in_consumer:
                ops.eltwise(2, pdy, self.prev.y(sp), pdy)
            finish()


class _BatchNormOp(_Op):
    """Network.batch_normalization (network.py:466-473): batch statistics when is_training, else the stored moving ones."""

    def __init__(self, eng, node, prev):
        super(_BatchNormOp, self).__init__(eng, node, prev)
        self.relu, self.training = node.attrs['relu'], node.attrs['is_training']

    def dy_needed(self):
        return True

    def out_shape(self, s):
        return s

    def alloc(self, sp, s):
        dev, C = self.eng.device, s[-1]
        sp.shape[self.key] = (s, s)
        sp.buf[self.key + '/y'] = torch.empty(s, dtype=BF16, device=dev)
        sp.buf[self.key + '/dy'] = torch.empty(s, dtype=BF16, device=dev)
        sp.buf[self.key + '/dx'] = torch.empty(s, dtype=BF16, device=dev)
        if self.training:
            M = int(np.prod(s[:-1]))
            sp.buf[self.key + '/mean'] = torch.empty(C, dtype=F32, device=dev)
            sp.buf[self.key + '/rstd'] = torch.empty(C, dtype=F32, device=dev)
            sp.buf[self.key + '/bnws'] = ops.bn_workspace(M, C, dev)

    def _x(self, sp):
        if self.prev.mask_in_consumer:
            raise NotImplementedError('%s: batch_normalization behind a fused-ReLU convolution is not lowered' % self.name)
        return self.prev.y(sp)

    def fwd(self, sp):
        e, (s, _) = self.eng, sp.shape[self.key]
        C = s[-1]
        M = int(np.prod(s[:-1]))
        x, y = self._x(sp).view(M, C), self.y(sp).view(M, C)
        if self.training:
            ops.bn_train_fwd(x, e.param(self.name + '/gamma'), e.param(self.name + '/beta'), BN_EPS, self.relu, sp.buf[self.key + '/bnws'],
                             out=y, save_mean=sp.buf[self.key + '/mean'], save_rstd=sp.buf[self.key + '/rstd'])
        else:
            ops.bn_infer_fwd(x, e.param(self.name + '/gamma'), e.param(self.name + '/beta'), e.param(self.name + '/moving_mean'),
                             e.param(self.name + '/moving_variance'), BN_EPS, self.relu, y)

    def bwd(self, sp):
        e, (s, _) = self.eng, sp.shape[self.key]
        C = s[-1]
        M = int(np.prod(s[:-1]))
        x, y, dy, dx = self._x(sp).view(M, C), self.y(sp).view(M, C), self.dy(sp).view(M, C), sp.buf[self.key + '/dx'].view(M, C)
        if self.training:
            ops.bn_train_bwd(x, y, dy, e.param(self.name + '/gamma'), sp.buf[self.key + '/mean'], sp.buf[self.key + '/rstd'],
                             e.grad(self.name + '/gamma'), e.grad(self.name + '/beta'), self.relu, sp.buf[self.key + '/bnws'], out=dx)
        else:
            ops.bn_infer_bwd(x, y, dy, e.param(self.name + '/gamma'), e.param(self.name + '/moving_mean'),
                             e.param(self.name + '/moving_variance'), e.grad(self.name + '/gamma'), e.grad(self.name + '/beta'), BN_EPS,
                             self.relu, dx)
        e.deliver(sp, self.prev, sp.buf[self.key + '/dx'])


class _DropoutOp(_Op):
    """Network.dropout = tf.nn.dropout(x, keep_prob) (network.py:626-628).  keep_prob is the layer's own argument: the network's
    keep_prob input slot (what the shipped graphs pass) takes the value the driver feeds — 0.5 in training steps (train.py:126), 1.0
    otherwise (train.py:156, test.py:75): Engine.train_keep_prob / inference = 1 — and a NUMBER written in the graph is a constant of
    the graph, applied in training and inference alike, as tf.nn.dropout does with a Python float (it used to be ignored).
    The mask is a hash of (layer name, rank, completed optimiser steps, element index): data-parallel ranks draw different masks."""

    def __init__(self, eng, node, prev):
        super(_DropoutOp, self).__init__(eng, node, prev)
        self.mask_in_consumer = False
        import zlib
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank(eng.group)
        self.seed = (zlib.crc32(self.name.encode()) ^ 0x5bd1e995 ^ (rank * 0x9E3779B1)) & 0xffffffff
        kp = node.attrs.get('keep_prob')
        self.const_keep_prob = float(kp) if isinstance(kp, (int, float)) and not isinstance(kp, bool) else None

    def dy_needed(self):
        return True

    def out_shape(self, s):
        return s

    def alloc(self, sp, s):
        sp.shape[self.key] = (s, s)
        sp.buf[self.key + '/y'] = torch.empty(s, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(s, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dx'] = torch.empty(s, dtype=BF16, device=self.eng.device)

    def _kp(self):
        if self.const_keep_prob is not None:
            return self.const_keep_prob
        return float(self.eng.train_keep_prob) if self.eng.training else 1.0

    def fwd(self, sp):
        x = self.prev.y(sp)
        if self.prev.mask_in_consumer:
            raise NotImplementedError('%s: dropout directly behind a fused-ReLU convolution is not lowered' % self.name)
        ops.dropout(x, self.y(sp), self.seed, self.eng.step_counter(), self._kp())

    def bwd(self, sp):
        ops.dropout(self.dy(sp), sp.buf[self.key + '/dx'], self.seed, self.eng.step_counter(), self._kp())
        self.eng.deliver(sp, self.prev, sp.buf[self.key + '/dx'])


class _AvgPoolOp(_PoolOp):
    """Network.avg_pool (network.py:352-359), window == stride."""

    def fwd(self, sp):
        (N, W, H, C), _ = sp.shape[self.key]
        x = self.prev.y(sp)
        if self.prev.mask_in_consumer:
            pass            # the ReLU is already applied to the stored activation; only its BACKWARD is deferred to the consumer
        ops.avgpool(x, self.y(sp), N, W, H, C, self.kw_t, self.kh_f)

    def bwd(self, sp):
        (N, W, H, C), _ = sp.shape[self.key]
        pdy, finish = self.eng.grad_dst(sp, self.prev)
        if pdy is not None:
            ops.avgpool(self.dy(sp), pdy, N, W, H, C, self.kw_t, self.kh_f, backward=True)
            if self.prev.mask_in_consumer:
                ops.eltwise(2, pdy, self.prev.y(sp), pdy)
            finish()


class _ConcatOp(_Op):
    """Network.concat along the channel axis (network.py:154-158, axis = 3): copies, no arithmetic."""

    def __init__(self, eng, node, prev):
        super(_ConcatOp, self).__init__(eng, node, prev)
        if node.attrs.get('axis') not in (3, -1):
            raise NotImplementedError('%s: concat is lowered along the channel axis only' % self.name)

    def dy_needed(self):
        return True

    def out_shape(self, s):
        return tuple(s[:-1]) + (sum(self._cin),)

    def alloc(self, sp, s):
        self._cin = [sp.oshape[p.key][-1] for p in self.inputs]
        o = self.out_shape(s)
        sp.shape[self.key] = (s, o)
        sp.buf[self.key + '/y'] = torch.empty(o, dtype=BF16, device=self.eng.device)
        sp.buf[self.key + '/dy'] = torch.empty(o, dtype=BF16, device=self.eng.device)
        for i, c in enumerate(self._cin):
            sp.buf['%s/part%d' % (self.key, i)] = torch.empty(tuple(s[:-1]) + (c,), dtype=BF16, device=self.eng.device)

    def fwd(self, sp):
        y, off = self.y(sp), 0
        for p, c in zip(self.inputs, self._cin):
            if p.mask_in_consumer:
                raise NotImplementedError('%s: concat of a fused-ReLU convolution output is not lowered' % self.name)
            y[..., off:off + c].copy_(p.y(sp))
            off += c

    def bwd(self, sp):
        dy, off = self.dy(sp), 0
        for i, (p, c) in enumerate(zip(self.inputs, self._cin)):
            part = sp.buf['%s/part%d' % (self.key, i)]
            part.copy_(dy[..., off:off + c])
            self.eng.deliver(sp, p, part)
            off += c


class _SoftmaxOp(_Op):
    """Network.softmax (network.py:441-447) over the last axis of fp32 logits — an inference output (CTC takes unnormalised
    activations, so nothing differentiates through it in this engine)."""

    def out_shape(self, s):
        return s

    def alloc(self, sp, s):
        sp.shape[self.key] = (s, s)
        sp.buf[self.key + '/y'] = torch.empty(s, dtype=F32, device=self.eng.device)

    def dy(self, sp):
        return None

    def fwd(self, sp):
        ops.softmax(self.prev.y(sp), self.y(sp))

    def bwd(self, sp):
        raise NotImplementedError('%s: softmax is an inference output; the loss takes the unnormalised logits' % self.name)


class _BiLstmOp(_Op):
    """bi_lstm (network.py:97-129: ND = 2 directions of LSTMCell(num_hids // 2)) and one cell of the unidirectional stacked `lstm`
    (network.py:130-152: ND = 1, LSTMCell(num_hids)), optionally followed by the FC + [N,T] -> [T,N] transpose."""

    def __init__(self, eng, node, prev):
        super(_BiLstmOp, self).__init__(eng, node, prev)
        a = node.attrs
        self.ND = 2 if node.op == 'bi_lstm' else 1
        self.U, self.C, self.D = a['num_hids'] // self.ND, a['nclasses'], a['din']
        self.cells = a.get('cells') or [self.name + '/fw', self.name + '/bw']       # variable-name prefix of each direction's LSTMCell
        self.fc = a.get('fc', self.name)                                            # variable-name prefix of the FC
        self.with_fc = a.get('with_fc', True)       # False: a hidden layer of a stack, output = [N, T, ND * U]
        if self.U % 32 or self.D % 32 or self.C % 8:
            raise NotImplementedError('%s: needs hidden units per direction %% 32 == 0, input features %% 32 == 0, classes %% 8 == 0' % self.name)
        dev, U, D, C, ND = eng.device, self.U, self.D, self.C, self.ND
        self.wxT = torch.empty((ND * 4 * U, D), dtype=BF16, device=dev)
        self.whT = torch.empty((ND, 4 * U, U), dtype=BF16, device=dev)
        self.bias = torch.empty(ND * 4 * U, dtype=F32, device=dev)
        self.wfcT = torch.empty((C, ND * U), dtype=BF16, device=dev) if self.with_fc else None
        self.wcat = torch.empty((D, ND * 4 * U), dtype=BF16, device=dev)

    def dy_needed(self):
        return True

    def out_shape(self, s):
        N, T, D = s
        return (T, N, self.C) if self.with_fc else (N, T, self.ND * self.U)

    def y(self, sp):
        if self.with_fc:
            return sp.buf[self.key + '/y']
        (N, T, _), _ = sp.shape[self.key]
        return sp.buf[self.key + '/hout'].view(N, T, self.ND * self.U)

    def dy(self, sp):
        if self.with_fc:
            return sp.buf[self.key + '/dy']
        (N, T, _), _ = sp.shape[self.key]
        return sp.buf[self.key + '/dhout'].view(N, T, self.ND * self.U)

    def _persistent(self, N):
        return self.ND == 2 and self.eng.persistent_lstm and ops.lstm_seq_supported(N, self.U)

    def alloc(self, sp, s):
        N, T, D = s
        U, C, ND, dev = self.U, self.C, self.ND, self.eng.device
        R = N * T
        sp.shape[self.key] = (s, self.out_shape(s))
        b = sp.buf
        b[self.key + '/xproj'] = torch.empty((R, ND * 4 * U), dtype=F32, device=dev)
        b[self.key + '/hout'] = torch.zeros((R, ND * U), dtype=BF16, device=dev)
        b[self.key + '/gates'] = torch.zeros((ND, R, 4 * U), dtype=F32, device=dev)
        b[self.key + '/cell'] = torch.zeros((ND, R, U), dtype=F32, device=dev)
        if self.with_fc:
            b[self.key + '/y'] = torch.empty((T, N, C), dtype=F32, device=dev)      # logits, time-major
            b[self.key + '/dy'] = torch.empty((R, C), dtype=BF16, device=dev)       # d loss / d logits, [N,T,C]
        b[self.key + '/dhout'] = torch.empty((R, ND * U), dtype=BF16, device=dev)
        b[self.key + '/dz'] = torch.zeros((R, ND * 4 * U), dtype=BF16, device=dev)
        b[self.key + '/dc'] = torch.zeros((ND, N, U), dtype=F32, device=dev)
        b[self.key + '/hprev'] = torch.empty((ND, R, U), dtype=BF16, device=dev)
        b[self.key + '/xh'] = torch.empty((ND, R, self.D + U), dtype=BF16, device=dev)
        if self.ND == 2:
            words = max(ops.lstm_seq_sync_words(N, U), 64) if self._persistent(N) else 64
            words = (words + 3) // 4 * 4                   # 16-byte granules: the blocks of a plan become views of ONE arena (ShapePlan)
            b[self.key + '/sync_f'] = torch.zeros(words, dtype=I32, device=dev)
            b[self.key + '/sync_b'] = torch.zeros(words, dtype=I32, device=dev)
            sp.lstm_sync_keys = getattr(sp, 'lstm_sync_keys', ()) + (self.key + '/sync_f', self.key + '/sync_b')

    def shadow_params(self):
        return [c + '/weights' for c in self.cells] + ([self.fc + '/weights'] if self.with_fc else [])

    def pack_jobs(self):
        e, U, D, ND = self.eng, self.U, self.D, self.ND
        jobs = []
        for d, cell in enumerate(self.cells):
            w = e.param(cell + '/weights')                  # [D+U, 4U], gate-major columns
            jobs.append(dict(type=0, R=D, Cc=4 * U, ldin=4 * U, lstm_units=U, src=w[:D], dst=self.wxT[d * 4 * U:(d + 1) * 4 * U]))
            jobs.append(dict(type=0, R=U, Cc=4 * U, ldin=4 * U, lstm_units=U, src=w[D:], dst=self.whT[d]))
            jobs.append(dict(type=2, R=D, Cc=4 * U, ldin=4 * U, ldout=ND * 4 * U, src=w, dst=self.wcat[:, d * 4 * U:]))
        if self.with_fc:
            wf = e.param(self.fc + '/weights')
            jobs.append(dict(type=0, R=wf.shape[0], Cc=wf.shape[1], ldin=wf.shape[1], src=wf, dst=self.wfcT))
        for d, cell in enumerate(self.cells):      # the biases' gate-order permutation rides in the same launch
            jobs.append(dict(type=4, lstm_units=U, n=4 * U, src=e.param(cell + '/biases'), dst=self.bias[d * 4 * U:(d + 1) * 4 * U]))
        return jobs

    def fwd(self, sp):
        e, U, C, ND = self.eng, self.U, self.C, self.ND
        (N, T, D), _ = sp.shape[self.key]
        R = N * T
        b = sp.buf
        x = self.prev.y(sp).view(R, D)
        if (self._persistent(N) and ND == 2 and os.environ.get('OCR_LSTM_FUSE_X', '1') != '0' and x.is_contiguous()
                and ops.lstm_fwd_seq_x_supported(N, U, D)):
            # the input projection inside the recurrent kernel (its MFMAs run while a step's first poll is under way): no projection GEMM,
            # no fp32 projection tensor
            ops.lstm_fwd_seq_x(x, self.wxT, self.bias, self.whT, sp.seq_len, b[self.key + '/hout'], b[self.key + '/gates'],
                               b[self.key + '/cell'], N, T, U, b[self.key + '/sync_f'], 1.0, prepared=_take_ring(sp, self.key + '/sync_f'))
        elif self._persistent(N):
            ops.gemm_nt(x, self.wxT, out=b[self.key + '/xproj'], bias=self.bias)
            ops.lstm_fwd_seq(b[self.key + '/xproj'], self.whT, sp.seq_len, b[self.key + '/hout'], b[self.key + '/gates'],
                             b[self.key + '/cell'], N, T, U, b[self.key + '/sync_f'], 1.0, prepared=_take_ring(sp, self.key + '/sync_f'))
        else:
            ops.gemm_nt(x, self.wxT, out=b[self.key + '/xproj'], bias=self.bias)
            for s in range(T):
                ops.lstm_fwd_step(b[self.key + '/xproj'], self.whT, sp.seq_len, b[self.key + '/hout'],
                                  b[self.key + '/gates'], b[self.key + '/cell'], N, T, U, s, 1.0, ND)
        if self.with_fc:
            ops.gemm_nt(b[self.key + '/hout'], self.wfcT, out=b[self.key + '/y'].view(R, C),
                        bias=e.param(self.fc + '/biases'), rowswap=(T, N))

    def bwd(self, sp):
        e, U, C, D, ND = self.eng, self.U, self.C, self.D, self.ND
        (N, T, _), _ = sp.shape[self.key]
        R = N * T
        b = sp.buf
        hout = b[self.key + '/hout']
        x = self.prev.y(sp).view(R, D)
        batched = (D + U) % 128 == 0 and (4 * U) % 128 == 0
        if self.with_fc:
            dl = b[self.key + '/dy']
            # FC: dW += H^T dL, db += colsum dL, dH = dL Wfc^T
            ops.gemm_tn(hout, dl, e.grad(self.fc + '/weights'), colsum=e.grad(self.fc + '/biases'))
            ops.gemm_nt(dl, e.shadow(self.fc + '/weights'), out=b[self.key + '/dhout'])
        # BPTT, all directions per launch
        wsh = e.shadow(self.cells[0] + '/weights')
        stride = e.offset(self.cells[1] + '/weights') - e.offset(self.cells[0] + '/weights') if ND == 2 else 0
        if self._persistent(N):
            ops.lstm_bwd_seq(wsh[D:], 4 * U, stride, sp.seq_len, b[self.key + '/dhout'], b[self.key + '/gates'],
                             b[self.key + '/cell'], b[self.key + '/dz'], N, T, U, b[self.key + '/sync_b'], prepared=_take_ring(sp, self.key + '/sync_b'))
        else:
            b[self.key + '/dc'].zero_()
            for s in range(T - 1, -1, -1):
                ops.lstm_bwd_step(wsh[D:], 4 * U, stride, sp.seq_len, b[self.key + '/dhout'], b[self.key + '/gates'],
                                  b[self.key + '/cell'], b[self.key + '/dz'], b[self.key + '/dc'], N, T, U, s, ND)
        dz = b[self.key + '/dz']
        gw = [e.grad(cell + '/weights') for cell in self.cells]
        gb = [e.grad(cell + '/biases') for cell in self.cells]
        if batched:
            # dW_d[D+U, 4U] = [x | h_prev,d]^T dz_d for all directions in ONE launch (the TF LSTMCell matrix is applied to
            # concat([x_t, h_{t-1}]), network.py:104-107): 4 short-K weight-gradient launches become 1
            xh = b[self.key + '/xh']
            ops.lstm_xh(x, hout, sp.seq_len, xh, N, T, D, U, ND)
            scs = (e.offset(self.cells[1] + '/biases') - e.offset(self.cells[0] + '/biases')) if ND == 2 else 0
            e.tn_push(sp, ops.tn_job(xh, D + U, dz, ND * 4 * U, gw[0], 4 * U, R, D + U, 4 * U, nbatch=ND, strideA=R * (D + U), strideB=4 * U,
                                     strideOut=stride, colsum=gb[0], strideColsum=scs),
                      (xh, dz, gw[0], gb[0]),
                      lambda: ops.gemm_tn_batched(xh, D + U, R * (D + U), dz, ND * 4 * U, 4 * U, gw[0], 4 * U, stride, R, D + U, 4 * U, ND,
                                                  colsum=gb[0], strideColsum=scs))
        else:
            ops.lstm_hprev(hout, sp.seq_len, b[self.key + '/hprev'], N, T, U, ND)
            for d in range(ND):
                dzd = dz[:, d * 4 * U:(d + 1) * 4 * U]
                ops.gemm_tn(x, dzd, gw[d][:D], Mk=R, I=D, J=4 * U, lda=D, ldb=ND * 4 * U, ldo=4 * U, colsum=gb[d])
                ops.gemm_tn(b[self.key + '/hprev'][d], dzd, gw[d][D:], Mk=R, I=U, J=4 * U, lda=U, ldb=ND * 4 * U, ldo=4 * U)
        pdy, finish = e.grad_dst(sp, self.prev)
        if pdy is not None:
            pmask = self.prev.y(sp).view(R, D) if self.prev.mask_in_consumer else None
            ops.gemm_nt(dz, self.wcat, out=pdy.view(R, D), mask=pmask)
            finish()


# ====================================================================================================== plan per shape
class ShapePlan(object):
    """All buffers for one (N, W) input shape plus the captured graphs."""

    def __init__(self, eng, N, W):
        self.N, self.W = N, W
        self.buf, self.shape = {}, {}
        dev = eng.device
        self.x = torch.zeros((N, W, eng.num_features), dtype=F32, device=dev)
        self.labels = torch.zeros(N * eng.max_label_len, dtype=I32, device=dev)
        self.labels_len = torch.zeros(N, dtype=I32, device=dev)
        self.seq_len = torch.ones(N, dtype=I32, device=dev)
        self.oshape = {'data': (N, W, eng.num_features)}       # output shape per op key (multi-input ops look their inputs up here)
        self.scratch = {}
        self.dy_done = set()
        self.w9_pending, self.w9_tables = [], {}    # deferred weight-gradient reductions of the running backward pass / their device tables
        self.tn_pending = []                        # plain weight-gradient products waiting for a partner (Engine.tn_push)
        for op in eng.ops:
            s = self.oshape[op.prev.key]
            op.alloc(self, s)
            self.oshape[op.key] = op.out_shape(s)
        # the hand-off blocks of all persistent LSTM launches of a step as views of ONE arena: a training step sets the whole arena to the
        # pattern the kernels start from inside its first kernel (conv1 + pool forward) instead of one fill launch per LSTM launch
        keys = getattr(self, 'lstm_sync_keys', ())
        self.lstm_arena = None
        if keys:
            self.lstm_arena = torch.zeros(sum(self.buf[k].numel() for k in keys), dtype=I32, device=dev)
            off = 0
            for k in keys:
                n = self.buf[k].numel()
                self.buf[k] = self.lstm_arena[off:off + n]
                off += n
        self.lstm_sync = tuple(self.buf[k] for k in keys)
        # deferred weight-gradient reductions keep every layer's slabs until the end of the backward body: that pays while the slabs
        # still sit in the Infinity Cache when the merged reduction reads them (162 MB for the CRNN at 64 x 256: 18 + 4 x 36 MB).  A deep graph (configs[4]:
        # 32 layers, ~0.4 GB of slabs) would read them back from HBM — measured 3 % slower than reducing behind each layer — so
        # such a plan falls back to ONE shared workspace and immediate reductions.
        # (round 4: with a flush threshold — OCR_W9_FLUSH_MB, default 80 — no more than that plus one layer of slabs is ever pending, so a deep
        #  plan keeps per-layer workspaces too and reduces ~80 MB at a time in merged launches instead of behind each of its 32 layers)
        own = [k for k in self.buf if k.endswith('/w9ws')]
        if own and sum(self.buf[k].numel() for k in own) > eng.w9_defer_max_bytes and not eng.w9_flush_bytes:
            need = max(self.buf[k].numel() for k in own)
            for k in own:
                del self.buf[k]
            self.buf['wgrad_ws'] = torch.empty(need, dtype=torch.uint8, device=dev)
        self.T, _, self.C = self.oshape[eng.ops[-1].key]
        self.costs = torch.zeros(N, dtype=F32, device=dev)
        self.ctc_grad = torch.empty((self.T, N, self.C), dtype=F32, device=dev)
        self.ctc_ws = torch.empty(ops.ctc_workspace_bytes(eng.max_label_len, self.T, N), dtype=torch.uint8, device=dev)
        self.decoded = torch.zeros((N, self.T), dtype=I32, device=dev)
        self.decoded_len = torch.zeros(N, dtype=I32, device=dev)
        self.graph_fb = None
        self.graph_fwd = None


# ====================================================================================================== engine
class Engine(object):
    """Owns parameters, optimiser state and per-shape plans for one Network on one GPU."""

    def __init__(self, net, device='cuda:0', seed=None, max_label_len=31, use_graphs=True, group=None,
                 persistent_lstm=True, fuse_conv1_pool=True):
        from .config import cfg
        if not torch.cuda.is_available():
            raise NativeError('Engine needs a ROCm GPU: the hot path has no CPU implementation')
        self.cfg = cfg
        self.net = net
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.num_features = cfg.NUM_FEATURES
        self.max_label_len = max_label_len
        self.use_graphs = use_graphs
        self.persistent_lstm = persistent_lstm
        self.fuse_conv1_pool = fuse_conv1_pool
        self.fuse_bn_stats = os.environ.get('OCR_FUSE_BN_STATS', '1') != '0'      # batch-norm statistics from the producing convolution's epilogue
        self.w9_flush_bytes = int(float(os.environ.get('OCR_W9_FLUSH_MB', '80')) * (1 << 20))    # slab bytes after which the pending reductions run (0: one at the end of the body); 80 MB: profiles/r04e
        self.tn_defer = os.environ.get('OCR_TN_JOBS', '1') != '0'                 # pairs of plain weight-gradient products as one gemm_tn3 launch
        self.tn_side = os.environ.get('OCR_TN_SIDE', '0') != '0'                  # ... on a side stream beside the main chain's short kernels
        self.side_stream = torch.cuda.Stream(device=self.device) if (self.tn_side and torch.device(self.device).type == 'cuda') else None
        self.group = group
        self.world = 1
        self._check_xcd_placement()
        self.force_allreduce = bool(os.environ.get('OCR_FORCE_ALLREDUCE'))   # exercise the RCCL call on a 1-rank group (tests)
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
        if os.environ.get('OCR_FAKE_WORLD'):            # single-GPU emulation of identical ranks (tests): see dist.allreduce_sum_
            self.world = int(os.environ['OCR_FAKE_WORLD'])
            self.force_allreduce = True
        self.overlap_allreduce = os.environ.get('OCR_OVERLAP_ALLREDUCE', '1') != '0'
        # OCR_DP_GRAPH=1 (round 6, opt-in): the WHOLE data-parallel step as ONE hipGraph — the two bucket all-reduces are captured on the communication
        # stream (fork behind the late backward, join in front of the optimiser) instead of being issued by Python between three graphs: no graph
        # boundaries, no host enqueue between them.  Opt-in because a captured RCCL collective cannot be validated on more than one rank here; the
        # engine checks at start-up that a captured collective replays correctly on THIS group and falls back to the three-graph schedule otherwise.
        self.dp_graph = os.environ.get('OCR_DP_GRAPH', '0') == '1'
        # OCR_W9_DEFER=0: every 3x3 weight gradient reduces its slabs right behind its own kernel (five reduce launches per step
        # for the CRNN); default: one merged reduction per backward body (bit-identical sums)
        self.defer_w9 = os.environ.get('OCR_W9_DEFER', '1') != '0'
        self.w9_defer_max_bytes = int(os.environ.get('OCR_W9_DEFER_MAX_MB', self.W9_DEFER_MAX_BYTES >> 20)) << 20     # only consulted with OCR_W9_FLUSH_MB=0
        self.w9_ws_pool = {}          # layer key -> the largest slab workspace any plan asked for so far (shared by the plans)
        self.comm_stream = torch.cuda.Stream(device=self.device)
        self.dp_host_s = [0.0, 0.0, 0.0, 0.0, 0.0, 0]      # host enqueue seconds of the data-parallel schedule's phases + step count (train_step)
        self._lower(net)                                 # operators first: the parameter layout follows their EXECUTION order
        self._layout(net)
        self._init_params(cfg.RNG_SEED if seed is None else seed)
        self.split_op = 0                                # first op of the late layers (backward part 1 = ops[split_op:])
        if self.split_layer is not None:
            for i, op in enumerate(self.ops):
                if op.name == self.split_layer:
                    self.split_op = i
                    break
            # the exchange of [late_begin, n_total) starts after backward part 1 = ops[split_op:]: no operator of part 2 may own
            # a variable in that range (it would be written while RCCL reads it), none of part 1 a variable below it
            for i, op in enumerate(self.ops):
                for name, owner in self._owner.items():
                    if owner == op.name:
                        assert (self.offsets[name] >= self.late_begin) == (i >= self.split_op), (op.name, name)
        self.plans = {}
        if self.dp_graph and (self.world > 1 or self.force_allreduce):
            self.dp_graph = self._check_graph_collectives()
        self.training = False                            # set by the run bodies: dropout keeps everything outside training steps
        self.train_keep_prob = 0.5                       # what the reference feeds in training steps (train.py:126)
        # optimiser scalars live from the start (all zero until setup_optimizer): graphs captured before the optimiser exists read the
        # step counter — the dropout masks' salt — through the same pointer afterwards (they kept a dummy zero for ever)
        self.scalars = torch.zeros(ops.optim_scalar_count(), dtype=torch.float64, device=self.device)
        self.opt_ready = False
        self.graph_opt = None
        self.iteration = 0
        self.refresh_weights()
        staged = getattr(net, '_staged_arrays', None)
        if staged:
            self.load_arrays(staged)

    # ------------------------------------------------------------------ parameters
    _xcd_checked = False

    def _check_xcd_placement(self):
        """The persistent LSTM's default hand-off keeps a workgroup group inside one XCD's L2 and relies on the dispatcher
        placing workgroups with equal (id & 7) on the same XCD.  Verified once per process on the device itself; if it does
        not hold (another partition mode, a future driver) the kernels are switched to the placement-independent protocol."""
        if Engine._xcd_checked:
            return
        Engine._xcd_checked = True
        n = 256
        out = torch.zeros(2 * n, dtype=torch.int32, device=self.device)
        nat_call('ocr_probe_xcc', out.data_ptr(), n, 64, torch.cuda.current_stream(self.device).cuda_stream)
        xcc = out[:n].cpu().numpy()
        colocated = all(len(set(xcc[r::8].tolist())) == 1 for r in range(8))
        if not colocated:
            nat_call('ocr_set_lstm_proto', 0)

    def _layout(self, net):
        """Flat parameter / gradient layout [early rest | early regularised | late regularised | late rest] — layout.py."""
        lay = FlatLayout(net.param_specs.values(), ALIGN, order=[op.name for op in self.ops])
        self._owner = lay.owner
        self.specs, self.offsets, self.n_total = lay.specs, lay.offsets, lay.n_total
        self.split_layer, self.reg_range, self.late_begin = lay.split_layer, lay.reg_range, lay.late_begin
        dev = self.device
        self.params = torch.zeros(self.n_total, dtype=F32, device=dev)
        # the gradient buffer has a hidden tail of GUARD_PAD floats behind the n_total gradients: word 0 of it is the data-parallel DROP FLAG (1.0 when
        # a persistent LSTM launch of this rank's step timed out, ocr_guard_flag); it rides at the end of the late bucket's all-reduce, so after
        # the exchange it is > 0 on EVERY rank iff any rank raised it, and every rank's optimiser launch drops the same step (_optim_body)
        self._grads_store = torch.zeros(self.n_total + self.GUARD_PAD, dtype=F32, device=dev)
        self.grads = self._grads_store[:self.n_total]
        self.drop_flag = self._grads_store[self.n_total:self.n_total + 1]
        self.params_bf16 = torch.zeros(self.n_total, dtype=BF16, device=dev)

    GUARD_PAD = 64         # floats behind the gradients (256 B: the exchanged range stays 256-B granular)

    def offset(self, name):
        return self.offsets[name]

    def _view(self, flat, name):
        s = self.specs[name]
        n = int(np.prod(s.shape))
        o = self.offsets[name]
        return flat[o:o + n].view(s.shape)

    def param(self, name):
        return self._view(self.params, name)

    def grad(self, name):
        return self._view(self.grads, name)

    def shadow(self, name):
        return self._view(self.params_bf16, name)

    def _init_params(self, seed):
        self.params.copy_(init_host_parameters(self.specs, self.offsets, self.n_total, seed))

    def load_arrays(self, arrays):
        """{TF variable name: numpy array} -> parameters (shapes must match the TF layouts)."""
        for name, arr in arrays.items():
            t = torch.as_tensor(np.asarray(arr, np.float32))
            if tuple(t.shape) != tuple(self.specs[name].shape):
                raise ValueError('%s: shape %s does not match %s' % (name, tuple(t.shape), self.specs[name].shape))
            self.param(name).copy_(t)
        self.refresh_weights()

    def state_arrays(self):
        return {name: self.param(name).detach().cpu().numpy().copy() for name in self.specs}

    PACK_DTYPE = np.dtype([('type', '<i4'), ('R', '<i4'), ('Cc', '<i4'), ('lstm_units', '<i4'), ('ldin', '<i8'),
                           ('ldout', '<i8'), ('src', '<u8'), ('dst', '<u8'), ('n', '<i8'), ('block_start', '<i4'),
                           ('nblocks', '<i4')])       # == struct PackJob in csrc/nn_ops.hip (64 bytes)

    def _build_pack_table(self):
        # plain bf16 copies only of the variables some op reads through Engine.shadow (the 3x3 convolution weights — most of the
        # parameters — have their own packed layouts): a flat shadow of everything was 28 MB read + 14 MB written per step
        jobs = []
        for op in self.ops:
            for name in op.shadow_params():
                n = int(np.prod(self.specs[name].shape))
                assert n % 4 == 0, name
                jobs.append(dict(type=3, src=self._view(self.params, name).reshape(-1), dst=self._view(self.params_bf16, name).reshape(-1), n=n))
        for op in self.ops:
            jobs.extend(op.pack_jobs())
        tab = np.zeros(len(jobs), self.PACK_DTYPE)
        start = 0
        for i, j in enumerate(jobs):
            t = j['type']
            if t == 0:
                nb = ((j['Cc'] + 63) // 64) * ((j['R'] + 63) // 64)
            elif t == 1:
                nb = min((j['n'] + 255) // 256, 512)
            elif t == 2:
                nb = min((j['R'] * j['Cc'] // 4 + 255) // 256, 512)
            elif t == 4:
                nb = min((j['n'] + 255) // 256, 512)
            else:
                nb = min((j['n'] // 4 + 255) // 256, 2048)
            tab[i] = (t, j.get('R', 0), j.get('Cc', 0), j.get('lstm_units', 0), j.get('ldin', 0), j.get('ldout', 0),
                      j['src'].data_ptr(), j['dst'].data_ptr(), j.get('n', 0), start, nb)
            start += nb
        assert tab.itemsize == 64
        self._pack_table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        self._pack_njobs, self._pack_blocks = len(jobs), start

    def refresh_weights(self):
        if not hasattr(self, '_pack_table'):
            self._build_pack_table()
        ops.pack_jobs(self._pack_table, self._pack_njobs, self._pack_blocks)
        for op in self.ops:
            op.refresh()

    # ------------------------------------------------------------------ lowering
    def _lower(self, net):
        """Topological lowering of the plan reachable from 'logits' (a chain for the shipped models, a DAG with residual
        adds for deeper extractors).  Only data edges count: the second input of bi_lstm is the time_step_len slot."""
        table = {'conv': _ConvOp, 'fc': _FcOp, 'max_pool': _PoolOp, 'reshape_squeeze': _ViewOp, 'dropout': _DropoutOp, 'bi_lstm': _BiLstmOp, 'lstm': _BiLstmOp,
                 'add': _AddOp, 'relu': _ReluOp, 'batch_norm': _BatchNormOp, 'avg_pool': _AvgPoolOp, 'concat': _ConcatOp,
                 'softmax': _SoftmaxOp}
        data_op = _InputOp(self)
        data_op.dy_needed = lambda: False
        built = {}
        self.ops = []
        for nd in execution_order(net.get_output('logits')):
            if nd.op not in table:
                raise NotImplementedError('layer %r (%s) has no gfx950 lowering yet' % (nd.op, nd.name))
            ins = [data_op if i.op == 'input' else built[id(i)] for i in (nd.inputs if nd.op in ('add', 'concat') else nd.inputs[:1])]
            op = table[nd.op](self, nd, ins[0])
            op.inputs = ins
            op.key = '%02d:%s' % (len(self.ops), nd.name)
            for i in ins:
                i.grad_owner().consumers += 1
            self.ops.append(op)
            if nd.op == 'conv' and (nd.attrs['s_h'], nd.attrs['s_w']) != (1, 1):       # strided: stride-1 convolution + strided pick
                if nd.attrs['bn']:
                    raise NotImplementedError('%s: a strided convolution with batch norm is not lowered' % nd.name)
                sub = _SubsampleOp(self, nd, op)
                sub.name = nd.name + '/stride'
                sub.inputs, sub.key = [op], '%02d:%s/stride' % (len(self.ops), nd.name)
                op.consumers += 1
                self.ops.append(sub)
                op = sub
            built[id(nd)] = op
        if os.environ.get('OCR_FUSE_ADD_RELU', '1') != '0':
            # residual blocks: add -> relu (the add's only consumer) as ONE pass forward, and one masked delivery per input backward
            for r in self.ops:
                a = r.prev
                if isinstance(r, _ReluOp) and isinstance(a, _AddOp) and a.consumers == 1:
                    a.relu, r.fused_into = True, a
                    a.consumers = r.consumers
                    for p in a.inputs:                 # a (BN, no ReLU) conv feeding only this add: mask applied inside its BN backward
                        if (isinstance(p, _ConvOp) and p.bn and not p.relu and p.consumers == 1 and p.kind != 'c1'
                                and a.inputs.count(p) == 1):
                            p.mask_from = a
                    # ... and the LAST such conv to execute also does the add + relu in its batch-norm apply pass
                    cands = [p for p in a.inputs if getattr(p, 'mask_from', None) is a]
                    if cands and len(a.inputs) == 2:
                        p = max(cands, key=lambda q: self.ops.index(q))
                        other = a.inputs[1] if a.inputs[0] is p else a.inputs[0]
                        if isinstance(other, _InputOp) or self.ops.index(other) < self.ops.index(p):
                            p.tail_into, a.fwd_by = (a, other), p
        if os.environ.get('OCR_FUSE_CONV_POOL', '1') != '0':
            for b in self.ops:
                a = b.prev
                if (isinstance(b, _PoolOp) and isinstance(a, _ConvOp) and a.kind == '3x3' and a.relu and not a.bn and a.biased
                        and a.consumers == 1 and (b.kw_t, b.kh_f) in ((1, 2), (2, 2))):
                    a.pool_after = b
        if os.environ.get('OCR_FUSE_BN_POOL', '1') != '0':
            # conv + batch norm + ReLU followed by the 1 x 2 max-pool over the feature axis as its only consumer (LSTM_train.py:32-33: conv4_2
            # -> pool): the batch-norm apply pass writes the pooled tensor too, and the batch-norm backward passes route the pool's gradient
            # themselves — no max-pool forward / backward launches, no full-resolution gradient tensor written and re-read twice
            for b in self.ops:
                a = b.prev
                if (isinstance(b, _PoolOp) and isinstance(a, _ConvOp) and a.bn and a.kind != 'c1' and a.consumers == 1
                        and a.tail_into is None and a.mask_from is None and (b.kw_t, b.kh_f) == (1, 2)):
                    a.bn_pool, b.bn_fused_into = b, a
        if self.fuse_conv1_pool:
            for a, b in zip(self.ops[:-1], self.ops[1:]):
                if (isinstance(a, _ConvOp) and a.kind == 'c1' and a.relu and isinstance(b, _PoolOp) and b.prev is a
                        and a.consumers == 1 and (b.kw_t, b.kh_f) == (2, 2)):
                    a.fused_pool, b.fused_into = b, a

    # ------------------------------------------------------------------ gradient delivery (residual graphs)
    def _scratch(self, sp, like):
        key = (tuple(like.shape), like.dtype)
        if key not in sp.scratch:
            sp.scratch[key] = torch.empty_like(like)
        return sp.scratch[key]

    def grad_dst(self, sp, producer):
        """Where a consumer's backward kernel should write d(loss)/d(producer output), plus a `finish` callback.
        Single-consumer producers (every tensor of the shipped chain) get their dy buffer directly; for a tensor with
        several consumers the first delivery writes it and later ones go through a scratch buffer + one add kernel."""
        d = producer.dy(sp)
        if d is None:
            return None, (lambda: None)
        owner = producer.grad_owner()
        if owner.consumers <= 1 or owner.key not in sp.dy_done:
            sp.dy_done.add(owner.key)
            return d, (lambda: None)
        tmp = self._scratch(sp, d)
        return tmp, (lambda: ops.eltwise(0, d, tmp, d))

    def grad_dst_acc(self, sp, producer, can_accumulate):
        """grad_dst for a consumer whose kernel can add into the destination itself: (dst, finish, accumulate)."""
        d = producer.dy(sp)
        if d is None:
            return None, (lambda: None), False
        owner = producer.grad_owner()
        first = owner.consumers <= 1 or owner.key not in sp.dy_done
        if first or not can_accumulate:
            dst, finish = self.grad_dst(sp, producer)
            return dst, finish, False
        return d, (lambda: None), True

    def deliver(self, sp, producer, grad, mask=None):
        """Pass-through gradient (add / relu backward): dy(producer) (+)= mask > 0 ? grad : 0."""
        d = producer.dy(sp)
        if d is None:
            return
        g = grad.view(d.shape)
        if mask is None and producer.mask_in_consumer:
            mask = producer.y(sp)
        owner = producer.grad_owner()
        first = owner.consumers <= 1 or owner.key not in sp.dy_done
        sp.dy_done.add(owner.key)
        if first:
            if mask is None:
                d.copy_(g)
            else:
                ops.eltwise(2, g, mask.view(d.shape), d)
        elif mask is None:
            ops.eltwise(0, d, g, d)
        else:
            ops.eltwise(4, g, mask.view(d.shape), d)        # d += mask > 0 ? g : 0

    def plan(self, N, W):
        key = (N, W)
        if key not in self.plans:
            self.plans[key] = ShapePlan(self, N, W)
        return self.plans[key]

    # ------------------------------------------------------------------ input binding
    def _bind(self, sp, data, seq_len, labels=None, labels_len=None):
        def as_i32(v):
            return v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, np.int32))
        sl = as_i32(seq_len)
        lab = ll = None
        if labels is not None:
            lab, ll = as_i32(labels), as_i32(labels_len)
            if lab.numel() > sp.labels.numel():
                raise ValueError('flat label vector longer than batch * max_label_len (%d)' % sp.labels.numel())
        ints = [t for t in (sl, lab, ll) if t is not None]
        if (torch.is_tensor(data) and data.is_cuda and data.dtype in (torch.uint8, F32) and data.is_contiguous()
                and data.numel() == sp.x.numel() and all(t.is_cuda and t.dtype == I32 and t.is_contiguous() for t in ints)
                and sl.numel() == sp.seq_len.numel() and (ll is None or ll.numel() == sp.labels_len.numel())):
            # device-resident batch (prefetching pipeline, bench): ONE kernel binds pixels (uint8 -> / 255, a quarter of the H2D
            # bytes) and the three int vectors.  Per-tensor copies were four blit kernels (hipMemcpyAsync D2D, ~5 us each, each
            # preceded by a queue barrier that left the GPU idle at the start of every step).
            ops.bind_batch(data, sp.x, sl, sp.seq_len, lab, sp.labels, ll, sp.labels_len)
            return
        if torch.is_tensor(data) and data.dtype == torch.uint8:
            ops.u8_to_unit_f32(data if data.is_cuda else data.to(self.device, non_blocking=True), sp.x)
            dsts, srcs = [sp.seq_len], [sl]
        else:
            x = torch.as_tensor(np.asarray(data, np.float32)) if not torch.is_tensor(data) else data
            dsts, srcs = [sp.x, sp.seq_len], [x, sl]
        if lab is not None:
            dsts += [sp.labels[:lab.numel()], sp.labels_len]
            srcs += [lab, ll]
        for d, t in zip(dsts, srcs):
            d.copy_(t, non_blocking=True)

    # ------------------------------------------------------------------ forward / backward bodies (capturable)
    def step_counter(self):
        """Device double counting the completed optimiser steps (scalars[6]) — the per-step salt of the dropout masks."""
        return self.scalars[6:7]

    def _forward(self, sp, training=False):
        self.training = training
        # the flat gradient buffer is cleared by the forward pass's first kernel where that is conv1 + pool (one launch less); else by a fill
        self._zero_pending = bool(training) and self.grads.numel() % 4 == 0 and os.environ.get('OCR_FUSE_ZERO', '1') != '0'
        if training and not self._zero_pending:
            self.grads.zero_()
        # likewise the fill of the persistent LSTM launches' hand-off blocks (OCR_FUSE_RINGFILL=0: every launch fills its own); a graph whose
        # first kernel is not conv1 + pool never consumes it and its LSTM launches prepare their blocks themselves
        sp.rings_ready, sp.rings_armed = set(), False
        self._ones_pending = sp.lstm_arena if (training and getattr(sp, 'lstm_arena', None) is not None
                                               and os.environ.get('OCR_FUSE_RINGFILL', '1') != '0') else None
        for op in self.ops:
            op.fwd(sp)
        self._ones_pending = None
        if self._zero_pending:
            self.grads.zero_()
            self._zero_pending = False

    def _loss_and_backward(self, sp, flush=True):
        sp.dy_done = set()
        sp.w9_pending = []
        sp.tn_pending = []
        logits = self.ops[-1].y(sp)
        # loss = mean over the GLOBAL batch -> d loss / d cost_n = 1 / (N * world)   (network.py:655)
        scale = ocr_dist.loss_scale(sp.N, self.world)
        if ops.ctc_train_supported(sp.C, sp.T, self.max_label_len):
            ops.ctc_loss_train(logits, self.ops[-1].dy(sp), scale, sp.labels, sp.labels_len, sp.seq_len, self.max_label_len,
                               sp.costs, blank=0)
        else:
            ops.ctc_loss(logits, sp.labels, sp.labels_len, sp.seq_len, self.max_label_len, blank=0, want_grad=True,
                         workspace=sp.ctc_ws, costs=sp.costs, grads=sp.ctc_grad)
            ops.tnc_to_ntc_bf16(sp.ctc_grad, self.ops[-1].dy(sp), scale)
        for op in reversed(self.ops[self.split_op:]):
            op.bwd(sp)
        self._flush_tn(sp)                  # (always: the exchange of the late gradients may follow this body)
        self._join_tn(sp)
        if flush:
            self._flush_w9(sp)

    def _backward_early(self, sp):
        for op in reversed(self.ops[:self.split_op]):
            op.bwd(sp)
        self._flush_tn(sp)
        self._join_tn(sp)
        self._flush_w9(sp)

    def tn_push(self, sp, job, keep, fallback):
        """A plain weight-gradient product (X^T dY of conv5 / of the BiLSTM cells) joins the pending list; TWO of them go out as ONE launch of
        the ping-pong kernel (csrc/gemm_tn3.hip: no split over the contraction, no atomics).  `keep`: tensors the descriptor points into;
        `fallback`: the product as its own launch (gemm_tn2), used when it stays alone — with a split over the pixels that kernel fills the
        chip better than 64-96 unsplit tiles would."""
        if not self.tn_defer or not ops.gemm_tn_jobs_supported([job]):
            fallback()
            return
        sp.tn_pending.append((job, keep, fallback))
        if len(sp.tn_pending) == 2:
            self._flush_tn(sp)

    def _flush_tn(self, sp):
        pend, sp.tn_pending = sp.tn_pending, []
        if len(pend) == 2:
            if self.tn_side:
                # the two plain weight-gradient products (47 us, operand-fill-bound on 160 of 256 CUs) are needed by the optimiser / the
                # exchange only: they run on a side stream beside the short kernels that follow on the main chain (conv5's data gradient,
                # col2im, conv4_2's batch-norm backward passes); _join_tn() at the end of the backward body
                main = torch.cuda.current_stream(self.device)
                self.side_stream.wait_stream(main)
                with torch.cuda.stream(self.side_stream):
                    ops.gemm_tn_jobs([pend[0][0], pend[1][0]])
                sp.tn_side_open = True
            else:
                ops.gemm_tn_jobs([pend[0][0], pend[1][0]])
        else:
            for _, _, fallback in pend:
                fallback()

    def _join_tn(self, sp):
        if getattr(sp, 'tn_side_open', False):
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
            sp.tn_side_open = False

    W9_DEFER_MAX_BYTES = 192 << 20       # total slab bytes of a plan up to which the reductions are deferred (Infinity Cache: 256 MB;
                                         # the headline plan has 162 MB); OCR_W9_DEFER_MAX_MB overrides
    W9_JOB_DTYPE = np.dtype([('dw', '<u8'), ('part', '<u8'), ('dbias', '<u8'), ('cs_part', '<u8'), ('n4', '<i8'), ('slab4', '<i8'),
                             ('S', '<i4'), ('rows', '<i4'), ('Cout', '<i4'), ('block_start', '<i4')])   # == struct W9ReduceJob (wgrad9.hip)

    def _flush_w9(self, sp):
        """ONE launch for the slab reductions the 3x3 weight-gradient kernels of this backward body left pending (dw += sum of the
        slabs, db += column sums).  The job table depends only on the plan's buffers, so it is uploaded once — on the eager run
        that precedes every capture — and the captured graphs replay the launch with the same device table."""
        pend, sp.w9_pending = sp.w9_pending, []
        sp.w9_pending_bytes = 0
        if not pend:
            return
        raw = b''.join(job for job, _ in pend)
        ent = sp.w9_tables.get(raw)
        if ent is None:
            tab = np.frombuffer(raw, dtype=self.W9_JOB_DTYPE).copy()
            assert tab.itemsize == 64 and len(tab) == len(pend)
            start = 0
            for i, (_, nblk) in enumerate(pend):
                tab['block_start'][i] = start
                start += nblk
            dev = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
            ent = sp.w9_tables[raw] = (dev, len(pend), start)
        ops.wgrad9_reduce_jobs(*ent)

    def _capture(self, fn):
        """hipGraph capture of fn() with Python's cyclic garbage collector paused: a collection that frees device tensors of
        an unreferenced engine (engine <-> op cycles) in the middle of a capture aborts the process."""
        import gc
        gc.collect()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        was = gc.isenabled()
        gc.disable()
        try:
            # thread_local: other threads keep issuing work while this one captures — the input pipeline's feeder thread (H2D copies
            # on its own stream) and, with several ranks, RCCL's watchdog; in the default global mode any of their calls may
            # invalidate the capture
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                fn()
        finally:
            if was:
                gc.enable()
        return g

    def _run_split(self, sp):
        """forward + loss + backward of the late layers as one graph, backward of the early layers as a second one (data-
        parallel runs: the exchange of the late gradients is issued between the two)."""
        def body1():
            self._forward(sp, training=True)
            self._loss_and_backward(sp)
            self._publish_guard(sp)

        def body2():
            self._backward_early(sp)

        if not self.use_graphs:
            body1()
            return body2
        if getattr(sp, 'graph_fb1', None) is None:
            body1(); body2()                                    # warm-up outside capture (also uploads the drop flag's address table)
            sp.graph_fb1, sp.graph_fb2 = self._capture(body1), self._capture(body2)
        sp.graph_fb1.replay()
        return sp.graph_fb2.replay

    def _check_graph_collectives(self):
        """Can a collective of this group be captured in a hipGraph and replayed?  One eager all-reduce (communicator set-up), then a captured one on
        the communication stream replayed twice: the buffer must hold world^2 x its start value.  Every rank must agree (an eager MIN all-reduce of
        the verdict), else nobody uses the captured schedule.  gloo (tests: ranks sharing one GPU, staged through the host) is never capturable."""
        import torch.distributed as dist
        fake = bool(os.environ.get('OCR_FAKE_WORLD'))
        if not fake:
            if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(self.group) != 'nccl':
                return False
        ok = True
        try:
            t = torch.ones(256, dtype=F32, device=self.device)
            main = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                ocr_dist.allreduce_sum_(t, self.group, force=self.force_allreduce)
            main.wait_stream(self.comm_stream)
            torch.cuda.synchronize()

            def body():
                cur = torch.cuda.current_stream(self.device)
                self.comm_stream.wait_stream(cur)
                with torch.cuda.stream(self.comm_stream):
                    ocr_dist.allreduce_sum_(t, self.group, force=self.force_allreduce)
                cur.wait_stream(self.comm_stream)
            g = self._capture(body)
            t.fill_(1.0)
            g.replay(); g.replay()
            torch.cuda.synchronize()
            w = float(self.world if (fake or self.world > 1) else 1)
            ok = bool(float(t[0]) == w * w and float(t[-1]) == w * w)
        except Exception as e:       # noqa: BLE001
            import sys
            sys.stderr.write('[engine] OCR_DP_GRAPH=1: a captured collective failed on this group (%s: %s) — three-graph schedule instead\n' % (type(e).__name__, e))
            ok = False
        if not fake and dist.get_world_size(self.group) > 1:
            v = torch.tensor([1.0 if ok else 0.0], device=self.device)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.group)
            ok = bool(float(v[0]) == 1.0)
        return ok

    def _run_dp_graph(self, sp):
        """The whole data-parallel step of this shape as ONE hipGraph (OCR_DP_GRAPH=1): forward, CTC, backward of the late layers, the drop flag ->
        [communication stream: all-reduce of the late bucket] || backward of the early layers -> [communication stream: all-reduce of the early
        bucket] -> join -> clip + optimiser + re-pack.  Same kernels, same order, same two buckets as the three-graph schedule of train_step."""
        if not self.opt_ready:
            self.setup_optimizer()

        def exchange(lo, hi):
            cur = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                self.allreduce_grads(lo, hi)

        def body():
            self._forward(sp, training=True)
            self._loss_and_backward(sp)
            self._publish_guard(sp)
            exchange(self.late_begin, self.n_total)
            self._backward_early(sp)
            exchange(0, self.late_begin)
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            self._optim_body(None)

        if getattr(sp, 'graph_dp', None) is None:
            # warm-up outside capture: forward + backward only (lazy module loads) — the optimiser mutates state, its first run is the first replay
            self._guard_addrs(sp)                            # (the address table of the drop flag's words is uploaded here, outside the capture)
            self._forward(sp, training=True)
            self._loss_and_backward(sp)
            self._backward_early(sp)
            sp.graph_dp = self._capture(body)
        sp.graph_dp.replay()

    def _run_step(self, sp):
        """Whole training step of this shape as one hipGraph (no exchange between backward and optimiser on a single GPU: the
        graph boundary cost ~8 us of idle GPU per step).  Only the forward/backward part is warmed up eagerly — the optimiser
        mutates state, so its first execution is the first replay."""
        if not self.opt_ready:
            self.setup_optimizer()

        def fb():
            self._forward(sp, training=True)
            self._loss_and_backward(sp, flush=False)      # one merged weight-gradient reduction at the end of the whole backward
            self._backward_early(sp)

        if getattr(sp, 'graph_step', None) is None:
            self._guard(sp)                              # (its address table is uploaded here, outside the capture)
            fb()

            def body():
                fb()
                self._optim_body(sp)
            sp.graph_step = self._capture(body)
        sp.graph_step.replay()

    def _run(self, sp, which):
        """Run (or capture-then-replay) the forward(+backward) body for this shape."""
        attr = 'graph_fb' if which == 'fb' else 'graph_fwd'

        def body():
            self._forward(sp, training=(which == 'fb'))
            if which == 'fb':
                self._loss_and_backward(sp, flush=False)
                self._backward_early(sp)
                self._publish_guard(sp)

        if not self.use_graphs:
            body()
            return
        g = getattr(sp, attr)
        if g is None:
            body()                                   # warm-up outside capture (lazy module loads)
            g = self._capture(body)
            setattr(sp, attr, g)
        g.replay()

    # ------------------------------------------------------------------ public API
    def forward(self, data, seq_len):
        """Inference forward; returns the logits [T, N, C] (time-major, as network.py:127-128)."""
        data_shape = data.shape
        sp = self.plan(data_shape[0], data_shape[1])
        self._bind(sp, data, seq_len)
        self._run(sp, 'fwd')
        return self.ops[-1].y(sp)

    def decode(self, data, seq_len, method='beam', beam_width=100):
        """Decoded label lists with zeros dropped (what accuracy_calculation / decodeRes see).
        method='beam'  : the reference's decoder — tf.nn.ctc_beam_search_decoder(beam 100, merge_repeated=True), whose
                         blank is class C-1; class 0 is an ordinary symbol that sparse_to_dense/ignore_value=0 strip later
                         (network.py:656-657, training.py:32).
        method='greedy': best path with blank 0."""
        sp = self.plan(data.shape[0], data.shape[1])
        self._bind(sp, data, seq_len)
        self._run(sp, 'fwd')
        logits = self.ops[-1].y(sp)
        if method == 'greedy':
            out, lens = ops.ctc_greedy_decode(logits, sp.seq_len, blank=0, pad_value=0)
        else:
            out, lens, _ = ops.ctc_beam_decode(logits, sp.seq_len, beam_width=beam_width, merge_repeated=True, pad_value=0)
        out = out.cpu().numpy()
        lens = lens.cpu().numpy()
        return [[int(v) for v in out[i, :lens[i]] if v != 0] for i in range(out.shape[0])]

    def setup_optimizer(self, solver=None, lr=None):
        c = self.cfg.TRAIN
        self.solver = ops.SOLVERS.get(solver or c.SOLVER, 1)       # reference: anything else -> Momentum (train.py:76)
        self.lr = float(c.LEARNING_RATE if lr is None else lr)
        # slot initial values as TensorFlow creates them: Adam m = v = 0, Momentum accumulator 0, RMSProp "rms" slot = ONES
        # (tf.train.RMSPropOptimizer: the first steps are ~ lr * g / sqrt(0.9 + 0.1 g^2), not lr * sign(g) / sqrt(0.1))
        self.state1 = torch.ones_like(self.params) if self.solver == 2 else torch.zeros_like(self.params)
        self.state2 = torch.zeros_like(self.params) if self.solver == 0 else None
        self.scalars.zero_()                             # in place: captured graphs keep pointing at it
        ops.optim_init(self.scalars, self.lr)
        self.opt_ready = True
        self.graph_opt = None                            # captured optimiser graphs hold the old slot tensors
        for sp in getattr(self, 'plans', {}).values():
            sp.graph_step = None
            sp.graph_dp = None

    def scale_lr(self, gamma):
        ops.optim_set_lr(self.scalars, gamma, multiply=True)
        self.lr *= gamma

    def _guard_on(self):
        return os.environ.get('OCR_LSTM_TIMEOUT_GUARD', '1') != '0'

    def _dp(self):
        return self.world > 1 or self.force_allreduce

    def _guard_addrs(self, sp):
        """Device addresses of the plan's persistent-LSTM error words (None: the plan has no persistent launch)."""
        if sp is None:
            return None
        g = getattr(sp, '_guard_addrs', None)
        if g is None:
            words = getattr(sp, 'lstm_sync', ())
            g = sp._guard_addrs = (torch.tensor([w[-1:].data_ptr() for w in words], dtype=torch.int64, device=self.device) if words else False)
        return g if g is not False else None

    def _guard(self, sp):
        """Single GPU: the error words themselves guard the update — a hand-off time-out inside the step drops the step's update on the device
        instead of applying a garbage gradient (ocr_optim_step_guarded2; the report, one step later, logs it: report_wait).  With several ranks
        every replica has to drop the SAME step: there the words are folded into the drop flag that rides on the late bucket (_publish_guard)."""
        if sp is None or self._dp() or not self._guard_on():
            return None
        return self._guard_addrs(sp)

    def _publish_guard(self, sp):
        """Data parallel: one tiny launch at the end of the backward graph that holds the LSTM (both error words are final there) writes this
        rank's 1.0 / 0.0 into the drop flag — the word right behind the late bucket, which the all-reduce that follows sums over the ranks."""
        hook = getattr(self, '_fault_hook', None)          # fault injection (tests): runs between the LSTM launches and the flag launch
        if hook is not None:
            hook(sp)
        if self._dp() and self._guard_on():
            ops.guard_flag(self._guard_addrs(sp), self.drop_flag)

    def _optim_body(self, sp=None):
        c = self.cfg.TRAIN
        if self.solver == 0:
            b1, b2, eps = 0.9, 0.999, 1e-8
        elif self.solver == 1:
            b1, b2, eps = float(c.MOMENTUM), 0.0, 0.0
        else:
            b1, b2, eps = 0.9, 0.0, 1e-10
        ops.optim_step(self.params, self.grads, self.state1, self.state2, self.reg_range, float(c.WEIGHT_DECAY), 10.0,
                       self.solver, b1, b2, eps, self.scalars, guard=self._guard(sp),
                       drop_flag=self.drop_flag if (self._dp() and self._guard_on()) else None)
        self.refresh_weights()

    def optimizer_step(self, sp=None):
        """sp: the plan whose backward pass produced the gradients — its LSTM error words guard the update when the step runs eagerly on one GPU
        (the shared optimiser graph of the multi-graph schedules is plan-independent: it is guarded by the exchanged drop flag)."""
        if not self.opt_ready:
            self.setup_optimizer()
        if not self.use_graphs:
            self._optim_body(sp)
            return
        if self.graph_opt is None:
            # capture WITHOUT a warm-up run: the optimiser mutates state, so the first real step is the capture's replay
            self.graph_opt = self._capture(self._optim_body)
        self.graph_opt.replay()

    def allreduce_grads(self, lo=0, hi=None):
        """Data-parallel exchange: all-reduce (sum) of a range of the flat fp32 gradient buffer over RCCL/xGMI.  The 1/world
        factor is already folded into the CTC gradient hand-off, so the sum IS the global-batch mean gradient; the
        clip then sees the same global norm a single GPU would see at the global batch size."""
        if self.world > 1 or self.force_allreduce:
            hi = self.n_total if hi is None else hi
            if hi == self.n_total and self._guard_on():
                hi += self.GUARD_PAD              # the range that ends the buffer carries the drop flag (_publish_guard)
            if hi > lo:
                ocr_dist.allreduce_sum_(self._grads_store[lo:hi], self.group, force=self.force_allreduce)

    def train_step(self, data, labels, labels_len, seq_len, fetch_loss=True):
        """One optimisation step on a batch laid out as gen.py:41-67 produces it.  Returns the total loss
        (mean CTC cost of the local batch + L2 term) as a Python float when fetch_loss, else None.
        Data-parallel schedule: graph 1 (forward, CTC, backward of the late layers) -> the late gradients (one contiguous
        range, ~87 % of the bytes for the CRNN) are all-reduced on the communication stream WHILE graph 2 (backward of the early layers)
        runs -> the early gradients are all-reduced on the same stream -> join -> graph 3 (clip + optimiser + re-pack)."""
        sp = self.plan(data.shape[0], data.shape[1])
        self._bind(sp, data, seq_len, labels, labels_len)
        if (self.world > 1 or self.force_allreduce) and self.overlap_allreduce and self.split_op > 0 and self.dp_graph and self.use_graphs:
            self._run_dp_graph(sp)
        elif (self.world > 1 or self.force_allreduce) and self.overlap_allreduce and self.split_op > 0:
            main = torch.cuda.current_stream(self.device)
            pc = time.perf_counter
            t0 = pc()
            rest = self._run_split(sp)
            t1 = pc()
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                self.allreduce_grads(self.late_begin, self.n_total)
            t2 = pc()
            rest()                                       # backward of the early layers, concurrent with the exchange above
            t3 = pc()
            # the early bucket goes out on the SAME stream as the late one: both collectives of a step are issued from one ordering
            # domain, in the same order on every rank (two streams relied on ProcessGroupNCCL's internal ordering)
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                self.allreduce_grads(0, self.late_begin)
            main.wait_stream(self.comm_stream)
            t4 = pc()
            self.optimizer_step()
            t5 = pc()
            # host-side enqueue time of the five phases (seconds, summed; bench.py prints the per-step averages so that a scaling record
            # explains itself: a host-bound schedule shows here, not in the kernel profile)
            h = self.dp_host_s
            h[0] += t1 - t0; h[1] += t2 - t1; h[2] += t3 - t2; h[3] += t4 - t3; h[4] += t5 - t4; h[5] += 1
        elif self.world == 1 and not self.force_allreduce and self.use_graphs:
            self._run_step(sp)                            # single GPU: forward, backward and optimiser as ONE graph
        else:
            self._run(sp, 'fb')
            self.allreduce_grads()
            self.optimizer_step(sp)
        self.iteration += 1
        self.last_plan = sp
        if fetch_loss:
            return self.last_loss()
        if self.iteration % 64 == 0:          # even when nobody reads the loss: look at the persistent LSTM's time-out words
            self._watchdog()                  # every 64 steps — WITHOUT a host sync (the report of 64 steps ago is read if it has landed)
        return None

    def _watchdog(self):
        """Queue a report of this step into the watchdog's OWN slot and evaluate the one queued 64 steps ago (an event query, no wait):
        a persistent-LSTM time-out raises at most 64 steps late instead of costing a pipeline drain every 64 steps."""
        pending = getattr(self, '_watch_pending', None)
        if pending is not None:
            if not pending[1].query():
                return                            # the 'watch' slot's copy is still in flight: keep that handle, queue nothing over it
            self.report_wait(pending, update_mirrors=False)      # a 64-steps-old report must not overwrite last_ctc / last_reg / last_gnorm
        self._watch_pending = self.report_async(_slot='watch')

    def last_loss(self):
        """Loss of the last step: mean CTC cost of the local batch + L2 term.  ONE tiny kernel gathers the mean cost, the
        optimiser's scalars and the persistent-LSTM time-out words into 4 doubles, ONE 32-byte D2H copy into pinned memory and ONE
        stream wait follow (four blocking .cpu() / .item() round trips cost ~0.1 ms per iteration of the training loop, which
        fetches the loss every step like the reference: train.py:130,139; four async copies were four blit kernels)."""
        return self.report_wait(self.report_async(_slot='sync'))

    REPORT_SLOTS = 4       # handles that may be outstanding at once (the training loop keeps one, OCR_LOSS_LAG = 1)

    def report_async(self, _slot=None):
        """Queue the report of the step that was just issued (kernel + 32-byte copy + event) and return a handle for report_wait().
        The training loop waits for it one iteration LATER, so the host never drains the queue between two steps.
        Slot ownership (a two-slot ring shared with the engine's own every-64-steps check let that check take the slot
        a pending handle still pointed at — one logged loss in 64 was the NEXT step's): handles rotate over REPORT_SLOTS slots that
        only this function hands out, a slot is reused only after REPORT_SLOTS - 1 further handles were issued and its previous
        handle was waited for (checked), and the synchronous last_loss() and the watchdog each have a slot of their own."""
        sp = self.last_plan
        words = getattr(sp, 'lstm_sync', ())
        ring = getattr(sp, '_report', None)
        if ring is None:
            addrs = torch.tensor([w[-1:].data_ptr() for w in words], dtype=torch.int64, device=self.device) if words else None
            mk = lambda: [torch.zeros(4, dtype=torch.float64, device=self.device), torch.zeros(4, dtype=torch.float64).pin_memory(),
                          torch.cuda.Event(), False]                      # [device block, pinned host block, event, handle outstanding]
            ring = sp._report = dict(addrs=addrs, i=0, slots=[mk() for _ in range(self.REPORT_SLOTS)], sync=mk(), watch=mk())
        if _slot is None:
            ring['i'] = (ring['i'] + 1) % self.REPORT_SLOTS
            slot = ring['slots'][ring['i']]
            if slot[3]:
                raise RuntimeError('report_async: %d reports are outstanding — call report_wait() on the older handles first'
                                   % self.REPORT_SLOTS)
        else:
            slot = ring[_slot]
        slot[3] = True
        dev, host, ev = slot[0], slot[1], slot[2]
        ops.step_report(sp.costs, self.scalars if self.opt_ready else None, ring['addrs'], dev)
        host.copy_(dev, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        return (host, ev, words, self.opt_ready, slot)

    def report_wait(self, handle, update_mirrors=True):
        host, ev, words, opt_ready, slot = handle
        ev.synchronize()
        slot[3] = False
        ctc, reg2, gnorm, bits = (float(v) for v in host.numpy())
        reg = 0.5 * float(self.cfg.TRAIN.WEIGHT_DECAY) * reg2 if (self.cfg.TRAIN.WEIGHT_DECAY > 0 and opt_ready) else 0.0
        if update_mirrors:
            self.last_ctc, self.last_reg, self.last_gnorm = ctc, reg, gnorm if opt_ready else 0.0
        bits = int(bits)
        dropped, bits = bool(bits >> 40 & 1), bits & 0xFFFFFFFF
        if dropped:
            # THIS step's update was dropped on the device (the step's own scalars[72], exported by the report kernel — not a comparison of the global
            # counter: a second report covering the same step, or two dropped steps in a row, used to raise — ADVICE r5): parameters and moments are
            # intact on every rank, training goes on; the reported loss of that step is meaningless.  bits == 0 here means ANOTHER rank timed out.
            self.dropped_reports = getattr(self, 'dropped_reports', 0) + 1
            bad = [i for i in range(len(words)) if (bits >> i) & 1]
            import sys
            sys.stderr.write('[engine] WARNING: %s reported an expired inter-workgroup wait; that step\'s update was dropped on the device%s, parameters intact\n'
                             % (('the persistent LSTM %s kernel' % ('forward', 'backward')[bad[0] % 2]) if bad else 'another rank\'s persistent LSTM kernel',
                                ' on every rank' if self._dp() else ''))
            return float('nan')
        if bits != 0:
            bad = [i for i in range(len(words)) if (bits >> i) & 1]
            # the sync block = group counters (64-word stride, at most 2 * ceil(N / 16) of them) | hand-off ring | error word: print the
            # counters only (the ring is hundreds of thousands of 0xFFFFFFFF words)
            w = words[bad[0]]
            raise NativeError('persistent LSTM %s kernel: inter-workgroup wait timed out (results invalid); group counters %s, error word %d'
                              % (('forward', 'backward')[bad[0] % 2], w[:64 * 16:64].tolist(), int(w[-1])))
        return ctc + reg

    def guard_counters(self):
        """(steps dropped by the guarded optimiser step, expired hand-off waits it saw) since setup_optimizer — two device doubles, one blocking read."""
        if not self.opt_ready:
            return 0, 0
        v = self.scalars[73:75].cpu().numpy()
        return int(v[0]), int(v[1])
