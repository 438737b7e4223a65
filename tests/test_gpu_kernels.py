"""-m gpu: every HIP kernel against an independent CPU computation on the same (bf16-rounded) inputs.

Tolerances: integer/index outputs bit-exact; fp32-accumulated contractions of bf16 operands within 2e-3 of the
operand-scale (fp32 summation order only) when the output is fp32 and within bf16 output quantisation (2^-8
relative) when the output is stored as bf16; CTC loss 1e-4 relative (north star asks 1e-3), gradients 1e-4 abs.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from lstm_ctc_ocr_amd import ops
from oracle import ctc as octc
from oracle import decode as odec
from oracle import graph as og

BF = torch.bfloat16


def bf(x):
    return x.to(BF).to(torch.float32)


def maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


def relerr(a, b):
    return maxerr(a, b) / max(float(b.double().abs().max()), 1e-12)


def gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


# ------------------------------------------------------------------------------------------- probe
def test_probe_tr16(dev):
    """Records the lane semantics of ds_read_b64_tr_b16 for two address patterns (evidence for gemm_tn's
    transposing read); asserts the hypothesis used by the TR path."""
    res = {}
    lanes = np.arange(64)
    tables = {
        "lane_linear": lanes * 8,
        # hypothesis H: group g = l>>4 reads a 4(k) x 16(col) row-major block with row stride 64 elements;
        # lane L=l&15 supplies the 8-byte chunk (row L>>2, cols 4*(L&3)..+3)
        "rowmajor64": (((lanes >> 4) * 4 + ((lanes & 15) >> 2)) * 64 + (lanes & 3) * 4) * 2,
    }
    for name, tab in tables.items():
        addr = torch.tensor(tab, dtype=torch.int32, device=dev)
        out = torch.zeros(256, dtype=torch.int32, device=dev)
        from lstm_ctc_ocr_amd import _native as nat
        nat.call("ocr_probe_tr16", addr.data_ptr(), out.data_ptr(), nat.stream())
        torch.cuda.synchronize()
        res[name] = out.cpu().numpy().reshape(64, 4).tolist()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/probe_tr16.json", "w"))
    got = np.array(res["rowmajor64"])
    exp = np.array([[((l >> 4) * 4 + j) * 64 + (l & 15) for j in range(4)] for l in range(64)])
    print("tr16 rowmajor64 lane0..3:", got[:4].tolist(), "expected", exp[:4].tolist())
    print("tr16 lane_linear lane0..3,16,17:", [res["lane_linear"][i] for i in (0, 1, 2, 3, 16, 17)])
    assert (got == exp).all(), "hypothesis H about ds_read_b64_tr_b16 does not hold — see gpurun_out/probe_tr16.json"


def test_occupy_cus_holds_the_stream_for_the_asked_time(dev):
    """ocr_occupy_cus (the CU stand-in of the one-GPU data-parallel emulation): resident for the asked time, not much longer."""
    from lstm_ctc_ocr_amd import _native as nat
    for blocks, lds in ((8, 96 * 1024), (32, 0)):
        nat.call("ocr_occupy_cus", blocks, 256, lds, 10.0, nat.stream())       # warm-up (sets the LDS attribute)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("ocr_occupy_cus", blocks, 256, lds, 400.0, nat.stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        assert 0.39 < ms < 0.6, (blocks, lds, ms)


# ------------------------------------------------------------------------------------------- CTC
def _ctc_case(dev, T, N, C, lens, in_lens, seed, blank=0, labels=None):
    rng = np.random.RandomState(seed)
    acts = (rng.randn(T, N, C) * 2).astype(np.float32)
    if labels is None:
        labels = [rng.randint(1, C, size=l).tolist() for l in lens]
    flat = np.array([v for l in labels for v in l], np.int32)
    ll = np.array([len(l) for l in labels], np.int32)
    il = np.array(in_lens, np.int32)
    ref_c, ref_g = octc.ctc_loss_c(acts, flat, ll, il, blank)
    a = torch.from_numpy(acts).to(dev)
    fl = torch.from_numpy(flat if len(flat) else np.zeros(1, np.int32)).to(dev)
    from lstm_ctc_ocr_amd import _native as nat
    for engine in (0, 1):          # the general one-wave kernel and the 4-wave LDS-resident kernel must both match
        nat.call("ocr_set_ctc_engine", engine)
        costs, grads = ops.ctc_loss(a, fl, torch.from_numpy(ll).to(dev), torch.from_numpy(il).to(dev), int(max(ll.max(), 1)), blank)
        torch.cuda.synchronize()
        assert np.allclose(costs.cpu().numpy(), ref_c, rtol=1e-4, atol=1e-4), (engine, costs.cpu().numpy(), ref_c)
        assert np.abs(grads.cpu().numpy() - ref_g).max() < 5e-4, (engine, np.abs(grads.cpu().numpy() - ref_g).max())
    costs, grads = ops.ctc_loss(a, fl, torch.from_numpy(ll).to(dev), torch.from_numpy(il).to(dev), int(max(ll.max(), 1)), blank)
    torch.cuda.synchronize()
    c = costs.cpu().numpy(); g = grads.cpu().numpy()
    assert np.allclose(c, ref_c, rtol=1e-4, atol=1e-4), (c, ref_c)
    assert np.abs(g - ref_g).max() < 5e-4, np.abs(g - ref_g).max()   # fp32 log-space over T frames
    if ops.ctc_train_supported(C, T, int(max(ll.max(), 1))):       # fused training form: bf16 [N,T,C] scaled gradient, in-kernel offsets
        gb = torch.full((N, T, C), 7.0, dtype=BF, device=dev)
        c3 = torch.empty(N, device=dev)
        ops.ctc_loss_train(a, gb, 0.25, fl, torch.from_numpy(ll).to(dev), torch.from_numpy(il).to(dev), int(max(ll.max(), 1)), c3, blank)
        assert np.allclose(c3.cpu().numpy(), ref_c, rtol=1e-4, atol=1e-4)
        want = bf(torch.from_numpy(ref_g).permute(1, 0, 2) * 0.25)
        assert maxerr(gb.float().cpu(), want) < 2e-3
    # score-only call must not touch gradients and give the same costs
    costs2, _ = ops.ctc_loss(a, fl, torch.from_numpy(ll).to(dev), torch.from_numpy(il).to(dev), int(max(ll.max(), 1)), blank,
                             want_grad=False)
    assert np.allclose(costs2.cpu().numpy(), ref_c, rtol=1e-4, atol=1e-4)


def test_ctc_known_answer(dev):
    acts = torch.tensor([[[0.1, 0.6, 0.1, 0.1, 0.1]], [[0.1, 0.1, 0.6, 0.1, 0.1]]], device=dev)
    lab = torch.tensor([1, 2], dtype=torch.int32, device=dev)
    one = torch.tensor([2], dtype=torch.int32, device=dev)
    costs, grads = ops.ctc_loss(acts, lab, one, one, 2)
    assert abs(float(costs[0]) - 2.46286) < 1e-5
    g = grads.cpu().numpy().reshape(2, 5)
    exp = np.full((2, 5), 0.177031); exp[0, 1] = exp[1, 2] = -0.708125
    assert np.abs(g - exp).max() < 1e-5


def test_ctc_tensorflow_known_answers(dev):
    """The device kernels on TensorFlow's own ctc_loss vectors (ctc_loss_op_test.py::testBasic; blank = C - 1 = 5): -log p = 3.34211 and
    5.42262, gradients = prob - onehot(the single alignment) — the CPU oracle is pinned on the same numbers in tests/test_oracle_ctc.py."""
    m0 = np.array([[0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553], [0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436],
                   [0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688], [0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533],
                   [0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]])
    m1 = np.array([[0.30176, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508], [0.24082, 0.397533, 0.0557226, 0.0546814, 0.0557528, 0.19549],
                   [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, 0.202456], [0.280884, 0.429522, 0.0326593, 0.0339046, 0.0326856, 0.190345],
                   [0.423286, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]])
    acts = torch.from_numpy(np.stack([np.log(m0), np.log(m1)], 1).astype(np.float32)).to(dev)
    lab = torch.tensor([0, 1, 2, 1, 0, 0, 1, 1, 0], dtype=torch.int32, device=dev)
    ll = torch.tensor([5, 4], dtype=torch.int32, device=dev); il = torch.tensor([5, 5], dtype=torch.int32, device=dev)
    from lstm_ctc_ocr_amd import _native as nat
    for engine in (0, 1):
        nat.call("ocr_set_ctc_engine", engine)
        costs, grads = ops.ctc_loss(acts, lab, ll, il, 5, 5)
        c = costs.cpu().numpy(); g = grads.cpu().numpy()
        assert abs(c[0] - 3.34211) < 5e-5 and abs(c[1] - 5.42262) < 5e-5, (engine, c)
        for n, (m, path) in enumerate(((m0, [0, 1, 2, 1, 0]), (m1, [0, 1, 5, 1, 0]))):
            want = m.copy(); want[np.arange(5), path] -= 1.0
            assert np.abs(g[:, n, :] - want).max() < 2e-5, (engine, n)
    nat.call("ocr_set_ctc_engine", 1)


def test_warpctc_abi_compute_ctc_loss(dev):
    """warp-ctc's own entry points (include/warpctc_abi.h: host labels / lengths / costs, ctcOptions by value) — the call
    warpctc_tensorflow.ctc makes (network.py:653-654): published known-answer vector, a ragged batch against the fp64 oracle
    and against ocr_ctc_loss, a non-zero blank, the autograd twin, and the loud refusal of CTC_CPU."""
    from lstm_ctc_ocr_amd import warpctc
    from lstm_ctc_ocr_amd._native import NativeError
    acts = torch.tensor([[[0.1, 0.6, 0.1, 0.1, 0.1]], [[0.1, 0.1, 0.6, 0.1, 0.1]]], device=dev)
    costs, grads = warpctc.compute(acts, [1, 2], [2], [2])
    assert abs(float(costs[0]) - 2.46286) < 1e-5
    exp = np.full((2, 5), 0.177031); exp[0, 1] = exp[1, 2] = -0.708125
    assert np.abs(grads.cpu().numpy().reshape(2, 5) - exp).max() < 1e-5
    costs0, none = warpctc.compute(acts, [1, 2], [2], [2], want_grad=False)           # gradients == NULL: score only
    assert none is None and abs(float(costs0[0]) - 2.46286) < 1e-5

    rng = np.random.RandomState(11)
    T, N, C = 40, 6, 20
    for blank in (0, C - 1):
        a = torch.from_numpy(rng.randn(T, N, C).astype(np.float32) * 2).to(dev)
        ll = np.array([5, 1, 9, 0, 12, 30], np.int32)                    # the last one is infeasible at its input length
        il = np.array([40, 7, 33, 12, 25, 20], np.int32)
        lo = 1 if blank == 0 else 0
        fl = np.concatenate([rng.randint(lo, lo + C - 1, n) for n in ll]).astype(np.int32)
        ref_c, ref_g = octc.ctc_loss_c(a.cpu().numpy(), fl, ll, il, blank)
        costs, grads = warpctc.compute(a, fl, ll, il, blank_label=blank)
        assert np.allclose(costs.numpy(), ref_c, rtol=1e-4, atol=1e-4)
        assert np.abs(grads.cpu().numpy() - ref_g).max() < 5e-4
        assert float(costs[5]) == 0.0 and float(grads[:, 5].abs().max()) == 0.0
        # differentiable twin: d(sum_n w_n cost_n)/d activations = w_n * gradient_n
        a2 = a.clone().requires_grad_(True)
        w = torch.arange(1, N + 1, dtype=torch.float32, device=dev)
        (warpctc.ctc(a2, fl, ll, il, blank_label=blank) * w).sum().backward()
        assert np.abs(a2.grad.cpu().numpy() - ref_g * np.arange(1, N + 1)[None, :, None]).max() < 3e-3
    with pytest.raises(NativeError):
        warpctc.compute(acts, [1, 2], [2], [2], loc=warpctc.CTC_CPU)


def test_ctc_c2_shape(dev):
    rng = np.random.RandomState(0)
    _ctc_case(dev, 63, 64, 64, [10] * 64, [63] * 64, 1)
    _ctc_case(dev, 63, 64, 64, rng.randint(4, 11, 64).tolist(), rng.randint(30, 64, 64).tolist(), 2)


def test_ctc_edge_cases(dev):
    # repeated characters, infeasible samples (L + repeats > T), empty label, T = 1
    labels = [[5, 5, 5, 5], [1, 2, 3], [], [7], [9, 9]]
    _ctc_case(dev, 8, 5, 16, None, [8, 2, 8, 1, 3], 3, labels=labels)
    _ctc_case(dev, 8, 5, 16, None, [7, 3, 5, 1, 2], 4, labels=labels)
    # long labels: S = 2L+1 > 64 exercises two / four slots per lane; large alphabet exercises strided classes
    _ctc_case(dev, 90, 3, 96, [40, 33, 5], [90, 80, 90], 5)
    _ctc_case(dev, 200, 2, 200, [100, 70], [200, 199], 6)
    # T = 79 (the longest plan of the variable-width workload): the fast kernel's tables need more than 64 KiB of LDS
    assert ops.ctc_train_supported(64, 79, 10)
    _ctc_case(dev, 79, 6, 64, [10, 4, 7, 10, 9, 5], [79, 60, 33, 79, 20, 12], 8)
    # non-zero blank
    _ctc_case(dev, 20, 4, 10, [3, 4, 5, 2], [20, 18, 15, 9], 7, blank=9,
              labels=[[1, 2, 3], [0, 0, 4, 8], [5, 6, 7, 8, 0], [2, 2]])


def test_ctc_greedy(dev):
    rng = np.random.RandomState(5)
    for (T, N, C) in [(63, 64, 64), (130, 7, 96), (5, 3, 4)]:
        acts = rng.randn(T, N, C).astype(np.float32)
        acts[rng.rand(T, N) < 0.4, 0] += 6.0      # plenty of blanks and repeats
        acts[:, :, 3] += (rng.rand(T, N) < 0.3) * 6.0
        il = rng.randint(1, T + 1, N).astype(np.int32)
        out, lens = ops.ctc_greedy_decode(torch.from_numpy(acts).to(dev), torch.from_numpy(il).to(dev))
        ref = odec.greedy_decode(acts, il)
        out = out.cpu().numpy(); lens = lens.cpu().numpy()
        for n in range(N):
            assert lens[n] == len(ref[n])
            assert out[n, :lens[n]].tolist() == ref[n]
            assert (out[n, lens[n]:] == 0).all()
    # hand cases from SURVEY §8c(5)
    am = [5, 5, 0, 5, 0, 0, 7]
    acts = np.full((7, 2, 8), -5.0, np.float32)
    for t, a in enumerate(am): acts[t, 0, a] = 5.0
    acts[:, 1, 0] = 5.0
    out, lens = ops.ctc_greedy_decode(torch.from_numpy(acts).to(dev), torch.tensor([7, 7], dtype=torch.int32, device=dev))
    assert out[0, :3].tolist() == [5, 5, 7] and int(lens[0]) == 3 and int(lens[1]) == 0


@pytest.mark.parametrize("T,N,C,beam", [(12, 6, 8, 100), (20, 5, 16, 4), (63, 8, 64, 100), (30, 3, 96, 25), (7, 4, 5, 2)])
def test_ctc_beam_search_matches_tf_semantics_oracle(dev, T, N, C, beam):
    rng = np.random.RandomState(T + C)
    acts = (rng.randn(T, N, C) * 3).astype(np.float32)
    acts[rng.rand(T, N) < 0.3, C - 1] += 5.0                    # TF blank (C-1) frames
    acts[rng.rand(T, N) < 0.3, 0] += 5.0                        # class-0 frames (the loss's blank, a normal symbol here)
    il = rng.randint(max(1, T // 2), T + 1, N).astype(np.int32)
    for merge in (True, False):
        out, lens, nlp = ops.ctc_beam_decode(torch.from_numpy(acts).to(dev), torch.from_numpy(il).to(dev), beam_width=beam,
                                             merge_repeated=merge)
        ref, scores = odec.beam_search_tf(acts, il, beam_width=beam, merge_repeated=merge)
        out, lens, nlp = out.cpu().numpy(), lens.cpu().numpy(), nlp.cpu().numpy()
        for n in range(N):
            assert out[n, :lens[n]].tolist() == ref[n], (n, out[n, :lens[n]].tolist(), ref[n])
            assert (out[n, lens[n]:] == 0).all()
            assert abs(-nlp[n] - scores[n]) < 1e-3 * max(1.0, abs(scores[n]))


# ------------------------------------------------------------------------------------------- GEMM NT
@pytest.mark.parametrize("M,N,K", [(4032, 1024, 512), (4032, 64, 512), (300, 132, 72), (128, 128, 32), (70, 520, 2048)])
def test_gemm_nt(dev, M, N, K):
    P = bf(gen((M, K), 1)); Q = bf(gen((N, K), 2)); bias = gen((N,), 3)
    ref = P @ Q.t()
    Pd, Qd, bd = P.to(dev).to(BF), Q.to(dev).to(BF), bias.to(dev)
    out = ops.gemm_nt(Pd, Qd, out_f32=True)
    assert relerr(out.cpu(), ref) < 2e-5
    out = ops.gemm_nt(Pd, Qd, bias=bd, relu=True)
    assert relerr(out.float().cpu(), bf(torch.relu(ref + bias))) < 1e-2
    # split-K with atomics accumulates onto the existing fp32 content
    base = gen((M, N), 4).to(dev)
    out = ops.gemm_nt(Pd, Qd, out=base.clone(), splits=3, bias=bd)
    assert relerr(out.cpu(), ref + bias + base.cpu()) < 2e-5
    # mask (ReLU backward fused) and accumulate
    mask = gen((M, N), 5)
    out = ops.gemm_nt(Pd, Qd, mask=mask.to(dev).to(BF))
    assert relerr(out.float().cpu(), bf(ref * (bf(mask) > 0))) < 1e-2
    acc = ops.gemm_nt(Pd, Qd, out=base.clone(), accumulate=True)
    assert relerr(acc.cpu(), ref + base.cpu()) < 2e-5


def test_gemm_nt_rowswap_and_rowgroups(dev):
    Nb, T, K, N = 5, 7, 64, 64
    P = bf(gen((Nb * T, K), 1)); Q = bf(gen((N, K), 2))
    out = torch.empty((T * Nb, N), dtype=torch.float32, device=dev)
    ops.gemm_nt(P.to(dev).to(BF), Q.to(dev).to(BF), out=out, rowswap=(T, Nb))
    ref = (P @ Q.t()).reshape(Nb, T, N).permute(1, 0, 2).reshape(T * Nb, N)
    assert relerr(out.cpu(), ref) < 2e-5
    # conv5-style overlapping rows: x [Nb, W, HC], window = rows w and w+1 -> K = 2*HC, W-1 rows per sample
    Nb, W, HC, Co = 3, 9, 64, 128
    x = bf(gen((Nb, W, HC), 3)); Wt = bf(gen((Co, 2 * HC), 4))
    rows = torch.stack([torch.cat([x[n, w], x[n, w + 1]]) for n in range(Nb) for w in range(W - 1)])
    out = ops.gemm_nt(x.to(dev).to(BF), Wt.to(dev).to(BF), M=Nb * (W - 1), N=Co, K=2 * HC, ldp=HC, row_group=W - 1,
                      row_skip=1, out_f32=True)
    assert relerr(out.cpu(), rows @ Wt.t()) < 2e-5


# ------------------------------------------------------------------------------------------- conv 3x3
def _conv_ref(x, w, b):   # x [N,W,H,C], w HWIO
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1)
    return y + b


@pytest.mark.parametrize("Nb,W,H,Ci,Co", [(4, 16, 8, 64, 128), (2, 12, 4, 256, 512), (64, 64, 4, 256, 512), (3, 20, 16, 64, 128),
                                          (16, 32, 16, 64, 128), (64, 128, 8, 64, 256), (5, 52, 4, 128, 192), (32, 64, 4, 512, 512),
                                          (7, 22, 8, 128, 256), (32, 64, 2, 512, 512), (3, 18, 2, 64, 128), (9, 64, 2, 128, 64),
                                          (17, 62, 4, 128, 128), (9, 30, 16, 64, 256),       # ragged last tiles of the 256- / 128-pixel kernels
                                          (4, 128, 8, 128, 128), (6, 192, 4, 192, 64),       # plane-layout kernel: several tiles per image, 3 chunks
                                          (8, 48, 16, 128, 64), (16, 32, 16, 128, 128), (8, 64, 16, 64, 64),    # ... at H = 16 (one / two halo buffers)
                                          (32, 40, 4, 128, 128), (32, 24, 8, 64, 128), (16, 50, 8, 128, 64),    # ... tiles crossing image boundaries (general width)
                                          # weight-stationary persistent kernel (conv_ws, round 5): small grids (fewer tiles than workgroups, three channel
                                          # tiles = no XCD map, several images per workgroup run); taken by default only from two tiles per CU, forced with OCR_CONV_WS=2
                                          (2, 32, 16, 64, 128), (3, 24, 16, 128, 192), (5, 48, 8, 128, 64), (40, 64, 16, 64, 64),
                                          # the EXACT shapes of the benchmarked step (BASELINE configs[1], N = 64, W = 256): conv2, conv3_1, conv3_2, conv4_2
                                          # (conv4_1 is (64, 64, 4, 256, 512) above) — the dispatcher's full-chip tiles / 64-split slabs only exist at this size
                                          (64, 128, 16, 64, 128), (64, 64, 8, 128, 256), (64, 64, 8, 256, 256), (64, 64, 4, 512, 512),
                                          # ... and the extremes of configs[3] (W = 80 and W = 320 padded batches)
                                          (64, 40, 16, 64, 128), (64, 20, 4, 512, 512), (64, 80, 8, 256, 256), (64, 80, 4, 256, 512),
                                          # ... and widths that are no multiple of the weight-gradient kernel's step (round 6: wgrad9p's zero-row instances — every
                                          # step of (32, 33, 4) / (16, 17, 8) crosses an image boundary at another column; W = 79 = a 316-pixel configs[3] batch)
                                          (32, 33, 4, 64, 64), (16, 17, 8, 64, 64), (64, 79, 4, 256, 512), (64, 79, 8, 256, 256), (32, 47, 4, 128, 64)])
def test_conv3x3_fwd_dgrad_wgrad(dev, Nb, W, H, Ci, Co):
    x = bf(gen((Nb, W, H, Ci), 1)); w = bf(gen((3, 3, Ci, Co), 2, 0.05)); b = gen((Co,), 3)
    ref = _conv_ref(x, w, b)
    wpack = torch.empty((Co, 3, 3, Ci), dtype=BF, device=dev)
    ops.pack_transpose(w.reshape(9 * Ci, Co).to(dev), wpack)          # [9Ci][Co] -> [Co][9Ci]
    y = ops.conv3x3(x.to(dev).to(BF), wpack, bias=b.to(dev), relu=True)
    assert relerr(y.float().cpu(), bf(torch.relu(ref))) < 1e-2
    # data gradient == conv with flipped / transposed weights, with the ReLU mask of the layer below fused
    dy = bf(gen((Nb, W, H, Co), 4))
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    _conv_ref(xr, wr, b).backward(dy)
    wd = torch.empty((Ci, 3, 3, Co), dtype=BF, device=dev)
    ops.pack_conv_dgrad(w.to(dev), wd)
    below = gen((Nb, W, H, Ci), 5)
    dx = ops.conv3x3(dy.to(dev).to(BF), wd, mask=below.to(dev).to(BF))
    assert relerr(dx.float().cpu(), bf(xr.grad * (bf(below) > 0))) < 1e-2
    # weight gradient accumulates into the fp32 TF-layout buffer
    dw = torch.zeros((3, 3, Ci, Co), dtype=torch.float32, device=dev)
    dbias = torch.zeros(Co, dtype=torch.float32, device=dev)
    ops.conv3x3_wgrad(x.to(dev).to(BF), dy.to(dev).to(BF), dw, dbias=dbias)
    assert relerr(dw.cpu(), wr.grad) < 1e-4
    assert relerr(dbias.cpu(), dy.reshape(-1, Co).sum(0)) < 1e-4          # bias gradient fused into the same pass
    ops.conv3x3_wgrad(x.to(dev).to(BF), dy.to(dev).to(BF), dw, splits=2, dbias=dbias)
    assert relerr(dw.cpu(), 2 * wr.grad) < 1e-4 and relerr(dbias.cpu(), 2 * dy.reshape(-1, Co).sum(0)) < 1e-4
    # nine-tap slab kernel (workspace form): same contract, no atomics -> bit-identical from run to run
    nbytes = ops.conv3x3_wgrad_workspace_bytes(Nb, W, H, Ci, Co)
    print('wgrad workspace bytes', nbytes)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    ws.fill_(0x7f)                                                    # scratch is neither zeroed nor kept: poison it
    runs = []
    for _ in range(2):
        dw2 = torch.zeros((3, 3, Ci, Co), dtype=torch.float32, device=dev)
        db2 = torch.zeros(Co, dtype=torch.float32, device=dev)
        ops.conv3x3_wgrad(x.to(dev).to(BF), dy.to(dev).to(BF), dw2, dbias=db2, workspace=ws)
        runs.append((dw2.cpu(), db2.cpu()))
    assert relerr(runs[0][0], wr.grad) < 1e-4, relerr(runs[0][0], wr.grad)
    assert relerr(runs[0][1], dy.reshape(-1, Co).sum(0)) < 1e-4
    if nbytes:
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    dw2 = torch.ones((3, 3, Ci, Co), dtype=torch.float32, device=dev)        # "+=" semantics like the atomics path
    ops.conv3x3_wgrad(x.to(dev).to(BF), dy.to(dev).to(BF), dw2, workspace=ws)
    assert relerr(dw2.cpu() - 1.0, wr.grad) < 1e-4


_CONV_GENERATION_ENVS = [dict(OCR_CONV_K2='1', OCR_K2_CFG='A'), dict(OCR_CONV_K2='1', OCR_K2_CFG='D'),
                         dict(OCR_CONV_K2='1', OCR_K2_CFG='A', OCR_CONV_K3='0'), dict(OCR_CONV_K2='1', OCR_K2_CFG='D', OCR_CONV_K3='0'),
                         dict(OCR_CONV_K2='0'), dict(OCR_CONV_WS='2'), dict(OCR_CONV_WS='0')]


@pytest.fixture(scope='module')
def conv_generation_runs():
    """The seven knob settings below, each a pytest subprocess over the convolution parity tests (the knobs are read once per process) — started
    together, four at a time: one after the other they were 200 s of the GPU suite, most of it the CPU references of a single process."""
    import concurrent.futures
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env):
        # conv_ws forced onto every shape it covers changes WHICH kernel computes the unfused side of the "write-out fusion == separate passes,
        # bit for bit" tests (those hold within one kernel family: same fp32 summation order): it runs the parity and fused-pool tests
        sel = 'test_conv3x3_fwd_dgrad_wgrad or test_conv3x3_relu_pool' if env.get('OCR_CONV_WS') == '2' else 'test_conv3x3'
        try:
            out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-q', '-k', sel],
                                 env=dict(os.environ, OMP_NUM_THREADS='4', MKL_NUM_THREADS='4', **env), capture_output=True, text=True, timeout=1200, cwd=root)
            return out.returncode, out.stdout[-3000:]
        except subprocess.TimeoutExpired as e:
            return -1, 'timed out: %s' % e
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        return list(ex.map(run, _CONV_GENERATION_ENVS))


@pytest.mark.parametrize("env", range(len(_CONV_GENERATION_ENVS)), ids=['env%d' % i for i in range(len(_CONV_GENERATION_ENVS))])
def test_conv_kernel_generations_through_the_convolution_parity_tests(dev, conv_generation_runs, env):
    """conv_k2.hip / conv_k3.hip (in-workgroup K split; tiles A 256 x 128 and D 256 x 64 pixels x channels; k3 = the plane layout of the
    halo, taken where it covers the shape) with each tile forced onto every shape it covers, conv_k2 alone (OCR_CONV_K3=0), and the
    conv_halo.hip kernels alone, through the same parity / fused-pool / accumulate tests (the knobs are read once per process; by default
    the dispatcher mixes the kernels per layer).  OCR_CONV_WS=2: the weight-stationary persistent kernel (conv_ws.hip) on EVERY shape it
    covers, whatever the grid; =0: none (the shapes it takes by default stay tested on the plane-layout kernels)."""
    rc, tail = conv_generation_runs[env]
    assert rc == 0, (_CONV_GENERATION_ENVS[env], tail)


@pytest.mark.parametrize("Mk,I,J", [(4032, 512, 2048), (4032, 256, 1024), (4032, 512, 64), (100, 72, 136), (300, 128, 128), (1000, 256, 384)])
def test_gemm_tn(dev, Mk, I, J):
    A = bf(gen((Mk, I), 1)); B = bf(gen((Mk, J), 2))
    out = torch.zeros((I, J), dtype=torch.float32, device=dev)
    cs = torch.zeros(J, dtype=torch.float32, device=dev)
    ops.gemm_tn(A.to(dev).to(BF), B.to(dev).to(BF), out, colsum=cs)
    assert relerr(out.cpu(), A.t() @ B) < 1e-4 and relerr(cs.cpu(), B.sum(0)) < 1e-4
    ops.gemm_tn(A.to(dev).to(BF), B.to(dev).to(BF), out, scale=0.5, splits=1)
    assert relerr(out.cpu(), 1.5 * (A.t() @ B)) < 1e-4


def test_gemm_tn_batched_and_xh(dev):
    """Two weight-gradient products in one launch (the BiLSTM directions) and the [x | h_prev] operand builder."""
    Nb, T, D, U = 8, 21, 512, 256
    R = Nb * T
    rng = np.random.RandomState(4)
    x = bf(torch.from_numpy(rng.randn(R, D).astype(np.float32)))
    hout = bf(torch.from_numpy(rng.randn(R, 2 * U).astype(np.float32)))
    lens = np.array([21, 5, 1, 21, 13, 7, 21, 2], np.int32)
    xh = torch.empty(2, R, D + U, dtype=BF, device=dev)
    ops.lstm_xh(x.to(dev).to(BF), hout.to(dev).to(BF), torch.from_numpy(lens).to(dev), xh, Nb, T, D, U)
    ref = torch.zeros(2, R, D + U)
    for d in range(2):
        ref[d, :, :D] = x
        for n in range(Nb):
            for t in range(int(lens[n])):
                tp = t - 1 if d == 0 else t + 1
                if 0 <= tp < lens[n]:
                    ref[d, n * T + t, D:] = hout[n * T + tp, d * U:(d + 1) * U]
    assert torch.equal(xh.float().cpu(), ref)
    dz = bf(torch.from_numpy((rng.randn(R, 8 * U) * 0.1).astype(np.float32)))
    gap = 1024 + 64                                                   # the two outputs are not adjacent in the flat buffer
    out = torch.zeros(2 * ((D + U) * 4 * U) + gap, device=dev)
    cs = torch.zeros(2 * 4 * U + 128, device=dev)
    so, sc = (D + U) * 4 * U + gap, 4 * U + 128
    ops.gemm_tn_batched(xh, D + U, R * (D + U), dz.to(dev).to(BF), 8 * U, 4 * U, out, 4 * U, so, R, D + U, 4 * U, 2, colsum=cs, strideColsum=sc)
    for d in range(2):
        want = ref[d].double().t() @ dz[:, d * 4 * U:(d + 1) * 4 * U].double()
        got = out[d * so:d * so + (D + U) * 4 * U].view(D + U, 4 * U).cpu().double()
        assert relerr(got, want) < 1e-4
        assert relerr(cs[d * sc:d * sc + 4 * U].cpu().double(), dz[:, d * 4 * U:(d + 1) * 4 * U].double().sum(0)) < 1e-4
    assert float(out[(D + U) * 4 * U:(D + U) * 4 * U + gap].abs().max()) == 0.0


@pytest.mark.parametrize("Nb,W,HC,Co,R2,I2,J2", [(64, 64, 1024, 512, 4032, 768, 1024),       # the headline step's two products: conv5 + both BiLSTM cells
                                                 (13, 21, 128, 128, 1000, 128, 256),          # contraction not a multiple of 64 (zero-page tail), tiles % 8 != 0
                                                 (3, 90, 64, 256, 300, 256, 128)])            # row groups of 89 rows: several 64-row stages per group
def test_gemm_tn_jobs(dev, Nb, W, HC, Co, R2, I2, J2):
    """csrc/gemm_tn3.hip (round 4): two plain weight-gradient products in ONE launch, one workgroup per 128 x 128 tile over the whole
    contraction (ping-pong K halves, no atomics): conv5-style overlapping rows (row_group W - 1, skip 1) with the bias column sums, and a
    batch of two products with strides — against fp64 references, accumulating into non-zero outputs, bit-identical from run to run."""
    rng = np.random.RandomState(7)
    x = bf(torch.from_numpy(rng.randn(Nb, W, HC).astype(np.float32)))
    M1 = Nb * (W - 1)
    dy = bf(torch.from_numpy((rng.randn(M1, Co) * 0.1).astype(np.float32)))
    rows = torch.stack([torch.cat([x[n, w], x[n, w + 1]]) for n in range(Nb) for w in range(W - 1)])
    xh = bf(torch.from_numpy(rng.randn(2, R2, I2).astype(np.float32)))
    dz = bf(torch.from_numpy((rng.randn(R2, 2 * J2) * 0.1).astype(np.float32)))
    xd, dyd, xhd, dzd = x.to(dev).to(BF), dy.to(dev).to(BF), xh.to(dev).to(BF), dz.to(dev).to(BF)
    gap = 256
    runs = []
    for _ in range(2):
        dw1 = torch.full((2 * HC, Co), 0.5, device=dev); db1 = torch.full((Co,), 0.25, device=dev)
        dw2 = torch.ones(2 * I2 * J2 + gap, device=dev); db2 = torch.ones(2 * J2 + 128, device=dev)
        j1 = ops.tn_job(xd, HC, dyd, Co, dw1, Co, M1, 2 * HC, Co, row_group=W - 1, row_skip=1, colsum=db1)
        j2 = ops.tn_job(xhd, I2, dzd, 2 * J2, dw2, J2, R2, I2, J2, nbatch=2, strideA=R2 * I2, strideB=J2, strideOut=I2 * J2 + gap,
                        colsum=db2, strideColsum=J2 + 128)
        assert ops.gemm_tn_jobs_supported([j1, j2])
        ops.gemm_tn_jobs([j1, j2])
        runs.append((dw1.cpu(), db1.cpu(), dw2.cpu(), db2.cpu()))
    dw1, db1, dw2, db2 = runs[0]
    assert relerr(dw1.double() - 0.5, rows.double().t() @ dy.double()) < 1e-4
    assert relerr(db1.double() - 0.25, dy.double().sum(0)) < 1e-4
    so = I2 * J2 + gap
    for d in range(2):
        want = xh[d].double().t() @ dz[:, d * J2:(d + 1) * J2].double()
        assert relerr(dw2[d * so:d * so + I2 * J2].view(I2, J2).double() - 1.0, want) < 1e-4
        assert relerr(db2[d * (J2 + 128):d * (J2 + 128) + J2].double() - 1.0, dz[:, d * J2:(d + 1) * J2].double().sum(0)) < 1e-4
    assert float((dw2[I2 * J2:I2 * J2 + gap] - 1.0).abs().max()) == 0.0 and float((db2[J2:J2 + 128] - 1.0).abs().max()) == 0.0
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))                    # no atomics: bit-reproducible
    # one job alone, and what is not covered
    dw1 = torch.zeros((2 * HC, Co), device=dev)
    ops.gemm_tn_jobs([ops.tn_job(xd, HC, dyd, Co, dw1, Co, M1, 2 * HC, Co, row_group=W - 1, row_skip=1)])
    assert relerr(dw1.cpu().double(), rows.double().t() @ dy.double()) < 1e-4
    assert not ops.gemm_tn_jobs_supported([ops.tn_job(xd, HC, dyd, Co, dw1, Co, M1, 2 * HC, 64)])           # J % 128 != 0
    assert not ops.gemm_tn_jobs_supported([ops.tn_job(xd, HC, dyd, Co, dw1, Co, 100, 2 * HC, Co)])          # fewer than 256 rows


def test_gemm_tn_conv5_rows(dev):
    Nb, W, HC, Co = 3, 9, 64, 128
    x = bf(gen((Nb, W, HC), 3)); dy = bf(gen((Nb * (W - 1), Co), 4))
    rows = torch.stack([torch.cat([x[n, w], x[n, w + 1]]) for n in range(Nb) for w in range(W - 1)])
    out = torch.zeros((2 * HC, Co), dtype=torch.float32, device=dev)
    ops.gemm_tn(x.to(dev).to(BF), dy.to(dev).to(BF), out, Mk=Nb * (W - 1), I=2 * HC, J=Co, lda=HC, row_group=W - 1, row_skip=1)
    assert relerr(out.cpu(), rows.t() @ dy) < 1e-4


# ------------------------------------------------------------------------------------------- conv1 / pool / bn
def test_conv1(dev):
    Nb, W, H, Co = 5, 24, 32, 64
    x = gen((Nb, W, H), 1).abs(); w = gen((3, 3, 1, Co), 2, 0.3); b = gen((Co,), 3, 0.1)
    ref = torch.relu(_conv_ref(x.unsqueeze(3), w, b))
    y = ops.conv1_fwd(x.to(dev), w.to(dev), b.to(dev))
    assert relerr(y.float().cpu(), bf(ref)) < 1e-2
    dz = bf(gen((Nb, W, H, Co), 4))
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    _conv_ref(x.unsqueeze(3), wr, br).backward(dz)
    dw = torch.zeros_like(w, device=dev); db = torch.zeros(Co, device=dev)
    ops.conv1_wgrad(x.to(dev), dz.to(dev).to(BF), dw, db)
    assert relerr(dw.cpu(), wr.grad) < 1e-4 and relerr(db.cpu(), br.grad) < 1e-4


@pytest.mark.parametrize("Nb,W,H", [(5, 24, 32), (40, 250, 32), (3, 30, 12)])
def test_conv1_pool_fused_equals_unfused(dev, Nb, W, H):
    _conv1_pool_fused_equals_unfused(dev, Nb, W, H)


def test_conv1_pool_second_generation_kernels(dev):
    """OCR_CONV1_V2=1 (measured slightly slower, rejected) exists only in the experiments flavour of the library (`make EXPERIMENTS=1` ->
    libocrhip_exp.so): the product build neither compiles those kernels nor reads the knob (run against the product library this
    test re-tested the first generation).  The knob is read once per process, so the variant runs in a child."""
    import os, subprocess, sys
    from lstm_ctc_ocr_amd import _native as nat
    exp = os.path.join(os.path.dirname(nat.LIB_PATH), 'libocrhip_exp.so')
    if not os.path.exists(exp):
        pytest.skip('experiments flavour not built (make -C lstm_ctc_ocr_amd/csrc EXPERIMENTS=1)')
    import ctypes
    try:
        fn = ctypes.CDLL(exp).ocr_build_id
        fn.restype = ctypes.c_char_p
        have = fn().decode()
    except (OSError, AttributeError):
        have = None
    if have != nat.source_build_id(experiments=True):
        pytest.skip('libocrhip_exp.so is stale (%s, tree %s): rebuild it with make EXPERIMENTS=1' % (have, nat.source_build_id(experiments=True)))
    env = dict(os.environ, OCR_CONV1_V2='1', OCR_NATIVE_LIB=exp)
    code = ("import sys; sys.path.insert(0, %r); import torch; from tests import test_gpu_kernels as t; "
            "[t._conv1_pool_fused_equals_unfused(torch.device('cuda', 0), *s) for s in ((5, 24, 32), (40, 250, 32), (3, 30, 12))]; print('V2_OK')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'V2_OK' in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("Nb,W,H", [(5, 24, 32), (12, 64, 32)])
def test_conv1_pool_train_kernels_against_torch(dev, Nb, W, H):
    """The step's own conv1 launches — ocr_conv1_pool_fwd_train (pooled map + routing codes) and ocr_conv1_pool_bwd_slab (per-block partial
    sums) — DIRECTLY against torch-CPU fp32 (LSTM_train.py:24-25: conv 3x3 SAME + ReLU, 2 x 2 max-pool; tf.gradients of both): the other
    conv1 + pool tests compare device kernels with each other.  The reference rounds the activation to bf16 where the device stores it
    (the pool routes on the bf16 values, first maximum wins: torch's max_pool2d scans the window in the same order — test_maxpool)."""
    Co = 64
    # operands on a 1/64 grid: every product is a multiple of 2^-12 and every 9-term sum is exact in fp32 whatever the order, so device and
    # torch round the SAME value to bf16 and the pool routes identically — what is left in the backward sums is fp32 summation order
    q = lambda t: torch.round(t * 64) / 64
    x = q(gen((Nb, W, H), 1).abs()); w = q(gen((3, 3, 1, Co), 2, 0.3)); b = q(gen((Co,), 3, 0.1))
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    z = _conv_ref(x.unsqueeze(3), wr, br)                                     # fp32 [Nb, W, H, Co]
    yq = bf(torch.relu(z.detach())).requires_grad_(True)                      # what conv1 stores
    p_ref = F.max_pool2d(yq.permute(0, 3, 1, 2), (2, 2), (2, 2)).permute(0, 2, 3, 1)
    codes = torch.zeros((Nb * (W // 2) * (H // 2), 8), dtype=torch.int32, device=dev)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    p = ops.conv1_pool_fwd(xd, wd, bd, codes=codes)
    assert maxerr(p.float().cpu(), p_ref.detach()) == 0.0
    dp = bf(gen(tuple(p.shape), 4))
    p_ref.backward(dp)
    dz = yq.grad * (yq.detach() > 0)                                          # routed through the pool, masked by the ReLU
    z.backward(dz)
    rows = ops.conv1_pool_bwd_slab_rows(Nb, W, H)
    for cd in (codes, None):                                                  # routing from the saved codes / from recomputed windows
        slab = torch.full((rows, 640), float('nan'), device=dev)
        ops.conv1_pool_bwd_slab(xd, wd, bd, dp.to(dev).to(BF), slab, codes=cd)
        tot = slab.double().sum(0).cpu()
        assert relerr(tot[:576].view(3, 3, 1, Co), wr.grad.double()) < 1e-4 and relerr(tot[576:], br.grad.double()) < 1e-4


def _conv1_pool_fused_equals_unfused(dev, Nb, W, H):
    """(40, 250, 32): more pooled pixels than one sweep of the forward grid, so the kernels' next-iteration prefetch runs; W / 2 = 125
    and H / 2 = 6 are not powers of two (32-bit index arithmetic)."""
    Co = 64
    x = gen((Nb, W, H), 1).abs().to(dev); w = gen((3, 3, 1, Co), 2, 0.3).to(dev); b = gen((Co,), 3, 0.1).to(dev)
    y = ops.conv1_fwd(x, w, b)
    p_ref = ops.maxpool_fwd(y, 2, 2)
    p = ops.conv1_pool_fwd(x, w, b)
    assert maxerr(p.float().cpu(), p_ref.float().cpu()) == 0.0                     # bit-identical pooled map
    dp = bf(gen(tuple(p.shape), 4)).to(dev).to(BF)
    dz = ops.maxpool_bwd(y, dp, 2, 2, relu_mask=True)
    dw_ref = torch.zeros_like(w); db_ref = torch.zeros(Co, device=dev)
    ops.conv1_wgrad(x, dz, dw_ref, db_ref)
    dw = torch.zeros_like(w); db = torch.zeros(Co, device=dev)
    ops.conv1_pool_bwd(x, w, b, dp, dw, db)
    assert relerr(dw.cpu(), dw_ref.cpu()) < 1e-5 and relerr(db.cpu(), db_ref.cpu()) < 1e-5
    # round 4: the training forward also saves the pool routing + ReLU bits (4 bits per pooled output) and clears a buffer on the way; the
    # backward pass that consumes the codes instead of recomputing the 2 x 2 windows sums the same terms in the same order
    codes = torch.full((Nb * (W // 2) * (H // 2), 8), -1, dtype=torch.int32, device=dev)
    junk = torch.full((4096 + 8,), 7.0, device=dev)
    ones = torch.zeros(4096 + 64, dtype=torch.int32, device=dev)
    p2 = ops.conv1_pool_fwd(x, w, b, zero=junk[:4096], codes=codes, ones=ones[:4096])
    assert bool((ones[:4096] == -1).all()) and bool((ones[4096:] == 0).all())                 # the all-ones fill: exactly the words it was given
    assert torch.equal(p2, p) and float(junk[:4096].abs().max()) == 0.0 and float(junk[4096:].min()) == 7.0
    dw2 = torch.zeros_like(w); db2 = torch.zeros(Co, device=dev)
    ops.conv1_pool_bwd(x, w, b, dp, dw2, db2, codes=codes)
    assert relerr(dw2.cpu(), dw.cpu()) < 5e-6 and relerr(db2.cpu(), db.cpu()) < 5e-6          # (atomics: block order differs from run to run; 1.2e-6 seen once in round 5)
    # slab form: per-block partial sums instead of atomics (the engine adds the rows with the merged slab reduction): bit-reproducible
    rows = ops.conv1_pool_bwd_slab_rows(Nb, W, H)
    slabs = []
    for cd in (codes, None, codes):
        slab = torch.full((rows, 640), float('nan'), device=dev)
        ops.conv1_pool_bwd_slab(x, w, b, dp, slab, codes=cd)
        slabs.append(slab.clone())
    assert torch.equal(slabs[0], slabs[2]) and torch.equal(slabs[0], slabs[1])                # run to run, and codes against recomputation
    tot = slabs[0].double().sum(0).cpu()
    assert relerr(tot[:576].view(3, 3, 1, Co), dw.double().cpu()) < 1e-6 and relerr(tot[576:], db.double().cpu()) < 1e-6


@pytest.mark.parametrize("kw,kh", [(2, 2), (1, 2)])
def test_maxpool(dev, kw, kh):
    Nb, W, H, C = 3, 8, 8, 64
    x = bf(gen((Nb, W, H, C), 1))
    x[x.abs() < 0.3] = 0.0                    # ties (ReLU zeros) exercise first-max routing
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr.permute(0, 3, 1, 2), (kw, kh), (kw, kh)).permute(0, 2, 3, 1)
    y = ops.maxpool_fwd(x.to(dev).to(BF), kw, kh)
    assert maxerr(y.float().cpu(), yr.detach()) == 0.0
    dy = bf(gen(tuple(yr.shape), 2))
    yr.backward(dy)
    dx = ops.maxpool_bwd(x.to(dev).to(BF), dy.to(dev).to(BF), kw, kh, relu_mask=False)
    assert maxerr(dx.float().cpu(), xr.grad) == 0.0
    dx = ops.maxpool_bwd(x.to(dev).to(BF), dy.to(dev).to(BF), kw, kh, relu_mask=True)
    assert maxerr(dx.float().cpu(), xr.grad * (x > 0)) == 0.0


@pytest.mark.parametrize("Nb,W,H,Ci,Co,kw,kh", [(64, 128, 16, 64, 128, 2, 2), (64, 64, 8, 256, 256, 1, 2), (64, 64, 4, 512, 512, 1, 2),
                                               (8, 64, 8, 64, 64, 2, 2), (16, 32, 4, 128, 192, 2, 2), (5, 26, 16, 64, 64, 1, 2),
                                               (3, 40, 10, 64, 128, 1, 2), (16, 64, 4, 128, 128, 2, 2), (16, 128, 8, 64, 128, 2, 2),
                                               (8, 32, 16, 64, 128, 2, 2), (8, 32, 16, 128, 64, 1, 2), (32, 40, 4, 64, 128, 2, 2),
                                               (32, 24, 8, 64, 64, 1, 2),
                                               # conv_ws instances (OCR_CONV_WS=2 forces them at these sizes): both pools on each
                                               (8, 32, 8, 128, 128, 1, 2), (8, 32, 8, 128, 64, 2, 2), (4, 32, 16, 64, 64, 1, 2), (4, 32, 16, 128, 64, 2, 2)])
def test_conv3x3_relu_pool_fused_equals_unfused(dev, Nb, W, H, Ci, Co, kw, kh):
    """conv + bias + ReLU with the following max-pool written by the same epilogue (LSTM_train.py:26-33): the full-resolution output
    and the pooled tensor are bit-identical to conv3x3 followed by maxpool_fwd."""
    if not ops.conv3x3_pool_supported(Nb, W, H, Ci, Co, kw, kh):
        pytest.skip("shape not covered by the fused epilogue")
    x = gen((Nb, W, H, Ci), 1).to(dev).to(BF)
    wp = (gen((Co, 3, 3, Ci), 2) * 0.05).to(dev).to(BF)
    b = gen((Co,), 3).to(dev)
    y0 = ops.conv3x3(x, wp, bias=b, relu=True)
    p0 = ops.maxpool_fwd(y0, kw, kh)
    y1 = torch.full_like(y0, 7.0)
    p1 = torch.full_like(p0, 7.0)
    ops.conv3x3_relu_pool(x, wp, y1, p1, b, kw, kh)
    assert torch.equal(y1, y0) and torch.equal(p1, p0)
    assert float(p0.float().abs().max()) > 0


def test_conv3x3_accumulate_epilogue(dev):
    """OCR_EPI_ACCUM on the data-gradient convolution: out += masked result (a tensor with several consumers collects its gradient
    without a scratch tensor + add pass) == bf16(out_old + fp32 result), and shapes outside the halo kernel refuse the flag."""
    from lstm_ctc_ocr_amd._native import NativeError
    Nb, W, H, Ci, Co = 8, 64, 8, 128, 64
    assert ops.conv3x3_accum_supported(Nb, W, H, Ci, Co)
    x = gen((Nb, W, H, Ci), 1).to(dev).to(BF); wp = (gen((Co, 3, 3, Ci), 2) * 0.05).to(dev).to(BF)
    mask = gen((Nb, W, H, Co), 3).to(dev).to(BF)
    old = gen((Nb, W, H, Co), 4).to(dev).to(BF)
    plain = ops.conv3x3(x, wp, mask=mask)
    out = old.clone()
    ops.conv3x3(x, wp, out=out, mask=mask, accumulate=True)
    want = (old.float() + plain.float())
    # the fused form rounds once (fp32 sum -> bf16), the two-pass form twice: at most one bf16 ulp apart
    assert float((out.float() - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -7
    assert float((out.float() - want.to(BF).float()).abs().mean()) < 1e-3
    assert not ops.conv3x3_accum_supported(2, 8, 8, 64, 64)                 # M < 1024: not the halo kernel
    xs = torch.zeros(2, 8, 8, 64, dtype=BF, device=dev); ws = torch.zeros(64, 3, 3, 64, dtype=BF, device=dev)
    with pytest.raises(NativeError):
        ops.conv3x3(xs, ws, out=torch.zeros(2, 8, 8, 64, dtype=BF, device=dev), accumulate=True)


def test_conv3x3_pool_fusion_refuses_uncovered_shapes(dev):
    from lstm_ctc_ocr_amd._native import NativeError
    assert not ops.conv3x3_pool_supported(4, 32, 16, 32, 64, 2, 2)          # C_in % 64
    assert not ops.conv3x3_pool_supported(4, 33, 16, 64, 64, 2, 2)          # odd W for a 2 x 2 window
    assert not ops.conv3x3_pool_supported(4, 32, 16, 64, 64, 2, 1)          # window along W only
    x = torch.zeros(4, 33, 16, 64, dtype=BF, device=dev); wp = torch.zeros(64, 3, 3, 64, dtype=BF, device=dev)
    with pytest.raises(NativeError):
        ops.conv3x3_relu_pool(x, wp, torch.empty_like(x), torch.empty(4, 16, 8, 64, dtype=BF, device=dev), torch.zeros(64, device=dev), 2, 2)


@pytest.mark.parametrize("M,C", [(16384, 512), (1000, 64), (4096, 512), (65536, 64), (8200, 128)])   # > 4 M elements: three launches; else two
def test_batchnorm(dev, M, C):
    x = bf(gen((M, C), 1) * 2 + 0.5); gamma = gen((C,), 2) + 1.5; beta = gen((C,), 3)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    mu = xr.mean(0); var = xr.var(0, unbiased=False)
    yr = torch.relu((xr - mu) * torch.rsqrt(var + 1e-3) * gr + br)
    ws = ops.bn_workspace(M, C, dev)
    xd = x.to(dev).to(BF)
    y, sm, sr = ops.bn_train_fwd(xd, gamma.to(dev), beta.to(dev), 1e-3, True, ws)
    assert relerr(sm.cpu(), mu.detach()) < 1e-4 and relerr(sr.cpu(), torch.rsqrt(var + 1e-3).detach()) < 1e-4
    assert relerr(y.float().cpu(), bf(yr.detach())) < 1e-2
    # residual-block tail in the apply pass: relu(bf16(bn) + res), bit-identical to batch norm (no relu), add, relu as separate passes
    res = bf(gen((M, C), 7)).to(dev).to(BF)
    y_plain, _, _ = ops.bn_train_fwd(xd, gamma.to(dev), beta.to(dev), 1e-3, False, ws)
    want = torch.empty_like(y_plain)
    ops.eltwise(3, y_plain, res, want)
    y_tail, _, _ = ops.bn_train_fwd(xd, gamma.to(dev), beta.to(dev), 1e-3, True, ws, residual=res)
    assert torch.equal(y_tail, want)
    dy = bf(gen((M, C), 4))
    # use the device's own (bf16) y for the mask so both sides agree on which outputs are exactly zero
    ydev = y.float().cpu()
    (yr * (ydev > 0)).backward(dy)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    dx = ops.bn_train_bwd(xd, y, dy.to(dev).to(BF), gamma.to(dev), sm, sr, dg, db, True, ws)
    assert relerr(dg.cpu(), gr.grad) < 1e-3 and relerr(db.cpu(), br.grad) < 1e-3
    assert relerr(dx.float().cpu(), bf(xr.grad)) < 2e-2


@pytest.mark.parametrize("M,C,relu", [(16384, 512, True), (4096, 256, True), (20480, 512, False), (1000, 64, True)])
def test_batchnorm_with_the_pool_behind_it(dev, M, C, relu):
    """Round 4: the 1 x 2 max-pool over row pairs that is a batch-norm layer's only consumer (LSTM_train.py:32-33) — written by the apply pass,
    and its gradient routed inside both backward passes: BIT-identical to batch norm + maxpool_fwd, and to maxpool_bwd + batch-norm backward."""
    x = bf(gen((M, C), 1) * 2 + 0.5).to(dev).to(BF)
    x[5] = x[4]                                                    # an exact tie inside a pair: the FIRST row of the pair wins
    gamma = (gen((C,), 2) + 1.5).to(dev); beta = gen((C,), 3).to(dev)
    ws = ops.bn_workspace(M, C, dev)
    y0, sm, sr = ops.bn_train_fwd(x, gamma, beta, 1e-3, relu, ws)
    p0 = ops.maxpool_fwd(y0.view(1, M // 2, 2, C), 1, 2)
    pooled = torch.empty(M // 2, C, dtype=BF, device=dev)
    y1, sm1, sr1 = ops.bn_train_fwd(x, gamma, beta, 1e-3, relu, ws, pooled=pooled)
    assert torch.equal(y1, y0) and torch.equal(pooled, p0.view(M // 2, C)) and torch.equal(sm1, sm) and torch.equal(sr1, sr)
    dp = bf(gen((M // 2, C), 4)).to(dev).to(BF)
    dy = ops.maxpool_bwd(y0.view(1, M // 2, 2, C), dp.view(1, M // 2, 1, C), 1, 2, relu_mask=False)
    dg0 = torch.zeros(C, device=dev); db0 = torch.zeros(C, device=dev)
    dx0 = ops.bn_train_bwd(x, y0, dy.view(M, C), gamma, sm, sr, dg0, db0, relu, ws)
    dg1 = torch.zeros(C, device=dev); db1 = torch.zeros(C, device=dev)
    dx1 = ops.bn_train_bwd(x, y0, dp, gamma, sm, sr, dg1, db1, relu, ws, pooled_dy=True)
    assert torch.equal(dx1, dx0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)


@pytest.mark.parametrize("Nb,W,H,Ci,Co", [(64, 64, 4, 256, 512), (64, 64, 4, 512, 512), (64, 80, 4, 256, 512), (32, 64, 4, 256, 256), (16, 64, 8, 128, 128),
                                          (32, 64, 2, 512, 512), (8, 22, 4, 256, 512)])
def test_conv3x3_with_batchnorm_statistics_in_the_epilogue(dev, Nb, W, H, Ci, Co):
    """Round 4: conv_k3 / conv_k3w leave per-tile partial sums of their (bf16) output behind; batch norm finishes from them.  The stored
    tensor is bit-identical to the plain convolution's; the statistics equal the ones of a pass over it up to fp32 summation order."""
    rows = ops.conv3x3_stats_rows(Nb, W, H, Ci, Co)
    M = Nb * W * H
    if not ops.conv3x3_kernel_choice(Nb, W, H, Ci, Co, relu=False).startswith('conv_k3'):
        assert rows == 0          # (8, 22, 4): 704 pixels; (16, 64, 8, 128, 128): 32 tiles — no plane-layout kernel: the caller keeps the statistics pass
        return
    assert rows == M // 256
    x = bf(gen((Nb, W, H, Ci), 1)).to(dev).to(BF); w = bf(gen((3, 3, Ci, Co), 2, 0.05)); b = gen((Co,), 3).to(dev)
    wpack = torch.empty((Co, 3, 3, Ci), dtype=BF, device=dev)
    ops.pack_transpose(w.reshape(9 * Ci, Co).to(dev), wpack)
    y0 = ops.conv3x3(x, wpack, bias=b, relu=False)
    ws = ops.bn_workspace(M, Co, dev)
    ws.fill_(0x7f)
    y1 = torch.empty_like(y0)
    ops.conv3x3_stats(x, wpack, y1, ws, bias=b)
    assert torch.equal(y1, y0)
    part = ws[:rows * 2 * Co * 4].view(torch.float32).view(rows, 2, Co).double()
    yd = y0.view(M, Co).double()
    assert relerr(part[:, 0].sum(0).cpu(), yd.sum(0).cpu()) < 1e-5 and relerr(part[:, 1].sum(0).cpu(), (yd * yd).sum(0).cpu()) < 1e-5
    # every tile's own row: rows t*256 .. t*256+255 of the [M][Co] view
    t = rows // 2
    assert relerr(part[t, 0].cpu(), yd[t * 256:(t + 1) * 256].sum(0).cpu()) < 1e-5
    gamma = (gen((Co,), 4) + 1.5).to(dev); beta = gen((Co,), 5).to(dev)
    ws2 = ops.bn_workspace(M, Co, dev)
    z0, m0, r0 = ops.bn_train_fwd(y0.view(M, Co), gamma, beta, 1e-3, True, ws2)
    z1, m1, r1 = ops.bn_train_fwd(y1.view(M, Co), gamma, beta, 1e-3, True, ws, partial_rows=rows)
    assert relerr(m1.cpu(), m0.cpu()) < 1e-5 and relerr(r1.cpu(), r0.cpu()) < 1e-5
    assert maxerr(z1.float().cpu(), z0.float().cpu()) <= 2.0 ** -6 * float(z0.float().abs().max())       # one bf16 ulp where a value sits on a rounding edge


@pytest.mark.parametrize("Nb,W,H,Ci,Co", [(64, 64, 4, 512, 256), (64, 64, 4, 512, 512), (64, 80, 4, 512, 256), (32, 64, 8, 128, 128), (16, 64, 8, 128, 128)])
def test_conv3x3_dgrad_with_batchnorm_backward_sums(dev, Nb, W, H, Ci, Co):
    """Round 4: the data gradient INTO a batch-norm + ReLU layer (conv_k3b): dx = (y > 0) ? conv3x3(dy, flipped weights) : 0, bit-identical to
    the masked convolution, plus that layer's batch-norm backward sums per tile; bn_train_bwd(partial_rows) then matches the three-pass form."""
    rows = ops.conv3x3_bnbwd_rows(Nb, W, H, Ci, Co)
    if not ops.conv3x3_kernel_choice(Nb, W, H, Ci, Co, bias=False, relu=False, mask=True).startswith('conv_k3'):
        assert rows == 0
        return
    M = Nb * W * H
    assert rows == M // 256
    dy = bf(gen((Nb, W, H, Ci), 1)).to(dev).to(BF); w = bf(gen((3, 3, Co, Ci), 2, 0.05))
    wd = torch.empty((Co, 3, 3, Ci), dtype=BF, device=dev)
    ops.pack_conv_dgrad(w.to(dev), wd)                               # [Co][9][Ci]: the data-gradient operand of a conv with C_in = Co, C_out = Ci
    z = bf(gen((M, Co), 3) * 2 + 0.5).to(dev).to(BF)
    gamma = (gen((Co,), 4) + 1.5).to(dev); beta = gen((Co,), 5).to(dev)
    ws = ops.bn_workspace(M, Co, dev)
    y, mean, rstd = ops.bn_train_fwd(z, gamma, beta, 1e-3, True, ws)
    dx0 = ops.conv3x3(dy, wd, mask=y.view(Nb, W, H, Co))
    ws1 = ops.bn_workspace(M, Co, dev); ws1.fill_(0x7f)
    dx1 = torch.empty_like(dx0)
    ops.conv3x3_dgrad_bnbwd(dy, wd, dx1, y.view(Nb, W, H, Co), z, mean, rstd, ws1)
    assert torch.equal(dx1, dx0)
    part = ws1[:rows * 2 * Co * 4].view(torch.float32).view(rows, 2, Co).double().cpu()
    g = dx0.view(M, Co).double().cpu()
    xhat = (z.double().cpu() - mean.double().cpu()) * rstd.double().cpu()
    assert relerr(part[:, 0].sum(0), g.sum(0)) < 1e-5 and relerr(part[:, 1].sum(0), (g * xhat).sum(0)) < 1e-5
    dg0 = torch.zeros(Co, device=dev); db0 = torch.zeros(Co, device=dev)
    dz0 = ops.bn_train_bwd(z, y, dx0.view(M, Co), gamma, mean, rstd, dg0, db0, True, ws)
    dg1 = torch.zeros(Co, device=dev); db1 = torch.zeros(Co, device=dev)
    dz1 = ops.bn_train_bwd(z, y, dx1.view(M, Co), gamma, mean, rstd, dg1, db1, True, ws1, partial_rows=rows)
    assert relerr(dg1.cpu(), dg0.cpu()) < 1e-5 and relerr(db1.cpu(), db0.cpu()) < 1e-5
    assert maxerr(dz1.float().cpu(), dz0.float().cpu()) <= 2.0 ** -6 * float(dz0.float().abs().max())


def test_small_ops(dev):
    a = bf(gen((1000, 512), 1))
    out = torch.ones(512, device=dev)
    ops.colsum(a.to(dev).to(BF), out)
    assert relerr(out.cpu(), a.sum(0) + 1) < 1e-4
    src = gen((1031,), 2).to(dev); dst = torch.empty(1031, dtype=BF, device=dev)
    ops.cast_bf16(src, dst)
    assert maxerr(dst.float().cpu(), bf(src.cpu())) == 0.0
    s2 = gen((10, 24), 3).to(dev); d2 = torch.zeros((10, 32), dtype=BF, device=dev)
    ops.cast2d_bf16(s2, 24, d2[:, 8:], 32, 10, 24)
    assert maxerr(d2[:, 8:].float().cpu(), bf(s2.cpu())) == 0.0 and float(d2[:, :8].float().abs().sum()) == 0.0
    g = gen((7, 5, 64), 4).to(dev); o = torch.empty((5, 7, 64), dtype=BF, device=dev)
    ops.tnc_to_ntc_bf16(g, o, 0.25)
    assert maxerr(o.float().cpu(), bf(g.cpu().permute(1, 0, 2) * 0.25)) == 0.0
    # LSTM packing: gate-major column g*U+u -> packed (u/16)*64 + g*16 + u%16, transposed
    U, D = 32, 24
    w = gen((D, 4 * U), 5)
    packed = torch.empty((4 * U, D), dtype=BF, device=dev)
    ops.pack_transpose(w.to(dev), packed, lstm_units=U)
    ref = torch.empty(4 * U, D)
    for gte in range(4):
        for u in range(U):
            ref[(u // 16) * 64 + gte * 16 + u % 16] = w[:, gte * U + u]
    assert maxerr(packed.float().cpu(), bf(ref)) == 0.0
    col = bf(gen((3 * 8, 2, 64), 6)); dx = torch.empty((3, 9, 64), dtype=BF, device=dev)
    ops.conv5_col2im(col.to(dev).to(BF), dx, 3, 9, 64)
    c = col.reshape(3, 8, 2, 64); ref = torch.zeros(3, 9, 64)
    ref[:, :8] += c[:, :, 0]; ref[:, 1:] += c[:, :, 1]
    assert maxerr(dx.float().cpu(), bf(ref)) == 0.0
    col = bf(gen((4 * 5, 2, 36), 8)); dx = torch.empty((4, 6, 36), dtype=BF, device=dev)            # row length % 8 != 0: the scalar kernel
    ops.conv5_col2im(col.to(dev).to(BF), dx, 4, 6, 36)
    c = col.reshape(4, 5, 2, 36); ref = torch.zeros(4, 6, 36)
    ref[:, :5] += c[:, :, 0]; ref[:, 1:] += c[:, :, 1]
    assert maxerr(dx.float().cpu(), bf(ref)) == 0.0
    col = bf(gen((7 * 32, 2, 1024), 7)); dx = torch.empty((7, 33, 1024), dtype=BF, device=dev)     # the headline row length (H * C = 2 * 512)
    ops.conv5_col2im(col.to(dev).to(BF), dx, 7, 33, 1024)
    c = col.reshape(7, 32, 2, 1024); ref = torch.zeros(7, 33, 1024)
    ref[:, :32] += c[:, :, 0]; ref[:, 1:] += c[:, :, 1]
    assert maxerr(dx.float().cpu(), bf(ref)) == 0.0


# ------------------------------------------------------------------------------------------- LSTM
def _lstm_device_forward(dev, x, seq_len, Ws, bs, U, persistent=False, prepared=False):
    """Runs the hoisted projection + per-step kernels exactly as the executor does. x: [N,T,D] bf16-rounded fp32."""
    N, T, D = x.shape
    R = N * T
    wxT = torch.empty((8 * U, D), dtype=BF, device=dev)
    whT = torch.empty((2, 4 * U, U), dtype=BF, device=dev)
    for d in range(2):
        Wd = Ws[d].to(dev)
        ops.pack_transpose(Wd[:D], wxT[d * 4 * U:(d + 1) * 4 * U], lstm_units=U, R=D, Cc=4 * U, ldin=4 * U)
        ops.pack_transpose(Wd[D:], whT[d], lstm_units=U, R=U, Cc=4 * U, ldin=4 * U)
    bias = torch.empty(8 * U, device=dev)
    ops.lstm_pack_bias(bs[0].to(dev), bs[1].to(dev), bias, U)
    xd = x.to(dev).to(BF).reshape(R, D)
    xproj = ops.gemm_nt(xd, wxT, bias=bias, out_f32=True)
    sl = torch.tensor(seq_len, dtype=torch.int32, device=dev)
    hout = torch.full((R, 2 * U), 7.0, dtype=BF, device=dev)          # poison: kernels must overwrite every row
    gates = torch.zeros((2, R, 4 * U), device=dev); cell = torch.zeros((2, R, U), device=dev)
    if persistent:
        assert ops.lstm_seq_supported(N, U)
        # prepared: the caller has set every word of the hand-off block to 0xFFFFFFFF (what the training engine does inside its first
        # kernel) and the call skips its own fill launch; the error word then reads -1 (untouched) or 1 (time-out)
        sync = torch.full((ops.lstm_seq_sync_words(N, U),), -1 if prepared else 0, dtype=torch.int32, device=dev)
        ops.lstm_fwd_seq(xproj, whT, sl, hout, gates, cell, N, T, U, sync, prepared=prepared)
        torch.cuda.synchronize()
        # (under the counter protocol the flag is ignored and the call prepares — zeroes — the block itself)
        assert int(sync[-1]) in ((-1, 0) if prepared else (0,)), "persistent LSTM forward: spin timeout"
    else:
        for s in range(T):
            ops.lstm_fwd_step(xproj, whT, sl, hout, gates, cell, N, T, U, s)
    return dict(xd=xd, sl=sl, hout=hout, gates=gates, cell=cell)


@pytest.mark.parametrize("N,T,D,U,lens,persistent", [(64, 21, 512, 256, None, False), (5, 9, 64, 32, [9, 4, 1, 7, 9], False),
                                                        (70, 6, 64, 32, None, False), (64, 63, 512, 256, None, True),
                                                        (100, 12, 64, 256, None, True), (3, 5, 64, 256, [5, 1, 3], True), (200, 4, 64, 256, None, True), (8, 21, 512, 256, None, True),
                                                        # BASELINE configs[4]: 512 units per direction (contraction axis split over two waves); layer-2 input 1024
                                                        (8, 9, 64, 512, None, False), (64, 63, 512, 512, None, True), (32, 21, 1024, 512, None, True),
                                                        (100, 12, 64, 512, None, True), (3, 5, 64, 512, [5, 1, 3], True)])
def test_lstm_fwd_bwd(dev, N, T, D, U, lens, persistent):
    rng = np.random.RandomState(1)
    seq_len = lens if lens is not None else rng.randint(max(1, T // 2), T + 1, N).tolist()
    x = bf(gen((N, T, D), 1))
    Ws = [gen((D + U, 4 * U), 2 + d, 0.08) for d in range(2)]
    bs = [gen((4 * U,), 4 + d, 0.1) for d in range(2)]
    xr = x.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    fw = og.lstm_direction(xr, seq_len, Wr[0], br[0], False, True)
    bw = og.lstm_direction(xr, seq_len, Wr[1], br[1], True, True)
    ref = torch.cat([fw, bw], 2)
    dh = bf(gen((N, T, 2 * U), 9))
    ref.backward(dh)
    # the persistent kernels come in two families (four waves per workgroup — the default since round 4 — and one): both through the same checks
    # ... and with the hand-off block prepared by the caller instead of by the call's own fill launch
    for ksplit, prepared in (((4, False), (4, True), (1, False)) if persistent else ((None, False),)):
        if ksplit is not None:
            ops.set_lstm_ksplit(ksplit)
        try:
            _lstm_device_checks(dev, N, T, D, U, x, seq_len, Ws, bs, Wr, br, xr, ref, dh, persistent, prepared)
        finally:
            if ksplit is not None:
                ops.set_lstm_ksplit(4)


def _lstm_device_checks(dev, N, T, D, U, x, seq_len, Ws, bs, Wr, br, xr, ref, dh, persistent, prepared=False):
    st = _lstm_device_forward(dev, x, seq_len, Ws, bs, U, persistent, prepared)
    got = st["hout"].float().cpu().reshape(N, T, 2 * U)
    # measured on MI355X (round 2): <= 3.9e-3 = one bf16 ulp of |h| in [0.5, 1) (a rounding flip of the stored h), usually 0 .. 1e-3
    assert maxerr(got, ref.detach()) < 8e-3, maxerr(got, ref.detach())
    # ---- backward
    R = N * T
    whb = torch.empty((2, D + U, 4 * U), dtype=BF, device=dev)
    for d in range(2):
        ops.cast_bf16(Ws[d].to(dev).contiguous(), whb[d])
    dz = torch.full((R, 8 * U), 3.0, dtype=BF, device=dev)
    dc = torch.zeros((2, N, U), device=dev)
    dhd = dh.to(dev).to(BF).reshape(R, 2 * U)
    if persistent:
        sync = torch.full((ops.lstm_seq_sync_words(N, U),), -1 if prepared else 0, dtype=torch.int32, device=dev)
        ops.lstm_bwd_seq(whb[:, D:], 4 * U, (D + U) * 4 * U, st["sl"], dhd, st["gates"], st["cell"], dz, N, T, U, sync, prepared=prepared)
        torch.cuda.synchronize()
        assert int(sync[-1]) in ((-1, 0) if prepared else (0,)), "persistent LSTM backward: spin timeout"
    else:
        for s in range(T - 1, -1, -1):
            ops.lstm_bwd_step(whb[:, D:], 4 * U, (D + U) * 4 * U, st["sl"], dhd, st["gates"], st["cell"], dz, dc, N, T, U, s)
    hprev = torch.empty((2, R, U), dtype=BF, device=dev)
    ops.lstm_hprev(st["hout"], st["sl"], hprev, N, T, U)
    for d in range(2):
        dW = torch.zeros((D + U, 4 * U), device=dev); dbias = torch.zeros(4 * U, device=dev)
        ops.gemm_tn(st["xd"], dz[:, d * 4 * U:(d + 1) * 4 * U], dW[:D], Mk=R, I=D, J=4 * U, lda=D, ldb=8 * U, ldo=4 * U, colsum=dbias)
        ops.gemm_tn(hprev[d], dz[:, d * 4 * U:(d + 1) * 4 * U], dW[D:], Mk=R, I=U, J=4 * U, lda=U, ldb=8 * U, ldo=4 * U)
        # max-abs error relative to the largest entry; measured 1.6e-3 .. 4.3e-3 (dW), 1.2e-3 .. 3.1e-3 (db), 3.2e-3 .. 4.3e-3 (dx):
        # bf16 storage of dz and dh
        assert relerr(dW.cpu(), Wr[d].grad) < 1e-2, ("dW", d, relerr(dW.cpu(), Wr[d].grad))
        assert relerr(dbias.cpu(), br[d].grad) < 1e-2, ("db", d, relerr(dbias.cpu(), br[d].grad))
    wcat = torch.empty((D, 8 * U), dtype=BF, device=dev)
    for d in range(2):
        ops.cast2d_bf16(Ws[d].to(dev), 4 * U, wcat[:, d * 4 * U:], 8 * U, D, 4 * U)
    dx = ops.gemm_nt(dz, wcat)
    assert relerr(dx.float().cpu().reshape(N, T, D), xr.grad) < 1e-2


@pytest.mark.parametrize("N,T,U,lens,D", [(64, 63, 256, None, 512), (8, 21, 256, None, 512), (3, 5, 256, [5, 1, 3], 512), (100, 12, 256, None, 512),
                                         (32, 21, 512, None, 512), (32, 21, 512, None, 1024), (8, 9, 256, None, 1024)])
def test_lstm_forward_with_the_input_projection_inside(dev, N, T, U, lens, D):
    """ocr_lstm_fwd_seq_x (the projection's MFMAs inside the recurrent kernel, no projection tensor) against the projection GEMM +
    ocr_lstm_fwd_seq2 on the same operands: same hidden states up to one bf16 rounding flip, same saved gates / cells up to fp32 summation
    order (K = D + U summed in one accumulator chain instead of GEMM + add)."""
    if not ops.lstm_fwd_seq_x_supported(N, U, D):
        pytest.skip("shape / protocol not covered by the fused kernel on this device")
    rng = np.random.RandomState(3)
    seq_len = lens if lens is not None else rng.randint(max(1, T // 2), T + 1, N).tolist()
    x = bf(gen((N, T, D), 1))
    Ws = [gen((D + U, 4 * U), 2 + d, 0.08) for d in range(2)]
    bs = [gen((4 * U,), 4 + d, 0.1) for d in range(2)]
    ref = _lstm_device_forward(dev, x, seq_len, Ws, bs, U, persistent=True)
    # ... and DIRECTLY against the CPU oracle (oracle/graph.py::lstm_direction = TF-1.0 LSTMCell under bidirectional_dynamic_rnn,
    # /root/reference/lib/networks/network.py:104-109): this kernel is the one the product runs
    with torch.no_grad():
        oracle_h = torch.cat([og.lstm_direction(x, seq_len, Ws[0], bs[0], False, True), og.lstm_direction(x, seq_len, Ws[1], bs[1], True, True)], 2)
    R = N * T
    wxT = torch.empty((8 * U, D), dtype=BF, device=dev)
    whT = torch.empty((2, 4 * U, U), dtype=BF, device=dev)
    for d in range(2):
        Wd = Ws[d].to(dev)
        ops.pack_transpose(Wd[:D], wxT[d * 4 * U:(d + 1) * 4 * U], lstm_units=U, R=D, Cc=4 * U, ldin=4 * U)
        ops.pack_transpose(Wd[D:], whT[d], lstm_units=U, R=U, Cc=4 * U, ldin=4 * U)
    bias = torch.empty(8 * U, device=dev)
    ops.lstm_pack_bias(bs[0].to(dev), bs[1].to(dev), bias, U)
    for prepared in (False, True):
        hout = torch.full((R, 2 * U), 7.0, dtype=BF, device=dev)
        gates = torch.zeros((2, R, 4 * U), device=dev); cell = torch.zeros((2, R, U), device=dev)
        sync = torch.full((ops.lstm_seq_sync_words(N, U),), -1 if prepared else 0, dtype=torch.int32, device=dev)
        ops.lstm_fwd_seq_x(ref["xd"], wxT, bias, whT, ref["sl"], hout, gates, cell, N, T, U, sync, prepared=prepared)
        torch.cuda.synchronize()
        assert int(sync[-1]) == (-1 if prepared else 0), "spin timeout"
        valid = (torch.arange(T, device=dev)[None, :] < ref["sl"][:, None]).reshape(R)         # rows past their length hold throw-away values
        assert maxerr(hout.float()[valid].cpu(), ref["hout"].float()[valid].cpu()) < 8e-3
        assert float(hout.float()[~valid].abs().max() if (~valid).any() else 0.0) == 0.0
        # measured against the oracle on MI355X: one bf16 ulp of |h| in [0.5, 1) at most (3.9e-3), as for the unfused family in test_lstm_fwd_bwd
        assert maxerr(hout.float().cpu().reshape(N, T, 2 * U), oracle_h) < 8e-3, maxerr(hout.float().cpu().reshape(N, T, 2 * U), oracle_h)
        for d in range(2):
            assert maxerr(gates[d][valid].cpu(), ref["gates"][d][valid].cpu()) < 2e-3
            assert maxerr(cell[d][valid].cpu(), ref["cell"][d][valid].cpu()) < 4e-3


@pytest.mark.parametrize("env", [dict(OCR_LSTM_PROTO='0'), dict(OCR_LSTM_ROWS='32')])
def test_lstm_persistent_other_protocols_and_tiles(dev, env):
    """The persistent kernels' counter protocol (0: the placement-independent fallback) and 32-row workgroups forced where 16 rows are
    the default, through the same parity cases (the knobs are read once per process: own interpreter)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-q', '-k', 'test_lstm_fwd_bwd and True'],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:]


# ------------------------------------------------------------------------------------------- optimiser
@pytest.mark.parametrize("solver", ["Adam", "Momentum", "RMS"])
def test_optimizer(dev, solver):
    n, r0, r1 = 4096 + 1024, 512, 4096 + 512              # regularised range in the middle of the flat buffer
    p = gen((n,), 1); lr, wd, clip = 1e-2, 1e-3, 10.0
    pd = p.to(dev); s1 = torch.zeros(n, device=dev); s2 = torch.zeros(n, device=dev)
    sc = torch.zeros(ops.optim_scalar_count(), dtype=torch.float64, device=dev)
    ops.optim_init(sc, lr)
    pr = p.double().clone(); m = torch.zeros(n, dtype=torch.float64); v = torch.zeros(n, dtype=torch.float64)
    for step in range(1, 4):
        g = gen((n,), 10 + step, 30.0 if step == 2 else 0.01)       # step 2 triggers the clip
        gd = g.to(dev).clone()
        b1, b2, eps = (0.9, 0.999, 1e-8) if solver == "Adam" else ((0.9, 0.0, 0.0) if solver == "Momentum" else (0.9, 0.0, 1e-10))
        ops.optim_step(pd, gd, s1, s2, (r0, r1), wd, clip, ops.SOLVERS[solver], b1, b2, eps, sc)
        gg = g.double().clone(); gg[r0:r1] += wd * pr[r0:r1]
        norm = float(gg.norm()); gg *= clip / max(norm, clip)
        if solver == "Adam":
            m = 0.9 * m + 0.1 * gg; v = 0.999 * v + 0.001 * gg * gg
            lrt = lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            pr = pr - lrt * m / (v.sqrt() + 1e-8)
        elif solver == "Momentum":
            m = 0.9 * m + gg; pr = pr - lr * m
        else:
            m = 0.9 * m + 0.1 * gg * gg; pr = pr - lr * gg / (m + 1e-10).sqrt()
        torch.cuda.synchronize()
        assert abs(float(sc[7]) - norm) / norm < 1e-5
        assert relerr(pd.cpu().double(), pr) < 1e-5, (solver, step)
