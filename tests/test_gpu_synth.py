"""GPU-side captcha synthesis (csrc/captcha_synth.hip through the C ABI): the kernel against the numpy model (same algorithm, bit for bit) and
against Pillow itself driven with the same parameters (oracle/synth_ref.py; everything but the noise arc is Pillow's arithmetic and must be
identical), then the stream the training loop consumes.  Reference: /root/reference/lib/lstm/utils/gen.py:31-67."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import synth_model as M  # noqa: E402
from lstm_ctc_ocr_amd import ops  # noqa: E402
from lstm_ctc_ocr_amd.config import cfg  # noqa: E402
from lstm_ctc_ocr_amd.utils import gen, synth  # noqa: E402
from oracle import synth_ref  # noqa: E402

pytestmark = pytest.mark.gpu
CONFIGS = {'C1': dict(), 'C2': dict(min_len=10, max_len=10, width=480), 'C4': dict(min_len=2, max_len=12, px_per_char=48)}


@pytest.fixture(scope='module')
def atlas():
    return synth.GlyphAtlas()


def _launch(P, atlas, W, n):
    dev = 'cuda:0'
    packed = torch.from_numpy(P['packed'][:n].reshape(-1).copy()).to(dev)
    out = torch.full((n, W, 32), 77, dtype=torch.uint8, device=dev)
    ops.captcha_synth(packed, n, P['packed'].shape[1], torch.from_numpy(atlas.data).to(dev), torch.from_numpy(synth.dot_stamp().reshape(-1)).to(dev),
                      out, W, max_glyphs=P['max_glyphs'], canvas_cap=int(P['canvas_w'][:n].max()), width_cap=int(P['widths'][:n].max()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('name', list(CONFIGS))
def test_kernel_equals_the_model_bit_for_bit(atlas, name):
    n = 12
    P = synth.draw_params(np.random.default_rng(21), n, atlas, **CONFIGS[name])
    W = gen.padded_width(int(P['nw_out'].max())) + 4
    got = _launch(P, atlas, W, n)
    stamp = synth.dot_stamp()
    for i in range(n):
        p = synth.unpack_image(P['packed'][i], P['max_glyphs'])
        _, small = M.render(p, atlas, stamp)
        assert np.array_equal(got[i, :p['nw_out']], small.T), (name, i, np.abs(got[i, :p['nw_out']].astype(int) - small.T.astype(int)).max())
        assert not got[i, p['nw_out']:].any()                           # right padding with 0 (gen.py:62)


@pytest.mark.parametrize('name', list(CONFIGS))
def test_kernel_equals_pillow_without_the_arc(atlas, name):
    n = 64
    P = synth.draw_params(np.random.default_rng(22), n, atlas, **CONFIGS[name])
    P['packed'][:, 8] = P['packed'][:, 6]                               # an empty arc box: the kernel skips the one primitive that is not Pillow's
    W = gen.padded_width(int(P['nw_out'].max()))
    got = _launch(P, atlas, W, n)
    want = synth_ref.render_batch(P, atlas, W, arc=False)
    assert np.array_equal(got, want), (name, int((got != want).sum()))


def test_kernel_with_arc_is_close_to_pillow(atlas):
    n = 64
    P = synth.draw_params(np.random.default_rng(23), n, atlas)
    W = gen.padded_width(int(P['nw_out'].max()))
    got = _launch(P, atlas, W, n).astype(int)
    want = synth_ref.render_batch(P, atlas, W).astype(int)
    assert (got != want).mean() < 0.03 and np.abs(got - want).mean() < 0.5


def test_entry_point_refuses_what_it_cannot_hold(atlas):
    from lstm_ctc_ocr_amd._native import NativeError
    P = synth.draw_params(np.random.default_rng(1), 2, atlas)
    dev = 'cuda:0'
    packed = torch.from_numpy(P['packed'].reshape(-1).copy()).to(dev)
    out = torch.zeros((2, 88, 32), dtype=torch.uint8, device=dev)
    a, s = torch.from_numpy(atlas.data).to(dev), torch.from_numpy(synth.dot_stamp().reshape(-1)).to(dev)
    with pytest.raises(NativeError):
        ops.captcha_synth(packed, 2, P['packed'].shape[1], a, s, out, 88, max_glyphs=P['max_glyphs'], canvas_cap=2048, width_cap=600)   # > 160 KB of LDS
    with pytest.raises(NativeError):
        ops.captcha_synth(packed, 2, 40, a, s, out, 88, max_glyphs=P['max_glyphs'])                 # records shorter than their glyph slots


def test_stream_feeds_the_training_step(atlas):
    """DeviceSynthStream yields what DeviceBatchStream yields; the labels are the strings that were drawn; Engine.train_step learns from it"""
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    st = synth.DeviceSynthStream('cuda:0', 32, seed=5, chunk=4)
    try:
        eng = Engine(get_network('LSTM_train'), device='cuda:0', seed=3)
        eng.setup_optimizer('Adam', 1e-3)
        losses = []
        for it in range(60):
            pix, lab, ll, steps = next(st)
            assert pix.dtype == torch.uint8 and pix.shape == (32, 88, 32) and pix.is_cuda and lab.dtype == torch.int32
            assert ll.shape == (32,) and steps.shape == (32,) and int(ll.sum()) == lab.numel()
            if it == 0:
                assert int(steps[0]) == 85 // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP
                assert 4 <= int(ll.min()) and int(ll.max()) <= 6 and 1 <= int(lab.min()) and int(lab.max()) <= len(cfg.CHARSET)
                assert not pix[:, 85:].any() and pix[:, :85].float().mean() > 128          # light background, zero padding
            losses.append(eng.train_step(pix, lab, ll, steps))
        assert np.isfinite(losses).all() and np.mean(losses[-10:]) < np.mean(losses[:5])
    finally:
        st.close()


def test_stream_delivers_what_its_seed_says(atlas):
    """the first batches of a stream = the kernel on the parameters its first worker draws from the stream's seed (determinism of the draws themselves:
    tests/test_synth.py::test_parameter_worker_messages) — labels, lengths, steps and pixels"""
    B, chunk = 8, 2
    st = synth.DeviceSynthStream('cuda:0', B, seed=9, chunk=chunk, workers=1)
    try:
        got = []
        for _ in range(chunk):
            got.append([t.clone() for t in next(st)])
        torch.cuda.synchronize()
    finally:
        st.close()
    P = synth.draw_params(np.random.default_rng(9), B * chunk, atlas)
    pos = 0
    for c, (pix, lab, ll, steps) in enumerate(got):
        s = slice(c * B, (c + 1) * B)
        sub = {'packed': P['packed'][s], 'canvas_w': P['canvas_w'][s], 'widths': P['widths'][s], 'max_glyphs': P['max_glyphs']}
        want = _launch(sub, atlas, pix.shape[1], B)
        assert np.array_equal(pix.cpu().numpy(), want)
        n = int(P['labels_len'][s].sum())
        assert lab.cpu().numpy().tolist() == P['labels'][pos:pos + n].tolist() and ll.cpu().numpy().tolist() == P['labels_len'][s].tolist()
        assert steps.cpu().numpy().tolist() == P['steps'][s].tolist()
        pos += n


def test_stream_with_variable_widths(atlas):
    """px_per_char batches: every batch padded to its own widest sample (gen.py:54-62), steps follow each sample's width (the engine's plans per W:
    tests/test_gpu_engine.py, bench.py --workload varwidth; through the training loop: tools/cli_throughput.py --synth --var)"""
    st = synth.DeviceSynthStream('cuda:0', 16, seed=11, chunk=2, workers=1, min_len=3, max_len=12, px_per_char=48)
    try:
        for _ in range(6):
            pix, lab, ll, steps = next(st)
            B, W, H = pix.shape
            assert B == 16 and H == 32 and W % cfg.POOL_SCALE == 0 and 76 <= W <= 320 and int(ll.sum()) == lab.numel()
            s = steps.cpu().numpy()
            nw = (s - cfg.OFFSET_TIME_STEP) * cfg.POOL_SCALE                      # each sample's own width, rounded down to the pool scale
            assert nw.max() <= W and W - nw.max() < 2 * cfg.POOL_SCALE
            host = pix.cpu().numpy()
            for i in range(B):
                assert not host[i, nw[i] + cfg.POOL_SCALE:].any() and host[i, :nw[i]].mean() > 100
        st.close()
        st.close()                                                                # idempotent
    finally:
        st.close()


def test_raw_ctypes_binding_of_the_entry_point(atlas):
    """INTEGRATION.md's stub, verbatim in spirit: plain ctypes on the shared library, no ops / _native wrappers — and Pillow as the checker"""
    import ctypes
    from lstm_ctc_ocr_amd import _native as nat
    lib = ctypes.CDLL(nat.LIB_PATH)
    lib.ocr_captcha_synth.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ocr_captcha_synth.restype = ctypes.c_int
    P = synth.draw_params(np.random.default_rng(31), 8, atlas)
    P['packed'][:, 8] = P['packed'][:, 6]                               # no arc: byte-identical to Pillow
    W = gen.padded_width(int(P['nw_out'].max()))
    n, S = P['packed'].shape
    rec = torch.from_numpy(P['packed']).cuda()
    atlas_dev = torch.from_numpy(atlas.data).cuda()
    stamp_dev = torch.from_numpy(synth.dot_stamp().reshape(-1)).cuda()
    pix = torch.empty((n, W, 32), dtype=torch.uint8, device='cuda')
    rc = lib.ocr_captcha_synth(rec.data_ptr(), n, S, P['max_glyphs'], atlas_dev.data_ptr(), stamp_dev.data_ptr(), stamp_dev.numel() // 2,
                               pix.data_ptr(), W, 32, int(P['canvas_w'].max()), int(P['widths'].max()), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(pix.cpu().numpy(), synth_ref.render_batch(P, atlas, W, arc=False))
