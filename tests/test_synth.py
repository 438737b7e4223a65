"""GPU-side captcha synthesis, the parts that need no GPU (lstm_ctc_ocr_amd/utils/synth.py, tools/synth_model.py, oracle/synth_ref.py):
the parameter draws against the generator they restate (/root/reference/lib/lstm/utils/gen.py:24-67 via lstm_ctc_ocr_amd/utils/gen.py), and the
numpy model of the kernel against Pillow itself driven with the same parameters."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import synth_model as M  # noqa: E402
from lstm_ctc_ocr_amd.config import cfg  # noqa: E402
from lstm_ctc_ocr_amd.utils import gen, synth  # noqa: E402
from oracle import synth_ref  # noqa: E402

CONFIGS = {'C1': dict(), 'C2': dict(min_len=10, max_len=10, width=480), 'C4': dict(min_len=2, max_len=12, px_per_char=48)}


@pytest.fixture(scope='module')
def atlas():
    return synth.GlyphAtlas()


def test_atlas_holds_the_generator_masks(atlas):
    assert atlas.table.shape == (len(cfg.CHARSET) * 3, 3)
    for ci in (0, 7, len(cfg.CHARSET) - 1):
        for si, size in enumerate(synth.SIZES):
            want = np.array(gen._glyph_mask(cfg.CHARSET[ci], size))
            assert np.array_equal(atlas.mask(ci * 3 + si), want)


@pytest.mark.parametrize('name', list(CONFIGS))
def test_parameter_draws_follow_the_generator(atlas, name):
    kw = CONFIGS[name]
    P = synth.draw_params(np.random.default_rng(5), 256, atlas, **kw)
    lo, hi = kw.get('min_len', cfg.MIN_LEN), kw.get('max_len', cfg.MAX_LEN)
    assert P['L'].min() >= lo and P['L'].max() <= hi and P['packed'].shape == (256, synth.words_per_image(hi))
    assert len(P['labels']) == P['labels_len'].sum()
    pos = 0
    for i in range(256):
        p = synth.unpack_image(P['packed'][i], P['max_glyphs'])
        s = P['strings'][i]
        assert len(s) == p['L'] == P['labels_len'][i]
        assert [gen.encode_maps[c] for c in s] == P['labels'][pos:pos + len(s)].tolist()
        pos += len(s)
        assert 238 <= p['bg'] <= 255 and 10 <= p['fg'] <= 200
        w = p['width']
        assert P['steps'][i] == int(32 / 60 * w) // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP and p['nw_out'] == int(32 / 60 * w)
        assert p['canvas_w'] == max(w, sum(g['nw'] for g in p['glyphs']))
        assert (p['dots'][:, 0] >= 0).all() and (p['dots'][:, 0] <= w).all() and (p['dots'][:, 1] <= 60).all()
        x0, y0, x1, y1 = p['arc']
        assert 0 <= x0 <= w // 5 and w - w // 5 <= x1 <= w and 12 <= y0 < y1 <= 61 and 0 <= p['arc_start'] <= 20 and 160 <= p['arc_end'] <= 200
        if i < 24:                   # the rotated boxes and the pen positions are render_captcha_gray's for the same draws
            from PIL import Image
            x = int(0.1 * int(sum(g['nw'] for g in p['glyphs']) / max(1, p['L'])))
            for k, g in enumerate(p['glyphs']):
                assert g['x'] == x or k > 0
                m = Image.fromarray(atlas.data[g['off']:g['off'] + g['mw'] * g['mh']].reshape(g['mh'], g['mw']))
                ang = np.degrees(np.arctan2(-g['mat'][1], g['mat'][0]))          # m1 = sin(-angle), m0 = cos(-angle)
                assert -30.0001 <= ang <= 30.0001
                assert gen._rotate_mask(m, ang).size == (g['nw'], g['nh'])
                assert 0 <= g['y'] and abs(g['y'] - max(0, int((60 - g['nh']) / 2))) <= 4


@pytest.mark.parametrize('name', list(CONFIGS))
def test_model_is_pillow_bit_for_bit_without_the_arc(atlas, name):
    """every stage the kernel shares with Pillow — rotation, paste, bicubic resize, dots, SMOOTH, bilinear resize — in Pillow's own arithmetic"""
    P = synth.draw_params(np.random.default_rng(11), 6, atlas, **CONFIGS[name])
    stamp = synth.dot_stamp()
    for i in range(6):
        p = synth.unpack_image(P['packed'][i], P['max_glyphs'])
        a_img, a_small = synth_ref.render_stages(p, atlas, arc=False)
        b_img, b_small = M.render(p, atlas, stamp, arc=False)
        assert a_img.shape == (60, p['width']) and a_small.shape == (32, p['nw_out'])
        assert np.array_equal(a_img, b_img) and np.array_equal(a_small, b_small)


def test_arc_differs_from_pillow_on_a_bounded_number_of_pixels(atlas):
    """the noise arc is the kernel's own outline algorithm: most of Pillow's arc pixels, a few per image beside them"""
    from PIL import Image, ImageDraw
    P = synth.draw_params(np.random.default_rng(3), 200, atlas)
    diff = total = 0
    for i in range(200):
        p = synth.unpack_image(P['packed'][i], P['max_glyphs'])
        im = Image.new('L', (p['width'], 60), 0)
        ImageDraw.Draw(im).arc(list(p['arc']), p['arc_start'], p['arc_end'], fill=255)
        ys, xs = np.nonzero(np.array(im))
        a = set(zip(xs.tolist(), ys.tolist()))
        b = M.arc_pixels(p['arc'], p['arc_start'], p['arc_end'], p['width'], 60, p['arc_lines'])
        diff += len(a ^ b)
        total += len(a)
    assert diff < 0.15 * total, (diff, total)
    stamp = synth.dot_stamp()
    frac, mean = [], []
    for i in range(8):
        p = synth.unpack_image(P['packed'][i], P['max_glyphs'])
        a_img, a_small = synth_ref.render_stages(p, atlas)
        b_img, b_small = M.render(p, atlas, stamp)
        frac.append((a_img != b_img).mean())
        mean.append(np.abs(a_small.astype(int) - b_small.astype(int)).mean())
    assert max(frac) < 0.03 and np.mean(mean) < 0.5, (frac, mean)


def test_synthesised_images_look_like_the_generators(atlas):
    """same distribution as gen.sample_image + groupBatch: ink statistics of 300 images each (Pillow on both sides)"""
    import random
    random.seed(7)
    ref = []
    for _ in range(300):
        im, _s = gen.sample_image()
        ref.append(gen._resize(im, int(32 / 60 * im.shape[1]), 32))
    P = synth.draw_params(np.random.default_rng(7), 300, atlas)
    syn = [synth_ref.render_stages(synth.unpack_image(P['packed'][i], P['max_glyphs']), atlas)[1] for i in range(300)]
    for f in (lambda a: a.mean(), lambda a: (a < 128).mean(), lambda a: a.std()):
        r, s = np.array([f(a) for a in ref]), np.array([f(a) for a in syn])
        se = np.sqrt(r.var() / len(r) + s.var() / len(s))
        assert abs(r.mean() - s.mean()) < 4 * se + 1e-3, (r.mean(), s.mean(), se)


def test_dot_stamp_is_translation_invariant():
    from PIL import Image, ImageDraw
    st = {(int(a), int(b)) for a, b in synth.dot_stamp()}
    assert len(st) >= 5
    im = Image.new('L', (50, 40), 0)
    ImageDraw.Draw(im).draw.draw_lines(((31, 17), (30, 16)), 255, 3)
    ys, xs = np.nonzero(np.array(im))
    assert {(int(x) - 31, int(y) - 17) for x, y in zip(xs, ys)} == st


def test_parameter_worker_messages(atlas):
    """the child process that draws the parameters for the feeder: message layout, determinism, back-pressure through the pipe"""
    import multiprocessing
    B, G, chunk = 8, cfg.MAX_LEN, 3
    kw = dict(min_len=None, max_len=None, width=160, px_per_char=None)

    def take(seed, n):
        ctx = multiprocessing.get_context('fork')
        rd, wr = ctx.Pipe(duplex=False)
        pr = ctx.Process(target=synth._param_worker, args=(wr.send_bytes, seed, B, G, chunk, atlas, kw), daemon=True)
        pr.start()
        wr.close()
        out = []
        try:
            for _ in range(n):
                m = np.empty(synth.batch_words(B, G), np.int32)
                assert rd.poll(30)
                rd.recv_bytes_into(m)
                out.append(m)
        finally:
            rd.close()
            pr.terminate()
            pr.join(5)
        return out
    a, b = take(3, 5), take(3, 5)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not np.array_equal(a[0], a[1])
    S = synth.words_per_image(G)
    ref = synth.draw_params(np.random.default_rng(3), B * chunk, atlas)
    for c, m in enumerate(a[:chunk]):
        W, ccap, wcap, nlab = m[:4]
        assert W == 88 and wcap == 160 and ccap == ref['canvas_w'][c * B:(c + 1) * B].max()
        assert np.array_equal(m[4:4 + B * S].reshape(B, S), ref['packed'][c * B:(c + 1) * B])
        o = 4 + B * S
        ll = m[o + B * G:o + B * G + B]
        assert nlab == ll.sum() and np.array_equal(ll, ref['labels_len'][c * B:(c + 1) * B])
        want = [gen.encode_maps[ch] for s in ref['strings'][c * B:(c + 1) * B] for ch in s]
        assert m[o:o + nlab].tolist() == want
        assert (m[o + B * G + B:o + B * G + 2 * B] == 85 // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP).all()


def test_parameter_worker_as_a_process_of_its_own(atlas):
    """what DeviceSynthStream starts: `python -m lstm_ctc_ocr_amd.utils.synth`, job pickled on stdin, raw int32 batch messages on stdout — the same
    messages the in-process worker loop produces for the seed"""
    B, G, chunk = 8, cfg.MAX_LEN, 2
    kw = dict(min_len=None, max_len=None, width=160, px_per_char=None)
    pr = synth.spawn_worker(3, B, G, chunk, atlas, kw)
    try:
        nbytes = synth.batch_words(B, G) * 4
        msgs = []
        for _ in range(3):
            buf = bytearray()
            while len(buf) < nbytes:
                part = pr.stdout.read(nbytes - len(buf))
                assert part, 'worker ended early (exit status %s)' % pr.poll()
                buf += part
            msgs.append(np.frombuffer(bytes(buf), np.int32))
    finally:
        pr.terminate()
        pr.wait(5)
        pr.stdout.close()
    ref = np.zeros((chunk, synth.batch_words(B, G)), np.int32)
    rng = np.random.default_rng(3)
    synth.fill_batches(synth.draw_params(rng, B * chunk, atlas, strings=False, **kw), B, G, ref)
    assert np.array_equal(msgs[0], ref[0]) and np.array_equal(msgs[1], ref[1])
    synth.fill_batches(synth.draw_params(rng, B * chunk, atlas, strings=False, **kw), B, G, ref)
    assert np.array_equal(msgs[2], ref[0])
