"""TEST INFRASTRUCTURE — the checker of the GPU captcha synthesis (lstm_ctc_ocr_amd/csrc/captcha_synth.hip): PIL itself, driven with the
parameters utils/synth.draw_params drew.  The body is lstm_ctc_ocr_amd.utils.gen.render_captcha_gray + groupBatch's resize with every random
draw replaced by the drawn value (reference: /root/reference/lib/lstm/utils/gen.py:31-37 generateImg, :41-67 groupBatch).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from PIL import Image, ImageDraw, ImageFilter

HEIGHT = 60


def render_stages(p, atlas, out_h=32, arc=True):
    """p: utils.synth.unpack_image(...).  Returns (captcha [60, width] uint8 before the final resize, resized [32, nw_out] uint8)."""
    canvas = Image.new('L', (p['canvas_w'], HEIGHT), p['bg'])
    cim = canvas.im
    for g in p['glyphs']:
        m = Image.fromarray(np.ascontiguousarray(atlas.data[g['off']:g['off'] + g['mw'] * g['mh']].reshape(g['mh'], g['mw'])))
        rot = Image.new('L', (g['nw'], g['nh']), None)
        rot.im.transform((0, 0, g['nw'], g['nh']), m.im, 0, g['mat'], 2, 1)        # AFFINE, BILINEAR, fill (gen._rotate_mask)
        cim.paste(p['fg'], (g['x'], g['y'], g['x'] + g['nw'], g['y'] + g['nh']), rot.im)
    if p['canvas_w'] > p['width']:
        canvas = canvas.resize((p['width'], HEIGHT))
    d = ImageDraw.Draw(canvas)
    ink = d._getink(p['fg'])[0]
    for px, py in p['dots']:
        d.draw.draw_lines(((int(px), int(py)), (int(px) - 1, int(py) - 1)), ink, 3)
    if arc:
        d.arc(list(p['arc']), p['arc_start'], p['arc_end'], fill=p['fg'])
    img = np.array(canvas.filter(ImageFilter.SMOOTH))
    return img, np.array(Image.fromarray(img).resize((p['nw_out'], out_h), Image.BILINEAR))


def render_batch(params, atlas, W, arc=True):
    """[n, W, 32] uint8, right-padded with 0: what group_batch_u8 writes for the same images."""
    from lstm_ctc_ocr_amd.utils.synth import unpack_image
    n = params['packed'].shape[0]
    out = np.zeros((n, W, 32), np.uint8)
    for i in range(n):
        _, small = render_stages(unpack_image(params['packed'][i], params['max_glyphs']), atlas, arc=arc)
        out[i, :small.shape[1], :] = small.T
    return out
