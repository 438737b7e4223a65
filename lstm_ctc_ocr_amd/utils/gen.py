"""On-the-fly captcha batches — the data/label contract of /root/reference/lib/lstm/utils/gen.py.

  gen_rand        :24-29   random string, MIN_LEN..MAX_LEN chars of CHARSET
  generateImg     :31-37   the reference renders with `captcha.ImageCaptcha(fonts=[cfg.FONT])` (160x60 RGB); that
                           package is not installable here, so a PIL renderer of the same geometry stands in
                           (warped glyphs + noise dots/curve).  Pixel-level identity with `captcha` is NOT claimed.
  groupBatch      :41-67   resize to height 32 keeping aspect (nw = int(32/h*w)), time_steps = nw//POOL_SCALE +
                           OFFSET_TIME_STEP, right-pad with 0 to the batch max width rounded up to x4, /255,
                           swap to [W, 32], flat label vector via encode_maps
  generator       :69-110  RGB -> "gray" with the reference's channel quirk (cv2 BGR2GRAY on an RGB array: SURVEY Q8)
  get_batch       :112-128 multiprocess prefetch through GeneratorEnqueuer
"""
import math
import os
import random
import sys
import time

import numpy as np
from PIL import Image, ImageDraw, ImageFilter, ImageFont

from ..config import cfg, get_encode_decode_dict
from .data_util import GeneratorEnqueuer

encode_maps, decode_maps = get_encode_decode_dict()
_FONT_CACHE = {}


_SUBSTITUTE_FONTS = ('/usr/share/fonts/truetype/dejavu/DejaVuSans-Bold.ttf', 'DejaVuSans-Bold.ttf')
_font_path = {}


def resolve_font():
    """Path of the TrueType file the renderer uses.  cfg.FONT ('fonts/Ubuntu-M.ttf', config.py:26 of the reference) is a binary
    asset of the reference repository and is not redistributed here: drop it into <ROOT_DIR>/fonts/ (or point $OCR_FONT at any
    .ttf) to render with it.  When it is absent the substitution is announced on stderr — once per process, never silently —
    and `OCR_STRICT_FONT=1` turns it into an error."""
    want = os.environ.get('OCR_FONT') or cfg.FONT
    if want in _font_path:
        return _font_path[want]
    path = want if os.path.isabs(want) else os.path.join(cfg.ROOT_DIR, want)
    if not os.path.exists(path):
        if os.environ.get('OCR_STRICT_FONT') == '1':
            raise IOError('captcha font %s not found (cfg.FONT / $OCR_FONT); OCR_STRICT_FONT=1 forbids a substitute' % path)
        for cand in _SUBSTITUTE_FONTS:
            try:
                ImageFont.truetype(cand, 12)
            except (IOError, OSError):
                continue
            sys.stderr.write('[gen] WARNING: captcha font %s not found; rendering with %s instead '
                             '(copy the font there or set $OCR_FONT; OCR_STRICT_FONT=1 makes this an error)\n' % (path, cand))
            path = cand
            break
        else:
            raise IOError('captcha font %s not found and no substitute TrueType font is installed' % path)
    _font_path[want] = path
    return path


def _font(size):
    path = resolve_font()
    key = (path, size)
    if key not in _FONT_CACHE:
        _FONT_CACHE[key] = ImageFont.truetype(path, size)
    return _FONT_CACHE[key]


# random.randint / random.choice cost ~2 us per call in CPython (three Python frames each) and the gray renderer makes ~85 of them per image: the
# same draws inlined (random.randrange -> _randbelow_with_getrandbits: k = n.bit_length(), reject r >= n) — VALUE-IDENTICAL to the library calls for
# the same generator state, so render_captcha (which keeps calling the library) and render_captcha_gray still see the same geometry
_getrandbits = random.getrandbits          # (bound to the module-level generator: random.seed() reseeds it in place)


def _rb(n):
    """random._randbelow(n): uniform integer in [0, n)"""
    k = n.bit_length()
    r = _getrandbits(k)
    while r >= n:
        r = _getrandbits(k)
    return r


def _ri(a, b):
    """random.randint(a, b)"""
    return a + _rb(b - a + 1)


def randRGB():
    return (random.randint(0, 255), random.randint(0, 255), random.randint(0, 255))


def _rotate_mask(m, angle):
    """m.rotate(angle, Image.BILINEAR, expand=1) for an 'L' image without PIL's Python layers (63 us per call of which 24 are the C transform: five
    glyphs per captcha made it 40 % of an image): the same matrix, the same output size, the same core call — bit-identical
    (tests/test_data.py::test_fast_render_path_is_bit_identical)."""
    angle = angle % 360.0
    if angle == 0 or angle == 180 or angle == 90 or angle == 270:
        return m.rotate(angle, Image.BILINEAR, expand=1)
    w, h = m.size
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    ca, sa = round(math.cos(a), 15), round(math.sin(a), 15)
    m0, m1, m3, m4 = ca, sa, -sa, ca
    m2 = m0 * -cx + m1 * -cy + 0.0 + cx
    m5 = m3 * -cx + m4 * -cy + 0.0 + cy
    xx, yy = [], []
    for x, y in ((0, 0), (w, 0), (w, h), (0, h)):
        xx.append(m0 * x + m1 * y + m2)
        yy.append(m3 * x + m4 * y + m5)
    nw = math.ceil(max(xx)) - math.floor(min(xx))
    nh = math.ceil(max(yy)) - math.floor(min(yy))
    tx, ty = -(nw - w) / 2.0, -(nh - h) / 2.0
    m2, m5 = m0 * tx + m1 * ty + m2, m3 * tx + m4 * ty + m5
    out = Image.new('L', (nw, nh), None)
    out.im.transform((0, 0, nw, nh), m.im, 0, (m0, m1, m2, m3, m4, m5), 2, 1)      # Transform.AFFINE, Resampling.BILINEAR, fill
    return out


def gen_rand(min_len=None, max_len=None):
    n = _ri(cfg.MIN_LEN if min_len is None else min_len, cfg.MAX_LEN if max_len is None else max_len)
    cs = cfg.CHARSET
    k = len(cs)
    return "".join([cs[_rb(k)] for _ in range(n)])


def render_captcha(chars, width=160, height=60):
    """Captcha-style RGB image of `chars` on a light background (stand-in for captcha.ImageCaptcha.generate_image)."""
    bg = tuple(random.randint(238, 255) for _ in range(3))
    fg = tuple(random.randint(10, 200) for _ in range(3))
    img = Image.new('RGB', (width, height), bg)
    glyphs = []
    for ch in chars:
        f = _font(random.choice((42, 50, 56)))
        box = f.getbbox(ch)
        w, h = max(1, box[2] - box[0] + 4), max(1, box[3] - box[1] + 4)
        g = Image.new('RGBA', (w, h), (0, 0, 0, 0))
        ImageDraw.Draw(g).text((2 - box[0], 2 - box[1]), ch, font=f, fill=fg + (255,))
        g = g.rotate(random.uniform(-30, 30), Image.BILINEAR, expand=1)
        glyphs.append(g)
    text_w = sum(g.size[0] for g in glyphs)
    avg = int(text_w / max(1, len(chars)))
    x = int(0.1 * avg)
    canvas_w = max(text_w, width)
    canvas = Image.new('RGB', (canvas_w, height), bg)
    for g in glyphs:
        y = int((height - g.size[1]) / 2) + random.randint(-4, 4)
        canvas.paste(g, (x, max(0, y)), g)
        x += g.size[0] + random.randint(-int(0.25 * avg), 0)
    if canvas_w > width:
        canvas = canvas.resize((width, height))
    img.paste(canvas.crop((0, 0, width, height)), (0, 0))
    d = ImageDraw.Draw(img)
    for _ in range(30):                                       # noise dots
        px, py = random.randint(0, width), random.randint(0, height)
        d.line(((px, py), (px - 1, py - 1)), fill=fg, width=3)
    x1, x2 = random.randint(0, int(width / 5)), random.randint(width - int(width / 5), width)   # noise curve
    y1, y2 = random.randint(int(height / 5), height - int(height / 5)), random.randint(int(height / 5), height)
    d.arc([x1, min(y1, y2), x2, max(y1, y2) + 1], random.randint(0, 20), random.randint(160, 200), fill=fg)
    return img.filter(ImageFilter.SMOOTH)


_GRAY_W = (0.114, 0.587, 0.299)         # cv2 BGR2GRAY weights applied to an RGB array (to_gray_reference)
_MASK_CACHE = {}


def _gray_of(rgb):
    return int(min(255, max(0, round(_GRAY_W[0] * rgb[0] + _GRAY_W[1] * rgb[1] + _GRAY_W[2] * rgb[2]))))


def _glyph_mask(ch, size):
    """Coverage mask ('L') of one character at one font size, rendered ONCE per process (the RGB path renders every glyph of every image)."""
    key = (resolve_font(), size, ch)
    m = _MASK_CACHE.get(key)
    if m is None:
        f = _font(size)
        box = f.getbbox(ch)
        w, h = max(1, box[2] - box[0] + 4), max(1, box[3] - box[1] + 4)
        m = Image.new('L', (w, h), 0)
        ImageDraw.Draw(m).text((2 - box[0], 2 - box[1]), ch, font=f, fill=255)
        _MASK_CACHE[key] = m
    return m


def render_captcha_gray(chars, width=160, height=60):
    """render_captcha() followed by to_gray_reference(), computed in ONE channel: the same random draws in the same order (so the same
    geometry for the same RNG state), every step of the RGB path is linear in the colour (alpha compositing, resize, line / arc fills, the 3 x 3
    SMOOTH kernel), hence gray(render_captcha(...)) up to the 8-bit rounding of the intermediate images (a few gray levels at glyph edges:
    tests/test_data.py).  About 3x cheaper per image: cached glyph masks instead of a FreeType render per glyph, 'L' instead of RGBA / RGB
    transforms, filter and resize, no float matmul for the gray conversion — the live generator is the training loop's bottleneck, not the GPU."""
    bg = (_ri(238, 255), _ri(238, 255), _ri(238, 255))
    fg = (_ri(10, 200), _ri(10, 200), _ri(10, 200))
    bgv, fgv = _gray_of(bg), _gray_of(fg)
    # A glyph of the RGB path is an RGBA image holding the full ink wherever its coverage is non-zero; PIL rotates RGBA with PREMULTIPLIED
    # alpha, so the rotated colour stays the ink and only the alpha is interpolated: pasting it = pasting the solid ink through the rotated mask.
    glyphs = []
    for ch in chars:
        m = _glyph_mask(ch, (42, 50, 56)[_rb(3)])
        glyphs.append(_rotate_mask(m, random.uniform(-30, 30)))
    text_w = sum(a.size[0] for a in glyphs)
    avg = int(text_w / max(1, len(chars)))
    x = int(0.1 * avg)
    canvas_w = max(text_w, width)
    canvas = Image.new('L', (canvas_w, height), bgv)
    cim = canvas.im
    q = int(0.25 * avg)
    for a in glyphs:
        y = int((height - a.size[1]) / 2) + _ri(-4, 4)
        y0 = max(0, y)
        cim.paste(fgv, (x, y0, x + a.size[0], y0 + a.size[1]), a.im)
        x += a.size[0] + _ri(-q, 0)
    if canvas_w > width:
        canvas = canvas.resize((width, height))
    img = canvas.crop((0, 0, width, height)) if canvas.size != (width, height) else canvas
    d = ImageDraw.Draw(img)
    ink = d._getink(fgv)[0]
    lines = d.draw.draw_lines
    for _ in range(30):                                       # noise dots
        px, py = _ri(0, width), _ri(0, height)
        lines(((px, py), (px - 1, py - 1)), ink, 3)
    x1, x2 = _ri(0, int(width / 5)), _ri(width - int(width / 5), width)   # noise curve
    y1, y2 = _ri(int(height / 5), height - int(height / 5)), _ri(int(height / 5), height)
    d.arc([x1, min(y1, y2), x2, max(y1, y2) + 1], _ri(0, 20), _ri(160, 200), fill=fgv)
    return img.filter(ImageFilter.SMOOTH)


def sample_image(min_len=None, max_len=None, width=160, px_per_char=None):
    """One training sample as the generators feed it: (image array, label string).  Single-channel configurations (the reference's:
    cfg.NCHANNELS == 1) are rendered directly in gray unless OCR_RENDER=rgb asks for render_captcha + to_gray_reference."""
    chars = gen_rand(min_len, max_len)
    if px_per_char:
        width = min(600, len(chars) * px_per_char + random.randint(-8, 8))          # 600 px -> W = 320 at H = 32
    if cfg.NCHANNELS == 1 and os.environ.get('OCR_RENDER', 'gray') != 'rgb':
        return np.array(render_captcha_gray(chars, width, 60)), chars
    im = np.array(render_captcha(chars, width, 60))
    return (to_gray_reference(im) if cfg.NCHANNELS == 1 else im), chars


def generateImg(min_len=None, max_len=None, width=160, height=60):
    """One captcha.  Defaults = the reference (4-6 characters on ImageCaptcha's 160x60 canvas, gen.py:31-37); the keyword
    arguments produce the wider workloads of BASELINE.json (10 characters on 480x60 -> W = 256 after the resize to H = 32)."""
    chars = gen_rand(min_len, max_len)
    return np.array(render_captcha(chars, width, height)), chars


def to_gray_reference(im_rgb):
    """cv2.cvtColor(im, COLOR_BGR2GRAY) applied to an RGB array (gen.py:77-78): 0.114 R + 0.587 G + 0.299 B."""
    w = np.array([0.114, 0.587, 0.299], np.float32)
    return np.clip(np.rint(im_rgb.astype(np.float32) @ w), 0, 255).astype(np.uint8)


def _resize(img, nw, nh):
    return np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))


def padded_width(max_w):
    """Width a batch is padded to: the widest sample rounded up to POOL_SCALE (gen.py:54), or — opt-in, OCR_WIDTH_BUCKET=b — to
    the next multiple of b.  Bucketing bounds the number of distinct shapes (one engine plan + hipGraphs per width: <= 9 instead
    of <= 61 for W in [80, 320] at b = 32) and evens out data-parallel ranks; per-sample time steps are unaffected, only
    batch-norm statistics see the extra zero columns."""
    q = cfg.POOL_SCALE
    b = int(os.environ.get('OCR_WIDTH_BUCKET', '0') or 0)
    if b > 0:
        q = -(-b // cfg.POOL_SCALE) * cfg.POOL_SCALE
    return int(math.ceil(max_w / q) * q)


def groupBatch(imgs, labels):
    max_w = -sys.maxsize
    time_steps, label_len, label_vec, img_batch = [], [], [], []
    nh = cfg.IMG_HEIGHT
    for i, img in enumerate(imgs):
        h, w = img.shape[:2]
        nw = int(nh / h * w)
        max_w = max(max_w, nw)
        imgs[i] = _resize(img, nw, nh)
        time_steps.append(nw // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP)
        label_vec.extend(encode_maps[c] for c in labels[i])
        label_len.append(len(labels[i]))
    max_w = padded_width(max_w)
    for img in imgs:
        w = img.shape[1]
        pad = [(0, 0), (0, max_w - w)] + [(0, 0)] * (img.ndim - 2)
        padded = np.pad(img, pad, mode='constant', constant_values=0).astype(np.float32) / 255.
        img_batch.append(np.reshape(padded.swapaxes(0, 1), [-1, cfg.NUM_FEATURES]))
    return img_batch, label_vec, label_len, time_steps


def generator(batch_size=32, vis=False, min_len=None, max_len=None, width=160, px_per_char=None):
    """Endless stream of groupBatch() tuples.  px_per_char: canvas width follows the label (len * px_per_char) instead of
    being fixed — variable-width batches (BASELINE.json configs[3])."""
    images, labels = [], []
    while True:
        try:
            im, label = sample_image(min_len, max_len, width, px_per_char)
            images.append(im)
            labels.append(label)
            if len(images) == batch_size:
                yield groupBatch(images, labels)
                images, labels = [], []
        except Exception as e:                                  # the reference swallows and continues (gen.py:106-110)
            print(e)
            import traceback
            traceback.print_exc()
            continue


def stream_seed(rank=None, stream=0):
    """Seed of one data stream: cfg.RNG_SEED, made distinct per data-parallel rank ($RANK) and per stream kind (0 = training,
    1 = validation).  Workers add their index (data_util.GeneratorEnqueuer), so the strides keep every (rank, stream, worker)
    triple apart."""
    from ..dist import rank_seed
    rank = int(os.environ.get('RANK', '0')) if rank is None else rank
    return rank_seed(cfg.RNG_SEED, rank) + 7919 * int(stream)


def get_batch(num_workers, seed=None, **kwargs):
    """Multiprocess prefetch of generator(**kwargs) batches (gen.py:112-128).  seed: base seed of the workers
    (default: stream_seed() — rank-dependent, so data-parallel replicas draw different samples)."""
    enqueuer = None
    try:
        enqueuer = GeneratorEnqueuer(generator(**kwargs), use_multiprocessing=True,
                                     random_seed=stream_seed() if seed is None else seed)
        enqueuer.start(max_queue_size=24, workers=num_workers)
        generator_output = None
        while True:
            while enqueuer.is_running():
                if not enqueuer.queue.empty():
                    generator_output = enqueuer.queue.get()
                    break
                else:
                    time.sleep(0.01)
            yield generator_output
            generator_output = None
    finally:
        if enqueuer is not None:
            enqueuer.stop()
