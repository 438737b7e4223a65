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


def _font(size):
    key = (cfg.FONT, size)
    if key not in _FONT_CACHE:
        path = cfg.FONT if os.path.isabs(cfg.FONT) else os.path.join(cfg.ROOT_DIR, cfg.FONT)
        try:
            _FONT_CACHE[key] = ImageFont.truetype(path, size)
        except (IOError, OSError):
            try:
                _FONT_CACHE[key] = ImageFont.truetype("DejaVuSans.ttf", size)
            except (IOError, OSError):
                print('cannot open the font')
                _FONT_CACHE[key] = ImageFont.load_default()
    return _FONT_CACHE[key]


def randRGB():
    return (random.randint(0, 255), random.randint(0, 255), random.randint(0, 255))


def gen_rand():
    n = random.randint(cfg.MIN_LEN, cfg.MAX_LEN)
    return "".join(random.choice(cfg.CHARSET) for _ in range(n))


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


def generateImg():
    chars = gen_rand()
    return np.array(render_captcha(chars)), chars


def to_gray_reference(im_rgb):
    """cv2.cvtColor(im, COLOR_BGR2GRAY) applied to an RGB array (gen.py:77-78): 0.114 R + 0.587 G + 0.299 B."""
    w = np.array([0.114, 0.587, 0.299], np.float32)
    return np.clip(np.rint(im_rgb.astype(np.float32) @ w), 0, 255).astype(np.uint8)


def _resize(img, nw, nh):
    return np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))


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
    max_w = int(math.ceil(max_w / cfg.POOL_SCALE) * cfg.POOL_SCALE)
    for img in imgs:
        w = img.shape[1]
        pad = [(0, 0), (0, max_w - w)] + [(0, 0)] * (img.ndim - 2)
        padded = np.pad(img, pad, mode='constant', constant_values=0).astype(np.float32) / 255.
        img_batch.append(np.reshape(padded.swapaxes(0, 1), [-1, cfg.NUM_FEATURES]))
    return img_batch, label_vec, label_len, time_steps


def generator(batch_size=32, vis=False):
    images, labels = [], []
    while True:
        try:
            im, label = generateImg()
            if cfg.NCHANNELS == 1:
                im = to_gray_reference(im)
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


def get_batch(num_workers, **kwargs):
    enqueuer = None
    try:
        enqueuer = GeneratorEnqueuer(generator(**kwargs), use_multiprocessing=True, random_seed=cfg.RNG_SEED)
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
