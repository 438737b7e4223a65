"""GPU-side captcha synthesis: the training batches of utils/gen.py rendered by ONE HIP launch instead of PIL worker processes
(SURVEY.md 7 "GPU-side synthesis", 8 f2; reference: /root/reference/lib/lstm/utils/gen.py:31-67 generateImg + groupBatch).

Why: PIL renders ~0.6-1.4 k captchas per second and core; the 16 cores of an MI355X box feed 0.42-0.53x of what ONE GPU trains
(profiles/r06_final9_cli_throughput_live.log), so the live training loop ran at the generator's pace.  What a captcha costs is pixel work —
five glyph rotations, a bicubic resize, a 3 x 3 filter, a bilinear resize — and none of it needs the host.

Split:
  host   draw_params(): every RANDOM NUMBER of gen.render_captcha_gray / gen.sample_image (string, colours, font sizes, angles, offsets,
         noise dots, noise arc) for a whole batch in a few vectorised numpy calls, plus the integer geometry that follows from them
         (rotated glyph boxes, pen positions, canvas width, output width, time steps, flat label vector) — ~100 words per image;
  device ocr_captcha_synth (csrc/captcha_synth.hip): one workgroup per image composes the glyph masks of a resident ATLAS (every
         character of cfg.CHARSET at the three font sizes, rendered once by FreeType on the host: gen._glyph_mask) through PIL's own
         arithmetic — affine bilinear transform in double, the paste's DIV255 blend, the resampler's 22-bit fixed-point coefficients,
         the float32 3 x 3 SMOOTH — in LDS, and writes the [W, 32] uint8 rows Engine.train_step binds.
The checker is PIL itself, driven with the SAME parameters (oracle/synth_ref.py, test infrastructure): tests/test_synth.py (numpy model of the
kernel against PIL, CPU) and tests/test_gpu_synth.py (the kernel against both).  The noise arc is the one primitive that is not PIL's algorithm
(a thin parametric ellipse outline instead of ImagingDrawArc's clip tree); the tests bound what that costs in pixels.
"""
import fcntl
import math
import os
import pickle
import queue
import select
import subprocess
import sys
import threading

import numpy as np

from ..config import cfg
from . import gen

SIZES = (42, 50, 56)
HEIGHT = 60
NDOTS = 30
HDR = 32                   # int32 words: width, canvas_w, nw_out, L, bg, fg, arc x0 y0 x1 y1, arc start, arc end, 4 reserved, 6 doubles (the
                           # arc's two end normals: arc_normals), 4 reserved
GLYPH_WORDS = 20           # atlas offset, mask w, mask h, pen x, pen y, rotated w, rotated h, reserved, 6 doubles (inverse affine map)
CANVAS_CAP = 1024          # widest canvas (sum of the rotated glyph widths) the kernel's LDS image holds
WIDTH_CAP = 600            # widest captcha (gen.sample_image caps px_per_char widths there)
TAPS_CAP = 32              # resampling taps per output column the kernel's coefficient table holds


class GlyphAtlas(object):
    """Coverage masks of every (character, size) pair, concatenated: data uint8 [total], table int32 [n_chars * 3][3] = (offset, w, h)."""

    def __init__(self, charset=None):
        self.charset = cfg.CHARSET if charset is None else charset
        chunks, rows, off = [], [], 0
        for ch in self.charset:
            for s in SIZES:
                m = np.array(gen._glyph_mask(ch, s), np.uint8)
                h, w = m.shape
                rows.append((off, w, h))
                chunks.append(m.reshape(-1))
                off += w * h
        pad = (-off) % 16
        self.data = np.concatenate(chunks + [np.zeros(pad, np.uint8)])
        self.table = np.array(rows, np.int32)
        self.max_rotated = int(max(math.ceil(math.hypot(w, h)) + 1 for _, w, h in rows))
        self.codes = np.array([gen.encode_maps[c] for c in self.charset], np.int32)

    def mask(self, index):
        off, w, h = self.table[index]
        return self.data[off:off + w * h].reshape(h, w)


def words_per_image(max_glyphs):
    return HDR + 2 * NDOTS + GLYPH_WORDS * max_glyphs


def _trunc(x):
    return np.trunc(x).astype(np.int64)


def _randint(rng, lo, hi, shape=None):
    """random.randint(lo, hi) with ARRAY bounds, vectorised: lo + floor(u * (hi - lo + 1)) (Generator.integers takes array bounds too, at ~50 ns per
    draw — the 30 noise dots of every image made it most of draw_params)."""
    lo, hi = np.asarray(lo), np.asarray(hi)
    u = rng.random(np.broadcast(lo, hi).shape if shape is None else shape)
    return lo + np.minimum(np.floor(u * (hi - lo + 1)).astype(np.int64), hi - lo)


def draw_params(rng, n, atlas, min_len=None, max_len=None, width=160, px_per_char=None, strings=True):
    """The random draws of n captchas (gen.sample_image + gen.render_captcha_gray, in their distributions) and the geometry that follows.
    rng: numpy Generator.  Returns a dict of arrays; 'packed' int32 [n, words_per_image(max_len)] is what the kernel reads."""
    lo = cfg.MIN_LEN if min_len is None else min_len
    hi = cfg.MAX_LEN if max_len is None else max_len
    G = int(hi)
    nchar = len(atlas.charset)
    L = rng.integers(lo, hi + 1, n)
    chars = rng.integers(0, nchar, (n, G))
    if px_per_char:
        widths = np.minimum(WIDTH_CAP, L * int(px_per_char) + rng.integers(-8, 9, n))
    else:
        widths = np.full(n, int(width), np.int64)
    gw = np.array(gen._GRAY_W)
    bg = rng.integers(238, 256, (n, 3))
    fg = rng.integers(10, 201, (n, 3))
    bgv = np.clip(np.rint(bg @ gw), 0, 255).astype(np.int64)
    fgv = np.clip(np.rint(fg @ gw), 0, 255).astype(np.int64)
    live = np.arange(G)[None, :] < L[:, None]
    mid = chars * 3 + rng.integers(0, 3, (n, G))
    angle = rng.uniform(-30.0, 30.0, (n, G))
    w = atlas.table[mid, 1].astype(np.float64)
    h = atlas.table[mid, 2].astype(np.float64)
    # Image.rotate(angle, BILINEAR, expand=1): gen._rotate_mask, vectorised
    a = -np.radians(angle % 360.0)
    ca, sa = np.round(np.cos(a), 15), np.round(np.sin(a), 15)
    m0, m1, m3, m4 = ca, sa, -sa, ca
    cx, cy = w / 2, h / 2
    m2 = m0 * -cx + m1 * -cy + 0.0 + cx
    m5 = m3 * -cx + m4 * -cy + 0.0 + cy
    xs = np.stack([m2, m0 * w + m2, m0 * w + m1 * h + m2, m1 * h + m2])
    ys = np.stack([m5, m3 * w + m5, m3 * w + m4 * h + m5, m4 * h + m5])
    nw = (np.ceil(xs.max(0)) - np.floor(xs.min(0))).astype(np.int64)
    nh = (np.ceil(ys.max(0)) - np.floor(ys.min(0))).astype(np.int64)
    tx, ty = -(nw - w) / 2.0, -(nh - h) / 2.0
    m2, m5 = m0 * tx + m1 * ty + m2, m3 * tx + m4 * ty + m5
    nw = np.where(live, nw, 0)
    text_w = nw.sum(1)
    avg = _trunc(text_w / np.maximum(1, L))
    q = _trunc(0.25 * avg)
    yj = rng.integers(-4, 5, (n, G))
    adv = _randint(rng, -q[:, None], 0, (n, G))
    pen_y = np.maximum(0, _trunc((HEIGHT - nh) / 2) + yj)
    pen_x = np.zeros((n, G), np.int64)
    x = _trunc(0.1 * avg)
    for g in range(G):
        pen_x[:, g] = x
        x = x + nw[:, g] + adv[:, g]
    canvas_w = np.maximum(text_w, widths)
    if canvas_w.max() > CANVAS_CAP:
        raise ValueError('captcha canvas of %d px: the synthesis kernel holds %d' % (canvas_w.max(), CANVAS_CAP))
    if widths.max() > WIDTH_CAP or widths.min() < 8:
        raise ValueError('captcha widths %d..%d outside [8, %d]' % (widths.min(), widths.max(), WIDTH_CAP))
    taps = 2 * np.ceil(2.0 * canvas_w / widths) + 1          # Resample.c: ksize of the bicubic pass canvas_w -> width
    if taps.max() > TAPS_CAP:
        raise ValueError('a %d px canvas on a %d px captcha needs %d resampling taps: the synthesis kernel holds %d'
                         % (canvas_w[taps.argmax()], widths[taps.argmax()], taps.max(), TAPS_CAP))
    dots = np.stack([_randint(rng, 0, widths[:, None], (n, NDOTS)), rng.integers(0, HEIGHT + 1, (n, NDOTS))], -1)
    fifth = _trunc(widths / 5)
    x1 = _randint(rng, 0, fifth)
    x2 = _randint(rng, widths - fifth, widths)
    y1 = rng.integers(HEIGHT // 5, HEIGHT - HEIGHT // 5 + 1, n)
    y2 = rng.integers(HEIGHT // 5, HEIGHT + 1, n)
    a_start = rng.integers(0, 21, n)
    a_end = rng.integers(160, 201, n)
    nw_out = _trunc(cfg.IMG_HEIGHT / HEIGHT * widths)
    steps = nw_out // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP

    S = words_per_image(G)
    packed = np.zeros((n, S), np.int32)
    hdr = np.stack([widths, canvas_w, nw_out, L, bgv, fgv, x1, np.minimum(y1, y2), x2, np.maximum(y1, y2) + 1, a_start, a_end], 1)
    packed[:, :hdr.shape[1]] = hdr
    packed[:, 16:28] = arc_normals(hdr[:, 6:10], a_start, a_end).view(np.int32).reshape(n, 12)
    packed[:, HDR:HDR + 2 * NDOTS] = dots.reshape(n, 2 * NDOTS)
    gl = packed[:, HDR + 2 * NDOTS:].reshape(n, G, GLYPH_WORDS)
    gl[:, :, 0] = atlas.table[mid, 0]
    gl[:, :, 1] = atlas.table[mid, 1]
    gl[:, :, 2] = atlas.table[mid, 2]
    gl[:, :, 3] = pen_x
    gl[:, :, 4] = pen_y
    gl[:, :, 5] = nw
    gl[:, :, 6] = np.where(live, nh, 0)
    mat = np.ascontiguousarray(np.stack([m0, m1, m2, m3, m4, m5], -1))             # [n, G, 6] float64
    gl[:, :, 8:20] = mat.view(np.int32).reshape(n, G, 12)
    labels = atlas.codes[chars][live]                       # flat, sample-major
    return {'packed': packed, 'labels': labels.astype(np.int32), 'labels_len': L.astype(np.int32), 'steps': steps.astype(np.int32),
            'nw_out': nw_out, 'widths': widths, 'canvas_w': canvas_w, 'chars': chars, 'L': L,
            'strings': [''.join(atlas.charset[c] for c in chars[i, :L[i]]) for i in range(n)] if strings else None, 'max_glyphs': G}


def arc_normals(box, start, end):
    """[n, 6] float64: the lines the noise arc ends at — the normals of the ellipse inscribed in box [n, 4] at its points of eccentric anomaly
    start / end degrees (PIL's Draw.c arc_init clips the outline with them), as (cu, cv, c0) with cu * u + cv * v + c0 >= 0 on the arc's side,
    (u, v) = pixel - centre."""
    box = np.asarray(box, np.float64)
    a, b = (box[:, 2] - box[:, 0]) / 2.0, (box[:, 3] - box[:, 1]) / 2.0
    al, ar = np.radians(np.asarray(start, np.float64)), np.radians(np.asarray(end, np.float64))
    d = a * a - b * b
    return np.ascontiguousarray(np.stack([-a * np.sin(al), b * np.cos(al), d * np.sin(al) * np.cos(al),
                                          a * np.sin(ar), -b * np.cos(ar), -d * np.sin(ar) * np.cos(ar)], 1))


def unpack_image(packed_row, max_glyphs):
    """One image's parameters back as a dict (the checker oracle/synth_ref.py and the numpy model tools/synth_model.py read this)."""
    p = np.asarray(packed_row, np.int32)
    d = dict(width=int(p[0]), canvas_w=int(p[1]), nw_out=int(p[2]), L=int(p[3]), bg=int(p[4]), fg=int(p[5]),
             arc=(int(p[6]), int(p[7]), int(p[8]), int(p[9])), arc_start=int(p[10]), arc_end=int(p[11]),
             arc_lines=tuple(np.ascontiguousarray(p[16:28]).view(np.float64).tolist()))
    d['dots'] = p[HDR:HDR + 2 * NDOTS].reshape(NDOTS, 2).astype(int)
    gl = p[HDR + 2 * NDOTS:].reshape(max_glyphs, GLYPH_WORDS)
    d['glyphs'] = []
    for g in range(d['L']):
        r = gl[g]
        d['glyphs'].append(dict(off=int(r[0]), mw=int(r[1]), mh=int(r[2]), x=int(r[3]), y=int(r[4]), nw=int(r[5]), nh=int(r[6]),
                                mat=tuple(np.ascontiguousarray(r[8:20]).view(np.float64).tolist())))
    return d


def dot_stamp():
    """Pixels ImageDraw's 3-wide line from (x, y) to (x - 1, y - 1) sets, relative to (x, y) (the installed PIL decides; the kernel takes the
    list as an argument)."""
    from PIL import Image, ImageDraw
    im = Image.new('L', (16, 16), 0)
    ImageDraw.Draw(im).draw.draw_lines(((8, 8), (7, 7)), 255, 3)
    ys, xs = np.nonzero(np.array(im))
    return np.stack([xs - 8, ys - 8], 1).astype(np.int32)


MSG_HDR = 4                # int32 words in front of a batch message: W, canvas_cap, width_cap, number of labels


def batch_words(B, G):
    """int32 words of one batch message / device block: header | B parameter records | flat labels (room for B * G) | label_len[B] | steps[B]"""
    return MSG_HDR + B * words_per_image(G) + B * G + 2 * B


def fill_batches(p, B, G, out):
    """Split the draws of k * B images (draw_params) into k batch blocks out[k][batch_words(B, G)] (int32)."""
    S = words_per_image(G)
    ends = np.cumsum(p['labels_len'])
    for c in range(out.shape[0]):
        s = slice(c * B, (c + 1) * B)
        l0 = int(ends[c * B - 1]) if c else 0
        l1 = int(ends[(c + 1) * B - 1])
        m = out[c]
        m[0], m[1], m[2], m[3] = gen.padded_width(int(p['nw_out'][s].max())), int(p['canvas_w'][s].max()), int(p['widths'][s].max()), l1 - l0
        o = MSG_HDR + B * S
        m[MSG_HDR:o] = p['packed'][s].reshape(-1)
        m[o:o + l1 - l0] = p['labels'][l0:l1]
        m[o + B * G:o + B * G + B] = p['labels_len'][s]
        m[o + B * G + B:o + B * G + 2 * B] = p['steps'][s]


def _param_worker(send, seed, B, G, chunk, atlas, kw):
    """Worker loop: draws parameters, `chunk` batches per numpy pass, and hands one int32 message per batch to `send` (a pipe write that blocks
    when the reader is behind: the back-pressure)."""
    try:
        rng = np.random.default_rng(seed)
        out = np.zeros((chunk, batch_words(B, G)), np.int32)
        while True:
            fill_batches(draw_params(rng, B * chunk, atlas, strings=False, **kw), B, G, out)
            for c in range(chunk):
                send(out[c])
    except (BrokenPipeError, EOFError, KeyboardInterrupt, OSError):
        pass
    except BaseException:
        import traceback
        traceback.print_exc()
        raise


_WORKER_CFG_KEYS = ('IMG_HEIGHT', 'POOL_SCALE', 'OFFSET_TIME_STEP', 'MIN_LEN', 'MAX_LEN')


def _worker_main():
    """`python -m lstm_ctc_ocr_amd.utils.synth`: a parameter worker of DeviceSynthStream as a process of its own.  The job (seed, batch geometry,
    the glyph atlas, the few cfg values draw_params reads) arrives pickled on stdin, the batch messages leave as raw int32 on stdout.  A fresh
    interpreter instead of a fork of the training process: forking a process that has tens of GB of device memory mapped copies its page tables
    (8 s per worker inside the GPU test-suite's process), and the child would carry the HIP runtime's state for nothing."""
    import pickle
    import sys
    job = pickle.load(sys.stdin.buffer)
    for k, v in job['cfg'].items():
        setattr(cfg, k, v)
    out = sys.stdout.buffer

    def send(row):
        out.write(memoryview(row).cast('B'))
        out.flush()
    _param_worker(send, job['seed'], job['B'], job['G'], job['chunk'], job['atlas'], job['kw'])


def spawn_worker(seed, B, G, chunk, atlas, kw):
    """Start one parameter worker (_worker_main in a fresh interpreter); its stdout delivers batch_words(B, G) int32 words per batch."""
    pkg_parent = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    # one thread per worker: a fresh interpreter's BLAS / OpenMP pools would spin on the cores the training process's launch loop needs
    env = dict(os.environ, PYTHONPATH=pkg_parent + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1',
               MKL_NUM_THREADS='1')
    pr = subprocess.Popen([sys.executable, '-m', 'lstm_ctc_ocr_amd.utils.synth'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=0, env=env)
    try:
        fcntl.fcntl(pr.stdout.fileno(), 1031, 1 << 20)                 # F_SETPIPE_SZ: room for a dozen batches ahead of the feeder
    except OSError:
        pass
    pickle.dump({'seed': seed, 'B': B, 'G': G, 'chunk': chunk, 'atlas': atlas, 'kw': kw, 'cfg': {k: cfg[k] for k in _WORKER_CFG_KEYS}}, pr.stdin)
    pr.stdin.close()
    return pr


class DeviceSynthStream(object):
    """Iterator of device-resident batches — the tuples utils.pipeline.DeviceBatchStream yields: (pixels uint8 [B, W, 32], labels int32 [n],
    label_len int32 [B], steps int32 [B]) — synthesised on the GPU.  `workers` processes (fresh interpreters: _worker_main) draw the parameters
    (~0.3 ms of numpy per batch: drawn on a thread of the training process they held its GIL in bursts of several ms and the launch loop stalled
    behind them); a feeder thread reads their messages round-robin straight into pinned
    memory, copies them on a side stream and launches the synthesis kernel there, `depth` batches ahead of the training stream.  A returned
    batch stays valid until the NEXT call of next().  Deterministic for a given (seed, workers, chunk)."""

    def __init__(self, device, batch_size, depth=4, seed=None, chunk=8, workers=2, **gen_kwargs):
        import torch
        from .. import ops
        self.torch, self.ops = torch, ops
        self.device = torch.device(device)
        self.B = int(batch_size)
        self.kw = dict(min_len=gen_kwargs.get('min_len'), max_len=gen_kwargs.get('max_len'), width=gen_kwargs.get('width', 160),
                       px_per_char=gen_kwargs.get('px_per_char'))
        self.atlas = GlyphAtlas()
        self.G = int(cfg.MAX_LEN if self.kw['max_len'] is None else self.kw['max_len'])
        seed = gen.stream_seed() if seed is None else seed
        self.d_atlas = torch.from_numpy(self.atlas.data).to(self.device)
        self.d_stamp = torch.from_numpy(dot_stamp().reshape(-1)).to(self.device)
        self.S = words_per_image(self.G)
        self.max_w = gen.padded_width(int(cfg.IMG_HEIGHT / HEIGHT * (WIDTH_CAP if self.kw['px_per_char'] else self.kw['width'])))    # widest padded batch
        nmeta = batch_words(self.B, self.G)
        self.depth = depth
        self.procs = [spawn_worker(seed + 1000003 * i, self.B, self.G, max(1, int(chunk)), self.atlas, self.kw) for i in range(max(1, int(workers)))]
        self.h_meta = [torch.empty(nmeta, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.d_meta = [torch.empty(nmeta, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.d_pix = [torch.empty(self.B * self.max_w * cfg.NUM_FEATURES, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.filled = [torch.cuda.Event() for _ in range(depth)]
        self.side = torch.cuda.Stream(device=self.device)
        self.free = queue.Queue()
        for k in range(depth):
            self.free.put((k, None))
        self.staged = queue.Queue()
        self.halt = threading.Event()
        self.error = None
        self._last = None
        self.thread = threading.Thread(target=self._feed, daemon=True)
        self.thread.start()

    def _feed(self):
        torch = self.torch
        B, S, G = self.B, self.S, self.G
        try:
            torch.cuda.set_device(self.device)
            turn = 0
            while not self.halt.is_set():
                try:
                    k, done = self.free.get(timeout=0.2)
                except queue.Empty:
                    continue
                if done is not None:
                    done.synchronize()                       # everything that read this buffer (and its pinned source) has finished
                hm = self.h_meta[k].numpy()
                pr = self.procs[turn % len(self.procs)]
                turn += 1
                mv, got = memoryview(hm).cast('B'), 0
                while got < len(mv):
                    ready, _, _ = select.select([pr.stdout], [], [], 0.2)
                    if self.halt.is_set():
                        return
                    if not ready:
                        continue
                    n = pr.stdout.readinto(mv[got:])
                    if not n:
                        raise RuntimeError('a synthesis parameter worker died (exit status %s; its traceback is on stderr)' % pr.poll())
                    got += n
                W, ccap, wcap, nlab = int(hm[0]), int(hm[1]), int(hm[2]), int(hm[3])
                with torch.cuda.stream(self.side):
                    self.d_meta[k].copy_(self.h_meta[k], non_blocking=True)
                    self.ops.captcha_synth(self.d_meta[k][MSG_HDR:], B, S, self.d_atlas, self.d_stamp, self.d_pix[k], W, stream=self.side, max_glyphs=G,
                                           canvas_cap=ccap, width_cap=wcap, out_h=cfg.IMG_HEIGHT)
                    self.filled[k].record(self.side)
                self.staged.put((k, W, nlab))
        except Exception as e:                               # surface in the consumer
            self.error = e

    def __iter__(self):
        return self

    def __next__(self):
        torch = self.torch
        while True:
            if self.error is not None:
                raise self.error
            try:
                k, W, nlab = self.staged.get(timeout=0.5)
                break
            except queue.Empty:
                continue
        if self._last is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.free.put((self._last, ev))
        self._last = k
        torch.cuda.current_stream(self.device).wait_event(self.filled[k])
        B, S, G = self.B, self.S, self.G
        meta = self.d_meta[k]
        o = MSG_HDR + B * S
        pix = self.d_pix[k][:B * W * cfg.NUM_FEATURES].view(B, W, cfg.NUM_FEATURES)
        return pix, meta[o:o + nlab], meta[o + B * G:o + B * G + B], meta[o + B * G + B:o + B * G + 2 * B]

    def close(self):
        """Stop the feeder thread and the parameter workers (idempotent)."""
        self.halt.set()
        if self.thread.is_alive():
            self.thread.join(timeout=2.0)
        procs, self.procs = self.procs, []
        for pr in procs:
            pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=2.0)
            except subprocess.TimeoutExpired:
                pr.kill()
            pr.stdout.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


if __name__ == '__main__':
    _worker_main()
