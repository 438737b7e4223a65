"""Prefetching input pipeline for the training driver.

What the reference does (lib/lstm/train.py:64,118-127, lib/lstm/utils/gen.py:112-128, lib/utils/data_util.py): 12 forked
generator processes pickle float32 batches through a multiprocessing.Queue; the training loop pulls one, converts lists to
arrays and feeds them — every byte crosses a pipe, is unpickled on the training thread and copied host-to-device from
pageable memory while the GPU waits.  Round 1 kept that shape and measured the real CLI at 15 k images/s against a device
rate of 40 k-100 k: the host, not the GPU, set the pace.

This module keeps the generator (same rendering, same groupBatch layout, same label encoding) and replaces the transport:

  workers (fork)  --render-->  SHARED-MEMORY RING of batch slots (uint8 pixels + int32 labels / lengths / steps, written in
                               place, never pickled; only slot numbers travel through the queues)
  feeder thread   --H2D----->  the ring is registered with the HIP runtime as pinned memory, so one asynchronous copy per
                               slot on a COPY STREAM moves it into one of a few device staging buffers while the previous
                               step is still running; an event per buffer tells the training stream when it may read
  training loop   <----------  device-resident (uint8 pixels, labels, label lengths, time steps) tuples; Engine.train_step
                               divides the pixels by 255 on the device (bit-identical to the host's astype(float32) / 255.)

Slots return to the workers as soon as their copy has completed; at most `depth` batches are resident on the device ahead
of the step that is running.
POOL mode (`pool=N`, $OCR_DATA_POOL): the workers render N batches once and exit; the stream then cycles over that fixed
dataset in a fresh random order per pass (the H2D path is unchanged).  PIL renders ~0.6-1.2 k captchas per second and core,
so a live stream needs ~40-100 cores to keep up with ONE MI355X (40-60 k images/s); on hosts with fewer cores the pool is
what lets the training loop run at the device's pace.  Seeds: utils.gen.stream_seed() (rank- and stream-dependent) + worker index.
The host part (ring, workers, batch layout) needs no GPU and is covered by the CPU test-suite.
"""
import mmap
import multiprocessing
import os
import queue
import random
import threading

import numpy as np

from ..config import cfg
from . import gen

_HDR = 4            # int32 header words per slot: W, number of labels, batch size, reserved


class SlotLayout(object):
    """Byte layout of one batch slot: header | label_len[B] | steps[B] | labels[B * max_label] | pixels uint8 [B, W, 32]."""

    def __init__(self, batch, max_w, max_label):
        self.batch, self.max_w, self.max_label = int(batch), int(max_w), int(max_label)
        self.meta_words = _HDR + 2 * self.batch + self.batch * self.max_label
        self.meta_bytes = (self.meta_words * 4 + 255) // 256 * 256
        self.pix_bytes = (self.batch * self.max_w * cfg.NUM_FEATURES + 255) // 256 * 256
        self.bytes = self.meta_bytes + self.pix_bytes

    def views(self, buf, slot):
        base = slot * self.bytes
        meta = np.frombuffer(buf, np.int32, self.meta_words, base)
        pix = np.frombuffer(buf, np.uint8, self.pix_bytes, base + self.meta_bytes)
        return meta, pix


def group_batch_u8(imgs, labels, layout, meta, pix):
    """gen.groupBatch (reference gen.py:41-67) writing uint8 pixels and int32 labels straight into a slot.  Returns W."""
    nh = cfg.IMG_HEIGHT
    B = len(imgs)
    resized, max_w = [], 0
    lab = []
    for i, img in enumerate(imgs):
        h, w = img.shape[:2]
        nw = int(nh / h * w)
        max_w = max(max_w, nw)
        resized.append(gen._resize(img, nw, nh))
        meta[_HDR + B + i] = nw // cfg.POOL_SCALE + cfg.OFFSET_TIME_STEP
        code = [gen.encode_maps[c] for c in labels[i]]
        meta[_HDR + i] = len(code)
        lab.extend(code)
    W = gen.padded_width(max_w)                      # POOL_SCALE, or the opt-in OCR_WIDTH_BUCKET
    if W > layout.max_w or len(lab) > B * layout.max_label:
        raise ValueError('batch does not fit its slot (W %d > %d or %d labels)' % (W, layout.max_w, len(lab)))
    out = pix[:B * W * cfg.NUM_FEATURES].reshape(B, W, cfg.NUM_FEATURES)
    out[:] = 0                                          # right padding with 0 (gen.py:62)
    for i, img in enumerate(resized):
        out[i, :img.shape[1], :] = img.T                # [H, w] -> [w, H]: the swapaxes of gen.py:63-64
    meta[_HDR + 2 * B:_HDR + 2 * B + len(lab)] = lab
    meta[0], meta[1], meta[2] = W, len(lab), B
    return W


def _worker(index, seed, layout, buf, free_q, ready_q, halt, gen_kwargs, once=False):
    random.seed(seed + index)
    np.random.seed(seed + index)
    try:
        while not halt.is_set():
            try:
                slot = free_q.get(timeout=0.1)
            except queue.Empty:
                if once:
                    return                              # pool mode: every slot has been rendered
                continue
            images, labels = [], []
            while len(images) < layout.batch:
                im, chars = gen.sample_image(gen_kwargs.get('min_len'), gen_kwargs.get('max_len'), gen_kwargs.get('width', 160),
                                             gen_kwargs.get('px_per_char'))          # the same call, hence the same RNG order, as gen.generator
                images.append(im)
                labels.append(chars)
            meta, pix = layout.views(buf, slot)
            group_batch_u8(images, labels, layout, meta, pix)
            ready_q.put(slot)
    except KeyboardInterrupt:
        pass
    except BaseException as e:                          # a batch that does not fit its slot, a PIL error, ...: tell the consumer
        import traceback                                # instead of dying silently (the ring would then starve and the training
        ready_q.put(('error', 'generator worker %d: %r\n%s' % (index, e, traceback.format_exc())))     # loop hang — ADVICE r2)
        raise


class SharedBatchRing(object):
    """Worker processes + shared-memory slots.  Iterating yields slot numbers; slot(i) gives numpy views; release(i) returns it."""

    def __init__(self, batch_size, workers, slots=None, max_w=None, max_label=None, seed=None, pool=0, **gen_kwargs):
        max_len = gen_kwargs.get('max_len') or cfg.MAX_LEN
        if max_w is None:
            canvas = 600 if gen_kwargs.get('px_per_char') else gen_kwargs.get('width', 160)
            max_w = gen.padded_width(int(cfg.IMG_HEIGHT / 60.0 * canvas + 1))
        self.layout = SlotLayout(batch_size, max_w, max_label or max_len)
        self.pool = int(pool)
        self.nslots = self.pool or slots or max(4, 2 * workers)
        self._order, self._seen, self._rng = [], 0, random.Random(12345 + (seed or 0))
        self.buf = mmap.mmap(-1, self.nslots * self.layout.bytes)          # anonymous shared mapping: inherited by fork
        ctx = multiprocessing.get_context('fork')
        self.free_q, self.ready_q, self.halt = ctx.Queue(), ctx.Queue(), ctx.Event()
        for i in range(self.nslots):
            self.free_q.put(i)
        seed = gen.stream_seed() if seed is None else seed
        gen.resolve_font()                                                   # announce a font substitution once, in the parent
        self.procs = [ctx.Process(target=_worker, args=(i, seed, self.layout, self.buf, self.free_q, self.ready_q, self.halt, gen_kwargs,
                                                        bool(self.pool)), daemon=True) for i in range(workers)]
        for p in self.procs:
            p.start()

    def slot(self, i):
        meta, pix = self.layout.views(self.buf, i)
        W, nlab, B = int(meta[0]), int(meta[1]), int(meta[2])
        return dict(W=W, B=B, nlab=nlab, label_len=meta[_HDR:_HDR + B], steps=meta[_HDR + B:_HDR + 2 * B],
                    labels=meta[_HDR + 2 * B:_HDR + 2 * B + nlab], pixels=pix[:B * W * cfg.NUM_FEATURES].reshape(B, W, cfg.NUM_FEATURES),
                    meta=meta, pix=pix)

    def get(self, timeout=None):
        if self.pool and self._seen >= self.pool:       # every pool batch has been rendered: cycle, reshuffled per pass
            if not self._order:
                self._order = list(range(self.pool))
                self._rng.shuffle(self._order)
            return self._order.pop()
        try:
            i = self.ready_q.get(timeout=timeout)
        except queue.Empty:
            self._check_workers()
            raise
        if isinstance(i, tuple):                        # ('error', text) from a worker
            raise RuntimeError(i[1])
        self._seen += 1
        return i

    def _check_workers(self):
        """Called when no batch arrived in time: a worker that was killed (out of memory, a signal) cannot report itself."""
        codes = [p.exitcode for p in self.procs]
        dead = [(k, c) for k, c in enumerate(codes) if c not in (None, 0)]
        if dead:
            raise RuntimeError('generator worker(s) died: %s' % ', '.join('#%d exit code %s' % kc for kc in dead))
        if self.procs and all(c is not None for c in codes) and self.ready_q.empty():
            if self.pool and self._seen < self.pool:
                raise RuntimeError('generator workers finished after %d of %d pool batches' % (self._seen, self.pool))
            if not self.pool:
                raise RuntimeError('all generator workers have exited')

    def release(self, i):
        if not self.pool:
            self.free_q.put(i)

    def close(self):
        self.halt.set()
        for p in self.procs:
            p.join(timeout=1.0)
            if p.is_alive():
                p.terminate()
        self.procs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBatchStream(object):
    """Iterator of device-resident batches: (pixels uint8 [B, W, 32], labels int32 [n], label_len int32 [B], steps int32 [B]).
    `depth` staging buffers; the feeder thread keeps them full, copies run on their own stream.  A returned batch stays valid until
    the NEXT call of next() (its buffer is handed back to the feeder then)."""

    def __init__(self, device, batch_size, workers=None, depth=4, seed=None, pool=None, **gen_kwargs):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        world = int(os.environ.get('WORLD_SIZE', '1'))
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
        if workers is None:               # leave a few cores to the drivers; share the host between the ranks of a node
            workers = max(2, min(96, (cores - 2 * world) // world))
        if pool is None:
            pool = int(os.environ.get('OCR_DATA_POOL', '0'))
        self.ring = SharedBatchRing(batch_size, workers, seed=seed, pool=pool, **gen_kwargs)
        lay = self.ring.layout
        self._register(self.ring.buf)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self.d_meta = [torch.empty(lay.meta_words, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.d_pix = [torch.empty(lay.pix_bytes, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.filled = [torch.cuda.Event() for _ in range(depth)]         # copy finished -> training stream may read
        # Buffer ownership is explicit: the feeder may only fill a buffer it took from `free`, and a buffer only gets there when the
        # consumer has moved on to the NEXT batch, together with an event recorded on the training stream behind everything that
        # read it.  (The first version looked at a per-buffer "consumed" event that was still unset — or left over from the
        # previous lap — while the consumer was about to bind that very buffer: the feeder overwrote batches that had been handed
        # out but not read yet; one step in a few thousand trained on pixels of one batch with labels of another.)
        self.free = queue.Queue()
        for k in range(depth):
            self.free.put((k, None))
        self.staged = queue.Queue()
        self.halt = threading.Event()
        self.error = None
        self.thread = threading.Thread(target=self._feed, daemon=True)
        self.thread.start()
        self._last = None

    def _register(self, buf):
        """Pin the ring: the H2D copies then are real DMA transfers that return at once instead of staged, blocking ones."""
        import ctypes
        torch = self.torch
        addr = ctypes.addressof(ctypes.c_char.from_buffer(buf))
        rc = torch.cuda.cudart().cudaHostRegister(addr, len(buf), 0)
        self.pinned = (int(rc) == 0)
        self._host = torch.frombuffer(buf, dtype=torch.uint8)

    def _feed(self):
        torch = self.torch
        lay = self.ring.layout
        try:
            torch.cuda.set_device(self.device)
            while not self.halt.is_set():
                try:
                    k, done = self.free.get(timeout=0.2)                  # a device buffer the consumer has let go of
                except queue.Empty:
                    continue
                slot = None
                while slot is None and not self.halt.is_set():
                    try:
                        slot = self.ring.get(timeout=0.2)
                    except queue.Empty:
                        continue
                if slot is None:
                    break
                info = self.ring.slot(slot)
                if done is not None:
                    done.synchronize()                                    # everything that read this buffer has finished
                base = slot * lay.bytes
                npix = info['B'] * info['W'] * cfg.NUM_FEATURES
                with torch.cuda.stream(self.copy_stream):
                    self.d_meta[k].copy_(self._host[base:base + lay.meta_words * 4].view(torch.int32), non_blocking=True)
                    self.d_pix[k][:npix].copy_(self._host[base + lay.meta_bytes:base + lay.meta_bytes + npix], non_blocking=True)
                    self.filled[k].record(self.copy_stream)
                self.filled[k].synchronize()                              # slot may go back to the workers
                self.ring.release(slot)
                self.staged.put((k, info['B'], info['W'], info['nlab']))
        except Exception as e:                                            # surface in the consumer
            self.error = e

    def __iter__(self):
        return self

    def __next__(self):
        torch = self.torch
        while True:
            if self.error is not None:
                raise self.error
            try:
                k, B, W, nlab = self.staged.get(timeout=0.5)
                break
            except queue.Empty:
                continue
        if self._last is not None:                   # the caller asks for the next batch: the previous one has been bound (or dropped),
            ev = torch.cuda.Event()                  # so its buffer goes back to the feeder, fenced behind the work issued so far
            ev.record(torch.cuda.current_stream(self.device))
            self.free.put((self._last, ev))
        self._last = k
        torch.cuda.current_stream(self.device).wait_event(self.filled[k])
        meta = self.d_meta[k]
        pix = self.d_pix[k][:B * W * cfg.NUM_FEATURES].view(B, W, cfg.NUM_FEATURES)
        return pix, meta[_HDR + 2 * B:_HDR + 2 * B + nlab], meta[_HDR:_HDR + B], meta[_HDR + B:_HDR + 2 * B]

    def close(self):
        self.halt.set()
        self.thread.join(timeout=2.0)
        self.ring.close()
