"""Background batch producer.

Interface of the reference's lib/utils/data_util.py:15-129 — GeneratorEnqueuer(generator, use_multiprocessing, wait_time,
random_seed) with start(workers, max_queue_size), is_running(), stop(timeout) and a public `.queue` — because get_batch() and
user code poll `enqueuer.queue` directly.  Own implementation: one `_Backend` (threads or forked processes) supplies the
queue, the stop flag and the worker factory, every worker is seeded `random_seed + worker index` (the reference reseeds its
processes from the OS, so its data stream is not reproducible — SURVEY Q9), and a bounded queue does the back-pressure.
"""
import multiprocessing
import queue
import random
import threading
import time

import numpy as np


class _Backend(object):
    """Queue / event / worker constructors for one of the two execution modes."""

    def __init__(self, processes, capacity):
        self.processes = processes
        if processes:
            ctx = multiprocessing.get_context('fork')       # the generator object is inherited, not pickled
            self.queue, self.halt = ctx.Queue(maxsize=capacity), ctx.Event()
            self._spawn = lambda fn, i: ctx.Process(target=fn, args=(i,), daemon=True)
        else:
            self.queue, self.halt = queue.Queue(maxsize=capacity), threading.Event()
            self._spawn = lambda fn, i: threading.Thread(target=fn, args=(i,), daemon=True)

    def launch(self, fn, count):
        workers = [self._spawn(fn, i) for i in range(count)]
        for w in workers:
            w.start()
        return workers

    def retire(self, worker, timeout):
        if not worker.is_alive():
            return
        if self.processes:
            worker.terminate()
        else:
            worker.join(timeout)


class GeneratorEnqueuer(object):
    def __init__(self, generator, use_multiprocessing=False, wait_time=0.05, random_seed=None):
        self._generator = generator
        self._use_multiprocessing = use_multiprocessing
        self.wait_time = wait_time
        self.random_seed = random_seed
        self._backend, self._workers = None, []
        self.queue = None

    # -- worker side -----------------------------------------------------------------------------------------------------
    def _produce(self, index):
        backend = self._backend
        if self.random_seed is not None:
            random.seed(self.random_seed + index)
            np.random.seed(self.random_seed + index)
        try:
            while not backend.halt.is_set():
                item = next(self._generator)
                while not backend.halt.is_set():            # bounded queue: wait for room, stay responsive to stop()
                    try:
                        backend.queue.put(item, timeout=self.wait_time)
                        break
                    except queue.Full:
                        continue
        except StopIteration:
            backend.halt.set()
        except Exception:
            backend.halt.set()
            raise

    # -- owner side ------------------------------------------------------------------------------------------------------
    def start(self, workers=1, max_queue_size=10):
        self._backend = _Backend(self._use_multiprocessing, max_queue_size)
        self.queue = self._backend.queue
        try:
            self._workers = self._backend.launch(self._produce, workers)
        except Exception:
            self.stop()
            raise

    def is_running(self):
        return self._backend is not None and not self._backend.halt.is_set()

    def stop(self, timeout=None):
        backend = self._backend
        if backend is None:
            return
        backend.halt.set()
        for w in self._workers:
            backend.retire(w, timeout)
        if backend.processes:
            backend.queue.close()
        self._backend, self._workers, self.queue = None, [], None

    def get(self):
        """Generator over the queued items while the enqueuer runs (data_util.py:115-128): `None` items are skipped, an empty
        queue is polled every `wait_time` seconds; ends when stop() is called, or when the source generator is exhausted and the
        queue has been drained."""
        while True:
            q, running = self.queue, self.is_running()
            if q is None:
                return
            try:
                item = q.get(timeout=self.wait_time)
            except queue.Empty:
                if not running:             # source exhausted and queue drained (the reference drops what is still queued)
                    return
                continue
            if item is not None:
                yield item
