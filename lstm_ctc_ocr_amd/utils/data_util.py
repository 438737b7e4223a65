"""Background batch producer with the interface of /root/reference/lib/utils/data_util.py:15-129
(GeneratorEnqueuer: start(workers, max_queue_size) / is_running() / stop() / .queue), re-implemented on
multiprocessing with deterministic per-worker seeding (the reference reseeds workers from the OS)."""
import multiprocessing
import queue as pyqueue
import random
import threading
import time

import numpy as np


class GeneratorEnqueuer(object):
    def __init__(self, generator, use_multiprocessing=False, wait_time=0.05, random_seed=None):
        self.wait_time = wait_time
        self._generator = generator
        self._use_multiprocessing = use_multiprocessing
        self._threads = []
        self._stop_event = None
        self.queue = None
        self.random_seed = random_seed

    def _worker(self, idx):
        if self.random_seed is not None:
            np.random.seed(self.random_seed + idx)
            random.seed(self.random_seed + idx)
        while not self._stop_event.is_set():
            try:
                if self._use_multiprocessing or self.queue.qsize() < self._max_queue_size:
                    self.queue.put(next(self._generator))
                else:
                    time.sleep(self.wait_time)
            except Exception:
                self._stop_event.set()
                raise

    def start(self, workers=1, max_queue_size=10):
        self._max_queue_size = max_queue_size
        try:
            if self._use_multiprocessing:
                ctx = multiprocessing.get_context('fork')
                self.queue = ctx.Queue(maxsize=max_queue_size)
                self._stop_event = ctx.Event()
                mk = lambda i: ctx.Process(target=self._worker, args=(i,), daemon=True)
            else:
                self.queue = pyqueue.Queue()
                self._stop_event = threading.Event()
                mk = lambda i: threading.Thread(target=self._worker, args=(i,), daemon=True)
            for i in range(workers):
                t = mk(i)
                self._threads.append(t)
                t.start()
        except Exception:
            self.stop()
            raise

    def is_running(self):
        return self._stop_event is not None and not self._stop_event.is_set()

    def stop(self, timeout=None):
        if self.is_running():
            self._stop_event.set()
        for t in self._threads:
            if t.is_alive():
                if self._use_multiprocessing:
                    t.terminate()
                else:
                    t.join(timeout)
        if self._use_multiprocessing and self.queue is not None:
            self.queue.close()
        self._threads = []
        self._stop_event = None
        self.queue = None
