"""Wall-clock timer with the tic/toc contract of /root/reference/lib/lstm/utils/timer.py:10-32 — the reference's only
performance instrument (it feeds the `speed: {:.3f}s / iter` line of train.py:135-138)."""
import time


class Timer(object):
    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
