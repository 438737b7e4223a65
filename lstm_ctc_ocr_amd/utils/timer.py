"""Stop-watch behind the `speed: {:.3f}s / iter` line of the training loop.

Public surface as in the reference's lib/lstm/utils/timer.py (tic(), toc(average=True), and the attributes total_time, calls,
start_time, diff, average_time that user code may read); implemented on the monotonic performance counter.
"""
from time import perf_counter


class Timer(object):
    __slots__ = ('total_time', 'calls', 'start_time', 'diff', 'average_time')

    def __init__(self):
        self.reset()

    def reset(self):
        self.total_time = self.start_time = self.diff = self.average_time = 0.0
        self.calls = 0

    def tic(self):
        """Start (or restart) an interval."""
        self.start_time = perf_counter()

    def toc(self, average=True):
        """Close the interval opened by tic(); returns the running mean of all intervals, or this one with average=False."""
        now = perf_counter()
        self.diff = now - self.start_time
        self.calls += 1
        self.total_time += self.diff
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
