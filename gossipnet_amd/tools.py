"""`nms_net.tools.Timer` for callers of the reference (test.py:56,68,71 brackets `sess.run` with tic() / toc();
reference nms_net/tools.py:11-35).  Same attributes and return values; what is timed is up to the caller: the kernels
behind Gnet.run are asynchronous, so pass `sync=torch.cuda.synchronize` to time completed work rather than launches.
"""
import time


class Timer(object):
    __slots__ = ("calls", "total_time", "start_time", "diff", "_sync")

    def __init__(self, sync=None):
        self.calls, self.total_time, self.start_time, self.diff = 0, 0.0, 0.0, 0.0
        self._sync = sync

    @property
    def average_time(self):
        return self.total_time / self.calls if self.calls else 0.0

    def _now(self):
        if self._sync is not None:
            self._sync()
        return time.time()          # (wall clock, as the reference: comparable across threads)

    def tic(self):
        self.start_time = self._now()

    def toc(self, average=True):
        self.diff = self._now() - self.start_time
        self.calls += 1
        self.total_time += self.diff
        return self.average_time if average else self.diff
