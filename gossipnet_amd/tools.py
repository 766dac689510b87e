"""`nms_net.tools.Timer` (reference nms_net/tools.py:11-35; test.py:56,68,71 times `sess.run` with it): wall-clock
tic/toc with a running average.  The kernels behind Gnet.run are asynchronous; pass `sync=` a callable (e.g.
`torch.cuda.synchronize`) to time completed work rather than launches."""
import time


class Timer(object):
    def __init__(self, sync=None):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.
        self._sync = sync

    def tic(self):
        if self._sync is not None:
            self._sync()
        self.start_time = time.time()

    def toc(self, average=True):
        if self._sync is not None:
            self._sync()
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
