"""Data-parallel glue: images shard across ranks (one process per GPU), parameters are replicated,
and the only exchange per step is ONE all-reduce (sum) of the flat fp32 gradient buffer
(581 793 elements = 2.33 MB for C=80, B=16) -- RCCL over xGMI on the GPU box (backend "nccl"),
gloo in the CPU tests.  The reference has no distributed code; the step
semantics (gradient = mean over the images of the global batch) are set through Gnet.grad_scale.
"""
import torch


def shard_images(images, rank, world, costs=None, per_rank=None):
    """Deal the images of a global step to ranks.  With per-image costs (edge counts), use the
    longest-processing-time rule so that ranks get balanced edge totals (cost is ~ E, not N);
    per_rank caps the number of images a rank may take (equal image counts: weak scaling)."""
    if costs is None:
        return images[rank::world]
    order = sorted(range(len(images)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    taken = [0] * world
    mine = []
    for i in order:
        open_ranks = [k for k in range(world) if per_rank is None or taken[k] < per_rank] or list(range(world))
        r = min(open_ranks, key=lambda k: (loads[k], k))
        loads[r] += costs[i]
        taken[r] += 1
        if r == rank:
            mine.append(i)
    return [images[i] for i in sorted(mine)]


class GradientExchange(object):
    """The one collective of a data-parallel step, issued on a side stream behind the backward pass.

    launch(flat_grads) queues the all-reduce behind everything the compute stream has issued so far (reduce_partials wrote the
    buffer) and records `done` behind it; NOTHING is made to wait there.  wait() makes the current stream wait for `done`: call
    it where the summed gradient is consumed -- train_step() does, in front of the optimizer update; bench.py hands `done` to
    Gnet.defer_backward_until(), so that step i + 1's graph build, forward pass and loss (which do not touch the gradient
    buffer) run beside step i's all-reduce and only its backward pass (which overwrites the buffer) queues behind it.
    `timed=True` brackets every all-reduce with events on the side stream (they see the collective itself, not the step);
    read_us() returns (mean microseconds per all-reduce, count) and resets; at most `keep` samples are held.
    Used by train_step(..., dist=...) and by bench.py --gpus N: the same code path."""

    def __init__(self, dist, device, group=None, timed=False, keep=1024):
        self.dist, self.group, self.device = dist, group, torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self.timed, self._events, self._keep = timed, [], int(keep)
        self.done = None

    def launch(self, flat_grads):
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)                       # behind reduce_partials
        with torch.cuda.stream(self.side):
            if self.timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
            self.dist.all_reduce(flat_grads, op=self.dist.ReduceOp.SUM, group=self.group)
            if self.timed:
                e1.record(self.side)
                self._events.append((e0, e1))
                if len(self._events) > self._keep:
                    del self._events[:len(self._events) - self._keep]
            self.done = torch.cuda.Event()
            self.done.record(self.side)
        return self.done

    def wait(self):
        """The current stream waits for the last launched all-reduce (no-op when none is pending)."""
        if self.done is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done)
            self.done = None

    def __call__(self, flat_grads):
        """launch + wait: the summed gradient is usable by whatever the current stream queues next."""
        self.launch(flat_grads)
        self.wait()
        return flat_grads

    def read_us(self):
        if not self._events:
            return None, 0
        self._events[-1][1].synchronize()
        us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in self._events]
        self._events = []
        return sum(us) / len(us), len(us)


def allreduce_gradients(flat_grads, dist, group=None):
    """One collective per step on the flat gradient buffer (in place, sum)."""
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def broadcast_parameters(flat_params, dist, src=0, group=None):
    dist.broadcast(flat_params, src=src, group=group)
    return flat_params
