"""Training step around the hot path -- the reference's train.py:26-37 (LearningRate schedule),
train.py:64-77 (get_optimizer: Adam / Momentum through slim.learning.create_train_op with optional
per-gradient norm clipping) and train.py:263-272 (EMA(0.7) loss smoothing).  SURVEY.md 8f rank 1.
The update itself runs in libgossipnet_hip.so (csrc/optim.hip) on the flat parameter buffer.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import cfg


class LearningRate(object):
    """Multi-step schedule over cfg.train.lr_multi_step = [(last_iteration, lr), ...] -- the table of the reference's
    LearningRate (train.py:26-37): entry k's rate applies up to AND INCLUDING its iteration, the last rate for ever after.

    Stateless: the rate is looked up from the iteration itself (bisection over the boundaries).  Fed the consecutive
    iterations of a training run it returns what the reference returns; the reference keeps a cursor that only moves
    when `iter ==` a boundary, so a run resumed past a boundary (train.py:274-283 restores global_step) would stay on
    the old rate there -- here the resumed iteration gets the rate of its own interval."""

    def __init__(self, table=None):
        table = cfg.train.lr_multi_step if table is None else table
        self.boundaries = [int(it) for it, _ in table]
        self.rates = [float(lr) for _, lr in table]

    def get_lr(self, iteration):
        import bisect
        k = bisect.bisect_left(self.boundaries, int(iteration))     # boundaries strictly below this iteration
        return self.rates[min(k, len(self.rates) - 1)]


class ExponentialMovingAverage(object):
    """tf.train.ExponentialMovingAverage(decay) without num_updates, applied to loss TENSORS (train.py:263-272).
    For a Tensor (not a Variable) TF of that era creates a ZEROS slot and does not debias, so the smoothed
    losses start at (1 - decay) * value and ramp up; shadow -= (1 - decay) * (shadow - value).  (Stated from the
    TF <= 1.0 source as remembered -- TensorFlow is not installable here to confirm; only the displayed loss
    depends on it.)"""

    def __init__(self, decay=0.7):
        self.decay = decay
        self.shadow = {}

    def apply(self, **values):
        for k, v in values.items():
            v = float(v)
            old = self.shadow.get(k, 0.0)
            self.shadow[k] = old - (1.0 - self.decay) * (old - v)
        return dict(self.shadow)

    def average(self, name):
        return self.shadow[name]


class Optimizer(object):
    """get_optimizer (train.py:64-77) for a Gnet: cfg.train.optimizer in {'adam', 'sgd'}; sgd = Momentum
    with cfg.train.momentum; cfg.train.gradient_clipping > 0 -> clip_by_norm of every gradient tensor."""

    def __init__(self, net):
        self.net = net
        self.lib = _lib.load()
        self.kind = cfg.train.optimizer
        if self.kind not in ("adam", "sgd"):
            raise ValueError('unknown optimizer {}'.format(self.kind))
        n = net.params.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=net.device)
        self.v = torch.zeros(n, dtype=torch.float32, device=net.device) if self.kind == "adam" else None
        self.global_step = 0
        self.clip = float(cfg.train.gradient_clipping)
        self._offs = torch.from_numpy(np.asarray(net.tensor_offsets(), np.int64)).to(net.device)
        self._ntensors = len(net._spec)

    def apply_gradients(self, lr, grad_scale=1.0):
        net, s = self.net, C.c_void_p(torch.cuda.current_stream(self.net.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        n = net.params.numel()
        if self.clip > 0:
            if grad_scale != 1.0:          # clip_gradient_norm acts on the gradient the optimizer sees: scale first
                net.grads.mul_(float(grad_scale))
                grad_scale = 1.0
            _lib.check(self.lib.gnet_clip_by_norm(p(net.grads), p(self._offs), self._ntensors, self.clip, s),
                       "gnet_clip_by_norm")
        self.global_step += 1
        if self.kind == "adam":
            _lib.check(self.lib.gnet_adam_step(p(net.params), p(net.grads), p(self.m), p(self.v), n, float(lr), 0.9,
                                               0.999, 1e-8, self.global_step, float(grad_scale), s), "gnet_adam_step")
        else:
            _lib.check(self.lib.gnet_momentum_step(p(net.params), p(net.grads), p(self.m), n, float(lr),
                                                   float(cfg.train.momentum), float(grad_scale), s), "gnet_momentum_step")


def train_step(net, opt, batch, lr, dist=None):
    """One iteration of train.py:316-320: forward + loss (+ l2) + backward [+ all-reduce] + update.
    Under data parallelism (dist) the caller sets net.grad_scale = 1 / (images of the global step); the l2
    regulariser is added once per step: every rank contributes 1 / world of it to the SUM all-reduce, which is the
    one flat-buffer collective of data_parallel.GradientExchange (side stream; bench.py --gpus N uses the same object)."""
    if dist is not None:
        net.reg_scale = 1.0 / dist.get_world_size()
    net.run(batch)
    if dist is not None:
        ex = getattr(net, "_grad_exchange", None)
        if ex is None or ex.dist is not dist:
            from .data_parallel import GradientExchange
            ex = net._grad_exchange = GradientExchange(dist, net.device)
        ex.launch(net.grads)
    with torch.cuda.device(net.device):
        if dist is not None:
            ex.wait()                    # where the summed gradient is consumed
        opt.apply_gradients(lr)
    return net.loss
