"""Training-step update on the GPU against the numpy restatement of TF's rules."""
import numpy as np
import pytest
import torch

from oracle import optim_oracle as oo
from tests.util import make_pair, make_image

pytestmark = pytest.mark.gpu


def test_adam_momentum_clip_against_oracle():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer
    net, _ = make_pair(1, 1)
    offs = np.concatenate([[0], np.cumsum([int(np.prod(s)) for _, s in net._spec])])
    rng = np.random.default_rng(0)
    for kind, clip in (("adam", -1.0), ("adam", 0.05), ("sgd", 0.5)):
        cfg.train.optimizer = kind
        cfg.train.gradient_clipping = clip
        opt = Optimizer(net)
        p = net.params.cpu().numpy().astype(np.float64)
        m = np.zeros_like(p); v = np.zeros_like(p)
        for t in range(1, 4):
            g = rng.normal(size=p.shape).astype(np.float32) * 0.01
            net.grads.copy_(torch.from_numpy(g).to(net.device))
            opt.apply_gradients(1e-3)
            gg = oo.clip_by_norm(g, offs, clip) if clip > 0 else g.astype(np.float64)
            if kind == "adam":
                p, m, v = oo.adam_step(p, gg, m, v, 1e-3, t)
            else:
                p, m = oo.momentum_step(p, gg, m, 1e-3, cfg.train.momentum)
            got = net.params.cpu().numpy()
            assert np.abs(got - p).max() < 2e-6, (kind, clip, t)
    cfg.train.optimizer = "adam"; cfg.train.gradient_clipping = -1.0


def test_train_steps_reduce_the_loss():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer, LearningRate, ExponentialMovingAverage, train_step
    net, _ = make_pair(80, 2)
    net.weight_reg = cfg.train.weight_decay
    cfg.train.lr_multi_step = [(3, 1e-3), (100, 1e-4)]
    opt, lr, ema = Optimizer(net), LearningRate(), ExponentialMovingAverage(0.7)
    batch = make_image(150, 80, seed=0)
    losses = []
    for it in range(1, 9):
        losses.append(float(train_step(net, opt, batch, lr.get_lr(it))))
        ema.apply(loss=losses[-1])
    assert losses[-1] < losses[0]
    assert lr.get_lr(50) == 1e-4 and opt.global_step == 8
    assert min(losses) <= ema.average("loss") <= max(losses)


def test_learning_rate_schedule_cpu_semantics():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import LearningRate
    cfg.train.lr_multi_step = [(2, 0.1), (4, 0.01)]
    lr = LearningRate()
    assert [lr.get_lr(i) for i in range(1, 7)] == [0.1, 0.1, 0.01, 0.01, 0.01, 0.01]   # train.py:31-37
