"""CPU ORACLE for the training-step update -- TEST INFRASTRUCTURE (see oracle/gnet_oracle.py header).
PARITY UNPINNED: restates the published update rules of TensorFlow's AdamOptimizer / MomentumOptimizer /
clip_by_norm (un-vendored TF; call sites train.py:64-77)."""
import numpy as np


def adam_step(p, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    p, g, m, v = (np.asarray(x, np.float64) for x in (p, g, m, v))
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v


def momentum_step(p, g, acc, lr, momentum):
    acc = momentum * np.asarray(acc, np.float64) + np.asarray(g, np.float64)
    return np.asarray(p, np.float64) - lr * acc, acc


def clip_by_norm(g, offsets, clip):
    g = np.array(g, np.float64)
    for b, e in zip(offsets[:-1], offsets[1:]):
        n = np.sqrt((g[b:e] ** 2).sum())
        if n > clip:
            g[b:e] *= clip / n
    return g
