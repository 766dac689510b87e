"""Fuzz of the optimizer kernels (Adam / Momentum, per-tensor clip_by_norm) against oracle/optim_oracle.py: 24 network layouts (with and
without the 16-byte-aligned reduce_imfeats tensors), tensors without gradient (zero norm under clipping), gradients of 1e-8 .. 1e3.
python tools/fuzz_optimizer.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet
from gossipnet_amd.train import Optimizer
from oracle import optim_oracle as oo, gnet_oracle as go
rng = np.random.default_rng(0)
bad = 0
for case in range(24):
    experiment_cfg()
    imf = bool(case % 2)
    cfg.gnet.num_blocks = int(rng.integers(1, 4))
    nc = int(rng.choice([1, 80]))
    if imf:
        cfg.gnet.imfeats = True; cfg.gnet.imfeat_dim = int(rng.choice([0, 64]))
    kind = ("adam", "sgd")[int(rng.integers(2))]
    clip = float(rng.choice([-1.0, 0.05, 5.0]))
    cfg.train.optimizer = kind; cfg.train.gradient_clipping = clip
    net = Gnet(nc, imfeat_channels=8, imfeat_stride=16) if imf else Gnet(nc)
    opt = Optimizer(net)
    offs = np.asarray(net.tensor_offsets(), np.int64)
    sizes = [int(np.prod(s)) for _, s in net._spec]
    p = net.params.cpu().numpy().astype(np.float64); m = np.zeros_like(p); v = np.zeros_like(p)
    for t in range(1, 4):
        g = np.zeros(p.shape, np.float32)
        for o, n in zip(offs, sizes):
            mode = int(rng.integers(4))
            if mode == 0: continue                                  # a tensor without gradient (zero norm under clipping)
            sc = [1e-2, 1e3, 1e-8][mode - 1]
            g[o:o + n] = rng.normal(size=n).astype(np.float32) * np.float32(sc)
        net.grads.copy_(torch.from_numpy(g).to(net.device))
        opt.apply_gradients(1e-3)
        gg = g.astype(np.float64)
        if clip > 0:
            gg = gg.copy()
            for o, n in zip(offs, sizes):
                nrm = np.sqrt((gg[o:o + n] ** 2).sum())
                gg[o:o + n] = gg[o:o + n] * clip / max(nrm, clip)
        if kind == "adam": p, m, v = oo.adam_step(p, gg, m, v, 1e-3, t)
        else: p, m = oo.momentum_step(p, gg, m, 1e-3, cfg.train.momentum)
        got = net.params.cpu().numpy()
        # padding elements (imfeat alignment) stay as they were
        err = np.abs(got - p).max() / max(1.0, np.abs(p).max())
        if not np.isfinite(got).all() or err > 3e-6:
            bad += 1; print("case", case, kind, clip, "imfeat", imf, "step", t, "err", err, "finite", np.isfinite(got).all())
print("optimizer fuzz: 24 configurations x 3 steps,", bad, "mismatches")
experiment_cfg()
