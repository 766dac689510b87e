"""Sanity at sizes past the configs: two dense 10 000-detection images in one step (E = 5.5 M; gradient = the sum of the single-image
gradients), a 40 000-detection image (E = 19.9 M: a training step refuses it with a clear error -- 32-bit offsets in the backward
pass -- inference runs), and the device repeating an earlier step bit for bit afterwards.   python tools/big_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
dev = torch.device("cuda", 0)
experiment_cfg()
net = Gnet(80, device=dev)
imgs = [make_image(10000, 80, seed=1), make_image(9000, 80, seed=2)]
gs, ls = [], []
for im in imgs:
    net.run(DeviceBatch([im], dev)); torch.cuda.synchronize()
    print("single: E", int(net.num_edges), "loss", float(net.loss), "finite", bool(torch.isfinite(net.grads).all().item()), flush=True)
    gs.append(net.grads.clone()); ls.append(float(net.loss))
t0 = time.time()
net.run(DeviceBatch(imgs, dev)); torch.cuda.synchronize()
print("pair: E", int(net.num_edges), "loss", float(net.loss), "vs", sum(ls), "time %.1f ms" % ((time.time() - t0) * 1e3), "peak GB %.1f" % (torch.cuda.max_memory_allocated(dev) / 2**30))
d = float((net.grads - gs[0] - gs[1]).abs().max() / (gs[0] + gs[1]).abs().max())
print("gradient of the pair vs the sum of the singles: rel err %.2e" % d)
im = make_image(40000, 80, seed=3, preset="coco_like")
try:
    net.run(DeviceBatch([im], dev)); torch.cuda.synchronize()
    print("N=40000 coco_like: E", int(net.num_edges), "loss", float(net.loss))
except Exception as e:
    print("N=40000:", type(e).__name__, e)
net.run(DeviceBatch([im], dev), training=False); torch.cuda.synchronize()
print("N=40000 inference: E", int(net.num_edges), "prediction finite", bool(torch.isfinite(net.prediction).all().item()), "peak GB %.1f" % (torch.cuda.max_memory_allocated(dev) / 2**30))
net.run(DeviceBatch(imgs[:1], dev)); torch.cuda.synchronize()
print("after: loss", float(net.loss), "== the first single", float(net.loss) == ls[0], "grads equal", bool(torch.equal(net.grads, gs[0])))
