"""Soak: a training loop over images of changing size (N, E, GT counts and presets differ from step to step, so the
workspace is re-planned and the side-stream hand-offs see every ordering), checked for finite losses, a loss that
falls on a repeated image, no growth of device memory, and -- every 50 steps -- bitwise reproducibility of a
step replayed from a parameter snapshot.   python tools/soak.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
from gossipnet_amd.train import Optimizer, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
experiment_cfg()
cfg.train.optimizer = "adam"
net = Gnet(80, device=dev)
opt = Optimizer(net)
rng = np.random.default_rng(0)
probe = DeviceBatch([make_image(600, 80, seed=12345)], dev)
net.run(probe); first = float(net.loss.sum().item())
peak0 = None
t0 = time.time()
for it in range(steps):
    k = int(rng.integers(1, 5))
    imgs = [make_image(int(rng.integers(40, 3000)), 80, seed=int(rng.integers(1 << 30)), preset=("dense", "coco_like")[int(rng.integers(2))]) for _ in range(k)]
    b = DeviceBatch(imgs, dev)
    if it % 50 == 49:                       # replay: same parameters, same batch -> the same gradient, bit for bit
        net.run(b); g1 = net.grads.clone()
        net.run(DeviceBatch([make_image(int(rng.integers(40, 800)), 80, seed=7)], dev))      # something else in between
        net.run(b); assert torch.equal(g1, net.grads), "step %d: replayed gradient differs" % it
    if os.environ.get("SOAK_VERBOSE"):
        print("step", it, [int(im["dets"].shape[0]) for im in imgs], [int(im["gt_boxes"].shape[0]) for im in imgs], flush=True)
    loss = train_step(net, opt, b, 1e-4)
    if os.environ.get("SOAK_VERBOSE"):
        torch.cuda.synchronize(); print("   E", int(net.num_edges), flush=True)
    l = float(loss.sum().item())
    assert np.isfinite(l), "step %d: loss %r" % (it, l)
    if it == 20:
        torch.cuda.synchronize(); peak0 = torch.cuda.max_memory_allocated(dev)
net.run(probe); last = float(net.loss.sum().item())
torch.cuda.synchronize()
peak1 = torch.cuda.max_memory_allocated(dev)
print("soak: %d steps in %.1f s; probe loss %.4f -> %.4f; peak memory %.2f GB at step 20, %.2f GB at the end" % (steps, time.time() - t0, first, last, peak0 / 2**30, peak1 / 2**30))
assert last < first, "the probe image's loss did not fall"
print("soak ok")
