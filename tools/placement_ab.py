"""A/B of the side-stream placements inside ONE process (interleaved rounds cancel clock drift): the reverse-edge transposition and the
zeroing half of the backward preparation beside pw_fwd (shipped until round 6) or behind the forward pass.  python tools/placement_ab.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
experiment_cfg()
net = Gnet(80, device=dev)
for images in (8, 1):
    batch = DeviceBatch([make_image(2000, 80, seed=i) for i in range(images)], dev)
    res = {}
    for rnd in range(4):
        for t_after in (False, True):
            for z_after in (False, True):
                net.transpose_after_forward = t_after
                net.zero_after_forward = z_after
                for _ in range(4):
                    net.run(batch)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    net.run(batch)
                torch.cuda.synchronize()
                res.setdefault((t_after, z_after), []).append((time.perf_counter() - t0) / steps * 1e3)
    for k, v in sorted(res.items()):
        print("images %d  transpose_after_forward=%-5s zero_after_forward=%-5s  ms/step: %s  (min %.4f)" % (images, k[0], k[1], " ".join("%.4f" % x for x in v), min(v)))
