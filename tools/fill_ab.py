"""A/B of the graph fill's placement inside ONE process (interleaved rounds): on the main stream at the head of the step (shipped until
round 6) or on the side stream behind the count.  python tools/fill_ab.py [steps]"""
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
    grads = {}
    for rnd in range(4):
        for side in (False, True):
            net.fill_on_side = side
            for _ in range(4):
                net.run(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net.run(batch)
            torch.cuda.synchronize()
            res.setdefault(side, []).append((time.perf_counter() - t0) / steps * 1e3)
            grads[side] = net.grads.clone()
    for k, v in sorted(res.items()):
        print("images %d  fill_on_side=%-5s  ms/step: %s  (min %.4f)" % (images, k, " ".join("%.4f" % x for x in v), min(v)))
    print("   gradients bitwise equal:", bool(torch.equal(grads[False], grads[True])))
