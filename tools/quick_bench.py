"""Quick timing of the two step shapes on the bench's own images (seeds 0..), no kernel table, no CPU baseline:
python tools/quick_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
experiment_cfg()
net = Gnet(80, device=dev)
for images in (1, 8):
    batch = DeviceBatch([make_image(2000, 80, seed=i) for i in range(images)], dev)
    for _ in range(5):
        net.run(batch)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = steps if images == 1 else max(10, steps // 2)
        for _ in range(n):
            net.run(batch)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print("images/step %d: %.1f det/s  %.4f ms/step  E/N %.1f" % (images, 2000 * images / best, best * 1e3, net.num_edges / (2000.0 * images)))
