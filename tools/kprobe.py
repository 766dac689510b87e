"""Per-class kernel milliseconds of one step (HIP events on every launch), training and inference, for quick A/B runs:
python tools/kprobe.py [images] [preset]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
dev = torch.device("cuda", 0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
preset = sys.argv[2] if len(sys.argv) > 2 else "dense"
experiment_cfg()
net = Gnet(80, device=dev)
imgs = [make_image(2000, 80, seed=i, preset=preset) for i in range(images)]
for mode, batch in (("train", DeviceBatch(imgs, dev)),
                    ("infer", DeviceBatch([{k: im[k] for k in ("dets", "det_scores", "det_classes")} for im in imgs], dev))):
    for _ in range(3):
        net.run(batch)
    net.enable_kernel_timing(classes=None, capacity=4096)
    reps = 5
    for _ in range(reps):
        net.run(batch)
    torch.cuda.synchronize()
    t = net.read_kernel_timing()
    print(mode, "E", int(net.num_edges), " ".join("%s=%.4f" % (k, v[0] / reps) for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])),
          "sum=%.3f" % (sum(v[0] for v in t.values()) / reps))
