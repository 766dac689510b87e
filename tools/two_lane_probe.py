"""Probe: the 8-image step as two 4-image lanes on two streams (does the tail of one lane's kernels fill with the other's?).
python tools/two_lane_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gossipnet_amd.config import experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
dev = torch.device("cuda", 0)
experiment_cfg()
imgs = [make_image(2000, 80, seed=i) for i in range(8)]


def timeit(fn, n=10, reps=3):
    for _ in range(4):
        fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


net = Gnet(80, device=dev)
b8 = DeviceBatch(imgs, dev)
print("one lane, 8 images: %.3f ms" % timeit(lambda: net.run(b8)))
for split in ((4, 4), (2, 2, 2, 2)):
    nets = [net] + [Gnet(80, device=dev, reuse=True) for _ in split[1:]]
    bs, o = [], 0
    for k in split:
        bs.append(DeviceBatch(imgs[o:o + k], dev)); o += k
    streams = [torch.cuda.Stream(dev) for _ in split]

    def lanes():
        cur = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(cur)
        for n_, b_, st in zip(nets, bs, streams):
            with torch.cuda.stream(st):
                n_.begin(b_)
        for n_, st in zip(nets, streams):
            with torch.cuda.stream(st):
                n_.run()
        for st in streams:
            cur.wait_stream(st)
    print("%d lanes %s: %.3f ms" % (len(split), split, timeit(lanes)))
    print("   sequential on one stream: %.3f ms" % timeit(lambda: [n_.run(b_) for n_, b_ in zip(nets, bs)]))
