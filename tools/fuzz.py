"""Fuzz: nasty but legal batches through training and inference steps -- empty images, single detections, sizes around the
32 / 64 / 256 tile edges, no ground truth, all-crowd ground truth, duplicated and zero-area boxes, tied and extreme scores,
one class only, and hostile ones (NaN / infinite coordinates and scores, classes outside 1..C) -- looking for GPU faults, hangs and non-finite results from finite, non-degenerate inputs.   python tools/fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
experiment_cfg()
cfg.gnet.num_blocks = 3
NC = 80
net = Gnet(NC, device=dev)
EDGE_SIZES = [0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1023, 1025]


def nasty_image():
    n = int(rng.choice(EDGE_SIZES)) if rng.uniform() < 0.6 else int(rng.integers(0, 1500))
    im = make_image(max(n, 1), NC, seed=int(rng.integers(1 << 30)), preset=("dense", "coco_like")[int(rng.integers(2))])
    for k in ("dets", "det_scores", "det_classes"):
        im[k] = im[k][:n].copy()
    mode = int(rng.integers(0, 10))
    n = im["dets"].shape[0]
    if mode == 0 and n > 1:                      # duplicates (ties everywhere)
        idx = rng.integers(0, max(n // 4, 1), n)
        im["dets"], im["det_classes"] = im["dets"][idx].copy(), im["det_classes"][idx].copy()
        if rng.uniform() < 0.5:
            im["det_scores"] = im["det_scores"][idx].copy()
    elif mode == 1 and n > 0:                    # zero-area and inverted boxes
        k = rng.integers(0, n, max(n // 5, 1))
        im["dets"][k, 2] = im["dets"][k, 0]
        k = rng.integers(0, n, max(n // 7, 1))
        im["dets"][k, 3] = im["dets"][k, 1] - 1.0
    elif mode == 2:                              # no ground truth
        for k in ("gt_boxes", "gt_crowd", "gt_classes"):
            im[k] = im[k][:0].copy()
    elif mode == 3:                              # all crowd
        im["gt_crowd"] = np.ones_like(im["gt_crowd"])
    elif mode == 4 and n > 0:                    # tied / extreme scores
        im["det_scores"] = rng.choice(np.array([0.0, 1.0, 0.5, 1e-30, 0.999999], np.float32), n).astype(np.float32)
    elif mode == 5 and n > 0:                    # one class, everything overlaps everything
        im["det_classes"][:] = 1; im["gt_classes"][:] = 1
        im["dets"] = (im["dets"][:1] + rng.normal(0, 0.5, (n, 4))).astype(np.float32)
    elif mode == 6 and n > 0:                    # far apart: no edges at all
        im["dets"] = (np.arange(n, dtype=np.float32)[:, None] * 50.0 + np.array([0, 0, 10, 10], np.float32)[None]).astype(np.float32)
    elif mode == 7:                              # many ground-truth boxes
        g = int(rng.integers(200, 1200))
        b = rng.uniform(0, 500, (g, 2)).astype(np.float32)
        im["gt_boxes"] = np.concatenate([b, b + rng.uniform(5, 120, (g, 2)).astype(np.float32)], 1)
        im["gt_crowd"] = rng.uniform(size=g) < 0.1
        im["gt_classes"] = rng.integers(1, NC + 1, g).astype(np.int32)
    elif mode == 8 and n > 0:                    # hostile: NaN / infinite coordinates and scores, classes outside 1..C
        k = rng.integers(0, n, max(n // 6, 1))
        im["dets"][k, int(rng.integers(4))] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e30, -1e30], np.float32))
        k = rng.integers(0, n, max(n // 6, 1))
        im["det_scores"][k] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), len(k))
        k = rng.integers(0, n, max(n // 6, 1))
        im["det_classes"][k] = rng.choice(np.array([0, -1, NC + 1, 2 ** 31 - 1, -2 ** 31], np.int64), len(k)).astype(np.int32)
        if im["gt_classes"].shape[0]:
            im["gt_classes"][0] = int(rng.choice(np.array([0, -7, NC + 5])))
            im["gt_boxes"][0, int(rng.integers(4))] = np.nan
        im["hostile"] = True
    return im


t0 = time.time()
for case in range(cases):
    imgs = [nasty_image() for _ in range(int(rng.integers(1, 6)))]
    desc = [(int(im["dets"].shape[0]), int(im["gt_boxes"].shape[0])) for im in imgs]
    hostile = any(im.pop("hostile", False) for im in imgs)
    try:
        b = DeviceBatch(imgs, dev)
        net.run(b)
        torch.cuda.synchronize()
        loss = net.loss.cpu().numpy() if net.num_dets > 0 else np.zeros(1)
        degenerate = any((im["dets"][:, 2] <= im["dets"][:, 0]).any() or (im["dets"][:, 3] <= im["dets"][:, 1]).any() for im in imgs if im["dets"].shape[0])
        if not degenerate and not hostile:
            assert np.isfinite(loss).all(), "non-finite loss"
            assert bool(torch.isfinite(net.grads).all().item()), "non-finite gradient"
        net.run(b, training=False)
        torch.cuda.synchronize()
    except Exception as e:
        print("case %d %s: %s: %s" % (case, desc, type(e).__name__, e), flush=True)
        raise
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", case, desc, "E", int(net.num_edges), flush=True)
print("fuzz: %d cases in %.1f s, no fault" % (cases, time.time() - t0))
