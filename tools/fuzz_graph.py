"""Fuzz of the neighbour search (`_iou` >= neighbor_thresh, row-major pairs, several images block-diagonal) against the oracle's
dense matrix: sizes around the sweep's 32 / 64 / 256 tile edges, duplicated boxes, zero-area and inverted boxes (NaN / negative
IoU), boxes exactly AT the threshold, huge and tiny coordinates, thresholds 0, 0.2, 0.5, 1.  Pairs and their IoUs bit-exact.
python tools/fuzz_graph.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
from oracle import gnet_oracle as go

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
SIZES = [0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1025]
t0 = time.time()
nets = {}
for case in range(cases):
    thr = float(rng.choice([0.2, 0.2, 0.2, 0.0, 0.5, 1.0]))
    if thr not in nets:
        experiment_cfg(); cfg.gnet.num_blocks = 1; cfg.gnet.neighbor_thresh = thr
        nets[thr] = Gnet(80, device=dev)
    experiment_cfg(); cfg.gnet.num_blocks = 1; cfg.gnet.neighbor_thresh = thr
    net = nets[thr]
    imgs = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.choice(SIZES)) if rng.uniform() < 0.6 else int(rng.integers(0, 2500))
        im = make_image(max(n, 1), 80, seed=int(rng.integers(1 << 30)), preset=("dense", "coco_like")[int(rng.integers(2))])
        dets = im["dets"][:n].copy()
        mode = int(rng.integers(7))
        if mode == 0 and n > 1:
            dets = dets[rng.integers(0, max(n // 5, 1), n)]
        elif mode == 1 and n > 0:
            k = rng.integers(0, n, max(n // 4, 1)); dets[k, 2] = dets[k, 0]
            k = rng.integers(0, n, max(n // 6, 1)); dets[k, 3] = dets[k, 1] - 2.0
        elif mode == 2 and n > 1:                      # pairs exactly at the threshold: b = a stretched so that IoU = 1/5, 1/2
            dets = np.tile(np.array([[0, 0, 10, 10]], np.float32), (n, 1))
            dets[1::2] = np.array([0, 0, 10, 50], np.float32) if thr == 0.2 else np.array([0, 0, 10, 20], np.float32)
            dets += np.float32(16.0) * rng.integers(0, 3, (n, 1)).astype(np.float32)
        elif mode == 3 and n > 0:
            dets = dets * np.float32(rng.choice([1e-3, 1e4, 1e7]))
        elif mode == 4 and n > 0:
            dets = np.round(dets / 8) * 8              # coarse grid: many equal areas and intersections
        imgs.append({"dets": dets.astype(np.float32), "det_scores": im["det_scores"][:n], "det_classes": im["det_classes"][:n]})
    desc = [int(im["dets"].shape[0]) for im in imgs]
    try:
        net.run(DeviceBatch(imgs, dev), training=False)
        torch.cuda.synchronize()
        pairs = net.neighbor_pair_idxs.cpu().numpy().reshape(-1, 2)
        ious = net.edge_iou.cpu().numpy()
        off, cnt = 0, 0
        for im in imgs:
            n = im["dets"].shape[0]
            if n:
                db = go.xyxy_to_boxdata(im["dets"])
                with np.errstate(all="ignore"):
                    m = go.iou(db, db)
                ref = np.argwhere(m >= np.float32(thr))
                k = len(ref)
                assert np.array_equal(pairs[cnt:cnt + k], ref + off), "pairs of an image"
                assert np.array_equal(ious[cnt:cnt + k], m[ref[:, 0], ref[:, 1]], equal_nan=True), "IoUs of the pairs"
                cnt += k
            off += n
        assert cnt == len(pairs), "extra pairs"
    except Exception as e:
        print("case %d thr %g dets %s: %s: %s" % (case, thr, desc, type(e).__name__, e), flush=True)
        raise
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", case, thr, desc, "E", len(pairs), flush=True)
experiment_cfg()
print("graph fuzz: %d cases in %.1f s, all bit-exact" % (cases, time.time() - t0))
