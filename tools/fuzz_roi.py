"""Fuzz of RoiPool / RoiPoolGrad against the C oracle: channel counts on both sides of every kernel selection (C = 1024 rows
kernel, multiples of 256 -> block backward, multiples of 4, odd), ROI counts around the backward's 64-ROI steps and 256-ROI
rounds, bins up to 16 x 16 and non-square, maps from 1 x 1, scales, ROIs larger than / outside / smaller than a pixel of the map,
several images.  top / argmax / ordered backward bit-exact, atomic backward <= 1e-5.   python tools/fuzz_roi.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gossipnet_amd.roi_pooling_layer.roi_pooling_op import roi_pool_raw, roi_pool_grad
from oracle import native

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
d = "cuda:0"
t0 = time.time()
for case in range(cases):
    C = int(rng.choice([1, 3, 4, 7, 16, 64, 100, 256, 260, 512, 768, 1024, 1280, 2048]))
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    if C >= 512:
        H, W = min(H, 14), min(W, 14)
    R = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 513])) if rng.uniform() < 0.6 else int(rng.integers(0, 400))
    if C >= 768:
        R = min(R, 257)
    ph, pw = (int(rng.integers(1, 17)), int(rng.integers(1, 17))) if rng.uniform() < 0.5 else (7, 7)
    if ph * pw * R * C > 6e7:
        ph, pw = 3, 2
    scale = float(rng.choice([1.0, 0.5, 0.25, 1 / 16.0, 1 / 3.0]))
    data = rng.normal(size=(B, H, W, C)).astype(np.float32)
    if rng.uniform() < 0.3:
        data = np.round(data)                                  # exact ties between pixels: the FIRST maximum must win
    kind = int(rng.integers(4))
    ext = max(H, W) / scale
    if kind == 0:                                              # ordinary boxes, partly outside the map
        xy = rng.uniform(-0.3 * ext, 1.1 * ext, (R, 2)); wh = rng.uniform(1, 0.7 * ext + 1, (R, 2))
    elif kind == 1:                                            # smaller than a map pixel
        xy = rng.uniform(0, ext, (R, 2)); wh = rng.uniform(0, 1.0 / scale, (R, 2))
    elif kind == 2:                                            # the whole map and beyond
        xy = rng.uniform(-2 * ext, 0, (R, 2)); wh = rng.uniform(ext, 4 * ext, (R, 2))
    else:                                                      # integer coordinates (ends on pixel boundaries, .5 roundings)
        xy = np.round(rng.uniform(0, ext, (R, 2)) * 2) / 2; wh = np.round(rng.uniform(0, ext, (R, 2)) * 2) / 2
    rois = np.concatenate([rng.integers(0, B, (R, 1)), xy, xy + wh], 1).astype(np.float32)
    desc = dict(B=B, H=H, W=W, C=C, R=R, ph=ph, pw=pw, scale=scale, kind=kind)
    try:
        rtop, ram = native.roi_pool(data, rois, ph, pw, scale)
        top, am = roi_pool_raw(torch.tensor(data, device=d), torch.tensor(rois, device=d).reshape(R, 5), ph, pw, scale)
        torch.cuda.synchronize()
        assert np.array_equal(am.cpu().numpy(), ram), "argmax"
        assert np.array_equal(top.cpu().numpy(), rtop), "top"
        g = rng.normal(size=rtop.shape).astype(np.float32)
        rgrad = native.roi_pool_grad((B, H, W, C), rois, ram, g, ph, pw, scale)
        got = roi_pool_grad(torch.tensor(data, device=d), torch.tensor(rois, device=d).reshape(R, 5), am, torch.tensor(g, device=d), ph, pw, scale, True)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), rgrad), "ordered backward"
    except Exception as e:
        print("case %d %s: %s: %s" % (case, desc, type(e).__name__, e), flush=True)
        raise
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", case, desc, flush=True)
print("roi fuzz: %d cases in %.1f s, all bit-exact" % (cases, time.time() - t0))
