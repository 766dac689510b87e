"""Fuzz of the standalone DetectionMatching op: random sizes around the wave / LDS edges, IoUs exactly at the 0.5 threshold, tied IoUs
and scores, values outside [0, 1] -- bit-exact against the C oracle -- and NaN IoUs / scores (no fault).   python tools/fuzz_matching.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gossipnet_amd.matching_module import detection_matching
from oracle import native
dev = "cuda:0"
rng = np.random.default_rng(0)
bad = 0
for case in range(2000):
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 2049])) if rng.uniform() < 0.5 else int(rng.integers(0, 3000))
    m = int(rng.choice([0, 1, 2, 63, 64, 65, 2047, 2048])) if rng.uniform() < 0.5 else int(rng.integers(0, 300))
    iou = rng.uniform(0, 1, (n, m)).astype(np.float32)
    mode = int(rng.integers(6))
    hostile = False
    if mode == 0 and n * m: iou[rng.uniform(size=iou.shape) < 0.3] = 0.5            # exactly at the threshold
    if mode == 1 and n * m: iou = np.round(iou * 4) / 4                              # many exact ties
    if mode == 2 and n * m: iou[rng.uniform(size=iou.shape) < 0.1] = np.nan; hostile = True
    if mode == 3 and n * m: iou *= 3.0; iou -= 1.0                                   # outside [0, 1]
    score = rng.uniform(0, 1, n).astype(np.float32)
    if mode == 4 and n: score = np.round(score * 3) / 3                              # tied scores
    if mode == 5 and n: score[rng.uniform(size=n) < 0.2] = np.nan; hostile = True
    ignore = rng.uniform(size=m) < 0.2
    out = detection_matching(torch.tensor(iou, device=dev), torch.tensor(score, device=dev), torch.tensor(ignore, device=dev))
    torch.cuda.synchronize()
    if not hostile:
        ref = native.det_matching(iou, score, ignore)
        for a, b in zip(out, ref):
            if not np.array_equal(a.cpu().numpy(), b):
                bad += 1; print("case", case, n, m, mode, "differs from the oracle"); break
print("matching fuzz: 2000 cases,", bad, "mismatches")
