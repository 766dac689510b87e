"""Generates the golden vectors of tests/golden/*.npz with the CPU oracle (oracle/gnet_oracle.py).

The reference itself cannot be run or built here (TensorFlow 0.12 / TF headers absent), so
these vectors are produced by the oracle -- the restatement pinned by the hand-derived KATs of
tests/test_oracle_kat.py -- and serve as regression fixtures for both the oracle (CPU tests) and the HIP
path (GPU tests).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gnet_oracle as go            # noqa: E402
from gossipnet_amd.synthetic import make_image  # noqa: E402

CASES = [  # name, N, C, B, seed, generator preset (M = N // dets_per_obj: dense 25, coco_like 8)
    ("n6_c1_b1", 6, 1, 1, 0, "dense"),
    ("n40_c1_b2", 40, 1, 2, 1, "dense"),
    ("n48_c80_b2", 48, 80, 2, 2, "dense"),
    # SURVEY 8c shapes (N, M, C, B) = (64, 8, 80, 2) and (300, 12, 80, 16), seeds 0..2
    ("n64_m8_c80_b2_s0", 64, 80, 2, 0, "coco_like"),
    ("n64_m8_c80_b2_s1", 64, 80, 2, 1, "coco_like"),
    ("n64_m8_c80_b2_s2", 64, 80, 2, 2, "coco_like"),
    ("n300_m12_c80_b16_s0", 300, 80, 16, 0, "dense"),
    ("n300_m12_c80_b16_s1", 300, 80, 16, 1, "dense"),
    ("n300_m12_c80_b16_s2", 300, 80, 16, 2, "dense"),
]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])
    for name, n, c, b, seed, preset in CASES:
        if only and name not in only:
            continue
        batch = make_image(n, c, seed=seed, preset=preset)
        params = go.init_params(c, b)
        orc = go.GnetOracle(c, b, params=params)
        st = {}
        out, grads = orc.forward_backward(batch, stats=st)
        flat_g = go.flatten(grads, c, b)
        np.savez_compressed(
            os.path.join(here, name + ".npz"),
            num_classes=c, num_blocks=b, seed=seed,
            **{"in_" + k: v for k, v in batch.items()},
            # the 16-block parameter vector (2.3 MB of incompressible floats) is regenerated from its seed by
            # go.init_params (torch CPU generator) and verified through its float64 checksums
            **({"params": go.flatten(params, c, b)} if b < 16 else
               {"params_seed": 42, "params_sum": np.float64(go.flatten(params, c, b).astype(np.float64).sum()),
                "params_sumsq": np.float64((go.flatten(params, c, b).astype(np.float64) ** 2).sum())}),
            neighbor_pair_idxs=out["neighbor_pair_idxs"].astype(np.int32),
            det_anno_iou=out["det_anno_iou"], raw_pw_feats=out["raw_pw_feats"], preset=preset,
            pw_feats=out["pw_feats"].detach().numpy(),
            block_feats=np.stack([x.detach().numpy() for x in out["block_feats"]]),
            prediction=out["prediction"].detach().numpy(),
            labels=out["labels"], weights=out["weights"].numpy(), det_gt_matching=out["det_gt_matching"],
            loss=np.float32(out["loss"].item()), loss_normed=np.float32(out["loss_normed"].item()),
            grads=flat_g, relu_margin=st["relu_margin"], max_gap=st["max_gap"])
        print(name, "E", len(out["neighbor_pair_idxs"]), "M", len(batch["gt_crowd"]), "loss", float(out["loss"]), "margins", st)


if __name__ == "__main__":
    main()
