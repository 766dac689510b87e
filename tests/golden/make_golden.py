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

CASES = [  # name, N, C, B, seed
    ("n6_c1_b1", 6, 1, 1, 0),
    ("n40_c1_b2", 40, 1, 2, 1),
    ("n48_c80_b2", 48, 80, 2, 2),
]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, n, c, b, seed in CASES:
        batch = make_image(n, c, seed=seed)
        params = go.init_params(c, b)
        orc = go.GnetOracle(c, b, params=params)
        st = {}
        out, grads = orc.forward_backward(batch, stats=st)
        flat_g = go.flatten(grads, c, b)
        np.savez_compressed(
            os.path.join(here, name + ".npz"),
            num_classes=c, num_blocks=b, seed=seed,
            **{"in_" + k: v for k, v in batch.items()},
            params=go.flatten(params, c, b),
            neighbor_pair_idxs=out["neighbor_pair_idxs"].astype(np.int32),
            det_anno_iou=out["det_anno_iou"], raw_pw_feats=out["raw_pw_feats"],
            pw_feats=out["pw_feats"].detach().numpy(),
            block_feats=np.stack([x.detach().numpy() for x in out["block_feats"]]),
            prediction=out["prediction"].detach().numpy(),
            labels=out["labels"], weights=out["weights"].numpy(), det_gt_matching=out["det_gt_matching"],
            loss=np.float32(out["loss"].item()), loss_normed=np.float32(out["loss_normed"].item()),
            grads=flat_g, relu_margin=st["relu_margin"], max_gap=st["max_gap"])
        print(name, "E", len(out["neighbor_pair_idxs"]), "loss", float(out["loss"]), "margins", st)


if __name__ == "__main__":
    main()
