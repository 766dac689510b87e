"""Synthetic detection images for the Gnet hot path (SURVEY.md §8d generator).

The container has no COCO detections (reference data/README.md points to an
external tarball), so every BASELINE.json config is restated as synthetic
input of the same shape.  numpy only; deterministic per seed.

Fields follow Gnet.get_batch_spec (reference nms_net/network.py:131-146):
  dets f32 [N,4] xyxy pixels, det_scores f32 [N], det_classes i32 [N] (1-based),
  gt_boxes f32 [M,4], gt_crowd bool [M], gt_classes i32 [M].
"""
import numpy as np

PRESETS = {
    # name: (dets_per_obj, s_max)
    "dense": (25, 320.0),      # SURVEY §8d default: N=2000 -> E/N ~ 79
    "coco_like": (8, 128.0),   # N=2000 -> E/N ~ 32
}


def make_image(n_dets, num_classes, seed=0, preset="dense", canvas=(640.0, 480.0)):
    dets_per_obj, s_max = PRESETS[preset]
    rng = np.random.default_rng(seed)
    W, H = canvas
    n_obj = max(1, n_dets // dets_per_obj)
    ow = np.exp(rng.uniform(np.log(24.0), np.log(s_max), n_obj))
    oh = np.exp(rng.uniform(np.log(24.0), np.log(s_max), n_obj))
    ow = np.minimum(ow, W - 1.0)
    oh = np.minimum(oh, H - 1.0)
    ox = rng.uniform(0.0, W - ow)
    oy = rng.uniform(0.0, H - oh)
    ocls = rng.integers(1, num_classes + 1, n_obj)

    pick = rng.integers(0, n_obj, n_dets)
    w = ow[pick] * np.exp(rng.normal(0.0, 0.2, n_dets))
    h = oh[pick] * np.exp(rng.normal(0.0, 0.2, n_dets))
    cx = ox[pick] + 0.5 * ow[pick] + rng.normal(0.0, 0.15, n_dets) * ow[pick]
    cy = oy[pick] + 0.5 * oh[pick] + rng.normal(0.0, 0.15, n_dets) * oh[pick]
    x1 = np.clip(cx - 0.5 * w, 0.0, W - 5.0)
    y1 = np.clip(cy - 0.5 * h, 0.0, H - 5.0)
    x2 = np.clip(cx + 0.5 * w, x1 + 4.0, W)
    y2 = np.clip(cy + 0.5 * h, y1 + 4.0, H)
    dets = np.stack([x1, y1, x2, y2], 1).astype(np.float32)

    # distinct scores in (0,1): a random permutation of an evenly spaced grid plus jitter
    scores = (rng.permutation(n_dets) + rng.uniform(0.25, 0.75, n_dets)) / n_dets
    scores = scores.astype(np.float32)
    same = rng.uniform(size=n_dets) < 0.8
    det_classes = np.where(same, ocls[pick], rng.integers(1, num_classes + 1, n_dets)).astype(np.int32)

    gt_boxes = np.stack([ox, oy, ox + ow, oy + oh], 1).astype(np.float32)
    gt_crowd = rng.uniform(size=n_obj) < 0.05
    gt_classes = ocls.astype(np.int32)
    return {
        "dets": dets, "det_scores": scores, "det_classes": det_classes,
        "gt_boxes": gt_boxes, "gt_crowd": gt_crowd, "gt_classes": gt_classes,
    }
