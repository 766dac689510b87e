"""Inference + evaluation harness either side of the hot path.

* rescore(): the per-image loop of the reference's test.py:42-83 -- feed dets/scores/classes only, read
  `net.prediction` (raw logits, no sigmoid), undo the image scale, time the forward pass.
* save_dets(): the Fast-R-CNN-format pickle of test.py:86-111: (dets[class][image] -> [n,5], image_ids,
  cat_ids), protocol 2.
* val_run() / compute_aps(): the validation signal of train.py:133-206: detections with weight > 0,
  101-point interpolated AP over recall thresholds 0:.01:1 with searchsorted(side='left').
"""
import pickle
import time

import numpy as np


def average_precision(scores_sorted_labels, num_objs):
    """train.py:189-206 (`_compute_ap`): labels already ordered by descending score."""
    labels = np.asarray(scores_sorted_labels)
    tp = np.cumsum(labels == 1).astype(np.float32)
    fp = np.cumsum(labels == 0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        recall = tp / num_objs
        precision = tp / (fp + tp)
    precision = np.maximum.accumulate(precision[::-1])[::-1]          # monotone envelope from the right
    last = recall[-1] if recall.size else 0.0
    recall = np.concatenate(([0], recall, [last, 2]))
    precision = np.concatenate(([1], precision, [0, 0]))
    thresholds = np.linspace(0.0, 1.0, 101, endpoint=True)
    return float(np.average(precision[np.searchsorted(recall, thresholds, side="left")]) * 100)


def compute_aps(scores, classes, labels, roidb):
    """train.py:162-186 -> (mAP, multiclass_ap, per-class APs)."""
    order = np.argsort(-scores)
    labels, classes = labels[order], classes[order]
    num_objs = sum(int(np.sum(~np.asarray(r["gt_crowd"], bool))) for r in roidb)
    multiclass_ap = average_precision(labels, num_objs)
    cls_ap = []
    for cls in np.unique(classes):
        n_cls = sum(int(np.sum(~np.asarray(r["gt_crowd"], bool) & (np.asarray(r["gt_classes"]) == cls))) for r in roidb)
        cls_ap.append(average_precision(labels[classes == cls], n_cls))
    return float(np.mean(cls_ap)), multiclass_ap, cls_ap


def val_run(net, roidb):
    """train.py:133-159: run every image with GT, keep detections whose loss weight is > 0."""
    all_labels, all_scores, all_classes = [], [], []
    for roi in roidb:
        if "dets" not in roi or np.asarray(roi["dets"]).size == 0:
            continue
        net.run(roi, training=True, backward=False)
        w = net.weights.cpu().numpy()
        mask = w > 0.0
        all_labels.append(net.labels.cpu().numpy()[mask])
        all_scores.append(net.prediction.cpu().numpy()[mask])
        all_classes.append(np.asarray(roi["det_classes"])[mask])
    return compute_aps(np.concatenate(all_scores), np.concatenate(all_classes), np.concatenate(all_labels), roidb)


def rescore(net, roidb):
    """test.py:42-83: new scores (logits) for every image; returns (detections, seconds per image)."""
    import torch
    out, t_total, n_img, n_det = [], 0.0, 0, 0
    for roi in roidb:
        if "dets" not in roi or np.asarray(roi["dets"]).size == 0:
            continue
        feed = {k: roi[k] for k in ("dets", "det_scores", "det_classes")}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.run(feed)
        new_scores = net.prediction.cpu().numpy().copy()
        t_total += time.perf_counter() - t0
        n_img += 1
        n_det += len(new_scores)
        out.append({"id": roi.get("id", n_img - 1), "dets": np.asarray(roi["dets"]) / roi.get("im_scale", 1.0),
                    "det_classes": np.asarray(roi["det_classes"]), "det_scores": new_scores})
    return out, (t_total / max(n_img, 1)), (n_det / max(n_img, 1))


def save_dets(classes, class_to_cat_id, dets_as_dicts, output_file):
    """test.py:86-111; `classes` includes the background entry at index 0 like imdb['classes']."""
    cat_ids = [class_to_cat_id.get(cls, -1) for cls in classes]
    dets = [[] for _ in cat_ids]
    image_ids = []
    for d in dets_as_dicts:
        image_ids.append(d["id"])
        present = set(np.unique(d["det_classes"]).tolist())
        for ci in range(len(cat_ids)):
            if ci in present:
                m = d["det_classes"] == ci
                dets[ci].append(np.concatenate((d["dets"][m, :], d["det_scores"][m][:, None]), axis=1))
            else:
                dets[ci].append(np.zeros((0, 5), dtype=np.float32))
    with open(output_file, "wb") as fp:
        pickle.dump((dets, image_ids, cat_ids), fp, protocol=2)
    return dets, image_ids, cat_ids
