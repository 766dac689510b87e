"""Per-class loss weights fed to Gnet(class_weights=...) -- the reference's `class_equal_weights`
(nms_net/class_weights.py:12-22 with imdb/tools.py:113-124 `get_class_counts`).

weights[k] = num_samples * expected[k] / count[k], expected = [1 - pos_weight, pos_weight / C, ...]: every class
(background = 0 included) carries its expected share of the loss whatever its frequency.  Counts start at one
(imdb/tools.py:114), every ground-truth box counts for its class, and an image's detections in excess of its
ground-truth boxes count as background (imdb/tools.py:121-123).
"""
import numpy as np

from .config import cfg


def get_class_counts(imdb):
    """int64 [C+1] frequency table of an imdb dict {'num_classes': C, 'roidb': [roi, ...]}."""
    n = int(imdb['num_classes']) + 1
    freq = np.ones(n, np.int64)
    for roi in imdb['roidb']:
        num_pos = 0
        if 'gt_classes' in roi:
            gt = np.asarray(roi['gt_classes']).reshape(-1).astype(np.int64)
            num_pos = gt.size
            if gt.size and (gt.min() < -n or gt.max() >= n):
                # the reference's loop `freq[cls] += 1` (imdb/tools.py:118-120) raises for a class outside [-n, n) ...
                raise IndexError("gt_classes outside [%d, %d]: %d .. %d" % (-n, n - 1, int(gt.min()), int(gt.max())))
            np.add.at(freq, gt, 1)        # ... and wraps a negative class the numpy way, like `freq[cls] += 1` does
        if 'det_classes' in roi:
            freq[0] += max(0, int(np.asarray(roi['det_classes']).size) - num_pos)
    return freq


def class_equal_weights(imdb):
    num_classes = int(imdb['num_classes'])
    pos = cfg.train.pos_weight
    expected = np.full(num_classes + 1, pos / num_classes, np.float32)
    expected[0] = 1 - pos
    counts = get_class_counts(imdb)
    return counts.sum() * expected / counts
