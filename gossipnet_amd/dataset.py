"""Feeding the hot path: the reference's nms_net/dataset.py (`load_roi` :17-43, `TestDataset` :68-84,
`ShuffledDataset` :87-112) for the detection-only networks -- what train.py:88-115 and test.py:58-71 hand to the Gnet.

Differences that follow from the engine, not from the data: `ShuffledDataset(batch_size=k)` may return k images per step
(the reference asserts 1, dataset.py:108: one image per `sess.run`; here a step is a block-diagonal batch of images); the
TF FIFOQueue `Prefetcher` (dataset.py:115-140) has no counterpart -- Gnet.run takes the dicts directly and uploads them
on its own stream.  Image decoding / resizing (`load_image` :46-65, scipy.misc) feeds the ResNet trunk, which is outside
this path: with need_images the roi must already carry the trunk's feature map under 'imfeats'
(and its 'im_scale'); dets and gt_boxes are multiplied by that scale exactly as dataset.py:37-41 does.
"""
import numpy as np


def load_roi(need_images, roi, is_training=False):
    roi = dict(roi)                       # never modify the imdb's record
    im_scale = 1.0
    if need_images:
        if 'imfeats' not in roi:
            raise ValueError("need_images: the roi carries no 'imfeats' feature map (the image trunk is not part of "
                             "this engine; supply the stride-16 map and 'im_scale')")
        im_scale = float(roi.get('im_scale', 1.0))
        for k in ('dets', 'gt_boxes'):
            if k in roi:
                roi[k] = np.asarray(roi[k]) * im_scale          # not in place
    roi['im_scale'] = im_scale
    return roi


class TestDataset(object):
    """Sequential pass over imdb['roidb'], one image per call (test.py:60-66)."""
    __test__ = False                       # not a pytest class

    def __init__(self, imdb, batch_size, need_images):
        if batch_size != 1:
            raise ValueError("TestDataset yields one image per call")
        self._roidb = imdb['roidb']
        self._need_images = need_images
        self._cur = 0

    def next_batch(self):
        roi = load_roi(self._need_images, self._roidb[self._cur], is_training=False)
        self._cur += 1
        return roi

    def __len__(self):
        return len(self._roidb)


class ShuffledDataset(object):
    """Epochs of a random permutation; a permutation that cannot fill the next step is discarded and redrawn
    (dataset.py:101-103).  batch_size 1 returns the roi dict (the reference's shape), k > 1 a list of k dicts."""

    def __init__(self, imdb, batch_size, need_images, rng=None):
        self._roidb = imdb['roidb']
        self._batch_size = int(batch_size)
        if not 1 <= self._batch_size <= len(self._roidb):
            raise ValueError("batch_size must lie in [1, len(roidb)]")
        self._need_images = need_images
        self._rng = np.random if rng is None else rng           # np.random: the reference's global stream
        self._shuffle()

    def _shuffle(self):
        self._perm = self._rng.permutation(len(self._roidb))
        self._cur = 0

    def next_batch(self):
        if self._cur + self._batch_size > self._perm.size:
            self._shuffle()
        inds = self._perm[self._cur:self._cur + self._batch_size]
        self._cur += self._batch_size
        rois = [load_roi(self._need_images, self._roidb[i], is_training=True) for i in inds]
        return rois[0] if self._batch_size == 1 else rois
