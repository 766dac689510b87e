"""GPU parity: HIP forward path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact: neighbour indices, det_det_iou at the pairs.  fp32 tensors: 1e-5 (relative to max(1, |ref|max))."""
import numpy as np
import pytest
import torch

from tests.util import make_pair, rel_err, make_image

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.mark.parametrize("n,c,b,seed", [(6, 1, 1, 0), (40, 1, 2, 1), (64, 80, 2, 2), (300, 80, 16, 0), (300, 1, 1, 1),
                                         (1000, 1, 16, 0), (33, 80, 3, 5), (1, 80, 2, 0)])
def test_forward_parity(n, c, b, seed):
    net, orc = make_pair(c, b)
    batch = make_image(n, c, seed=seed)
    ref = orc.forward(batch, with_loss=False, keep=True)
    net.run(batch, training=True, backward=False)
    torch.cuda.synchronize()
    pairs = net.neighbor_pair_idxs.cpu().numpy()
    assert np.array_equal(pairs, ref["neighbor_pair_idxs"]), "neighbour indices must be bit-exact"
    ious = ref["det_det_iou"][pairs[:, 0], pairs[:, 1]]
    assert np.array_equal(net.edge_iou.cpu().numpy(), ious)
    assert rel_err(net.pw_feats.cpu().numpy(), ref["pw_feats"].detach().numpy()) < TOL
    bf = net.block_feats
    for k in range(1, b + 1):
        assert rel_err(bf[k].cpu().numpy(), ref["block_feats"][k].detach().numpy()) < TOL, "block %d" % k
    assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < TOL


def test_forward_inference_mode_matches_training_mode():
    net, orc = make_pair(80, 4)
    batch = make_image(200, 80, seed=3)
    net.run(batch, training=True, backward=False)
    p1 = net.prediction.cpu().numpy().copy()
    infer = {k: batch[k] for k in ("dets", "det_scores", "det_classes")}
    net.run(infer)
    assert np.array_equal(net.prediction.cpu().numpy(), p1)


def test_forward_multi_image_batch_equals_per_image():
    net, orc = make_pair(80, 3)
    imgs = [make_image(n, 80, seed=s) for n, s in ((120, 0), (1, 1), (75, 2))]
    net.run(imgs, training=True, backward=False)
    pred = net.prediction.cpu().numpy().copy()
    off = 0
    for im in imgs:
        ref = orc.forward(im, with_loss=False)
        k = im["dets"].shape[0]
        assert rel_err(pred[off:off + k], ref["prediction"].detach().numpy()) < TOL
        off += k
