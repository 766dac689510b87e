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
    if n <= 300:      # the dense matrix (Gnet.det_det_iou, network.py:176) on demand, bit-exact (NaN for degenerate pairs alike)
        assert np.array_equal(net.det_det_iou.cpu().numpy(), ref["det_det_iou"], equal_nan=True)
    assert rel_err(net.pw_feats.cpu().numpy(), ref["pw_feats"].detach().numpy()) < TOL
    bf = net.block_feats
    for k in range(1, b + 1):
        assert rel_err(bf[k].cpu().numpy(), ref["block_feats"][k].detach().numpy()) < TOL, "block %d" % k
    assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < TOL


@pytest.mark.parametrize("c", [1, 80])
def test_geometry_and_score_columns_vs_oracle(c):
    """SURVEY 8a F1 (`_geometry_feats`, network.py:411-454) on its own: the 7 geometry columns the pw-MLP reads
    (within a few ulp: logf / sqrtf / IEEE division on the device vs numpy) and the one-hot x score columns in the
    factored form the kernels use -- (row, score) of the centre and of the neighbour -- bit-exact."""
    net, orc = make_pair(c, 1)
    batch = make_image(300, c, seed=9)
    ref = orc.forward(batch)
    net.run(batch, training=True, backward=False)
    torch.cuda.synchronize()
    raw = ref["raw_pw_feats"]
    E, cp = raw.shape[0], max(c, 1)
    geo = net.debug_view("geo", E * 8).cpu().numpy().reshape(E, 8)
    assert rel_err(geo[:, :7], raw[:, 2 * cp:]) < 2e-6
    assert np.array_equal(geo[:, 0], raw[:, 2 * cp])                       # the IoU column is the graph's: bit-exact
    info = net.debug_view("einfo", E * 4, dtype=torch.int32).cpu().numpy().reshape(E, 4)
    dense = np.zeros((E, 2 * cp), np.float32)
    dense[np.arange(E), info[:, 0]] = info[:, 2].view(np.float32)
    dense[np.arange(E), info[:, 1]] += info[:, 3].view(np.float32)
    assert (info[:, 0] < cp).all() and (info[:, 1] >= cp).all()
    assert np.array_equal(dense, raw[:, :2 * cp])


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


def test_argmax_edges_recorded_for_backward():
    """Training forward keeps, per (detection, column), the first edge that attains the segment maximum
    (the sparse backward routes the SegmentMax gradient through it; network.py:383-386)."""
    net, orc = make_pair(80, 3)
    net.keep_edge_activations = True
    batch = make_image(300, 80, seed=5)
    net.run(batch)
    torch.cuda.synchronize()
    n, e = 300, net.num_edges
    rp = net._view(net._buf.row_ptr, n + 1, torch.int32).cpu().numpy()
    for blk in (1, 3):
        pm = net._view(net._buf.blk_pm[blk], n * 64, torch.int64).view(n, 64).cpu().numpy()
        pa = net._view(net._buf.blk_parg[blk], n * 64, torch.int64).view(n, 64).cpu().numpy()
        h1 = net._view(net._buf.blk_h1[blk], e * 64, torch.float32).view(e, 64).cpu().numpy().astype(np.float64)
        w2 = net.variables["gnet/block%d/pw_fc2/weights" % blk].cpu().numpy().astype(np.float64)
        b2 = net.variables["gnet/block%d/pw_fc2/biases" % blk].cpu().numpy().astype(np.float64)
        h2 = np.maximum(h1 @ w2 + b2, 0.0)
        mx = (pm >> 32).astype(np.uint32).view(np.float32) if False else np.frombuffer((pm >> 32).astype(np.uint32).tobytes(), np.float32).reshape(n, 64)
        assert np.array_equal(pm >> 32, pa >> 32)                     # same maxima in both records
        arg = (pa & 0xffffffff).astype(np.int64)
        pos = mx > 0
        c_idx = np.repeat(np.arange(n)[:, None], 64, 1)
        assert np.all(arg[pos] >= rp[c_idx[pos]]) and np.all(arg[pos] < rp[c_idx[pos] + 1])   # an edge of that detection
        j_idx = np.repeat(np.arange(64)[None, :], n, 0)
        assert np.abs(h2[arg[pos], j_idx[pos]] - mx[pos]).max() < 1e-5
