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
    factored form the kernels use -- per-detection tables of score x fc1 row for the centre and the neighbour role -- bit-exact."""
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
    # the score columns enter fc1 as one weight row per detection: tc[i] = score_i W1[class_i - 1] + b1 (centre role),
    # tn[i] = score_i W1[C + class_i - 1] (neighbour role) -- plain fp32 multiply / add, bit-exact against numpy
    n = batch["dets"].shape[0]
    tc = net.debug_view("pw_tc", n * 256).cpu().numpy().reshape(n, 256)
    tn = net.debug_view("pw_tn", n * 256).cpu().numpy().reshape(n, 256)
    w1 = net.variables["gnet/pw_feats/fc1/weights"].cpu().numpy()
    b1 = net.variables["gnet/pw_feats/fc1/biases"].cpu().numpy()
    sc = batch["det_scores"].astype(np.float32)[:, None]
    cls = (batch["det_classes"] - 1) if c > 1 else np.zeros(n, np.int64)
    assert np.array_equal(tc, sc * w1[cls] + b1[None, :])
    assert np.array_equal(tn, sc * w1[cp + cls])
    # and the dense one-hot x score block of the oracle's raw features times the same rows gives the same sums
    pairs = ref["neighbor_pair_idxs"]
    want = raw[:, :2 * cp].astype(np.float64) @ w1[:2 * cp].astype(np.float64) + b1
    got = tc[pairs[:, 0]].astype(np.float64) + tn[pairs[:, 1]]
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


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


@pytest.mark.parametrize("n,c,b,seed", [(300, 80, 3, 5), (1000, 1, 2, 1), (65, 80, 2, 0)])
def test_argmax_edges_recorded_for_backward(n, c, b, seed):
    """Training forward keeps, per (detection, column), the segment maximum, its tie count and an edge that attains it
    (the sparse backward routes the SegmentMax gradient through it; network.py:383-386); the backward preparation turns
    them into winner sets, a bitmap and a list.  All of it is checked EXACTLY against numpy reductions of the pw_fc2
    pre-activations the forward kernel itself computed (kept with keep_edge_activations), and those against the oracle."""
    from oracle.pins import winner_records_exact
    net, orc = make_pair(c, b)
    net.keep_edge_activations = True
    batch = make_image(n, c, seed=seed)
    ref = orc.forward(batch, keep=True)
    net.run(batch)
    torch.cuda.synchronize()
    for blk in range(1, b + 1):
        H = winner_records_exact(net, blk)
        assert rel_err(H, ref["pre"]["sel"][blk - 1]) < TOL


def _efw_begin(gw, tiles, waves):
    """Python restatement of common.hpp's efw_begin (edge_fwd_w's tile ranges, sized by dispatch layer)."""
    rb = lambda g, n, p: (g * n) // p
    if waves % 96 != 0 or tiles < 6 * waves:
        return rb(gw, tiles, waves)
    if gw >= waves:
        return tiles
    wpx = waves // 8
    tw = wpx // 3
    x, jx = divmod(gw, wpx)
    k, j = divmod(jx, tw)
    cut = (0, 405, 745, 1000)
    x0 = rb(x, tiles, 8)
    tx = rb(x + 1, tiles, 8) - x0
    b0, b1 = x0 + tx * cut[k] // 1000, x0 + tx * cut[k + 1] // 1000
    return b0 + rb(j, b1 - b0, tw)


@pytest.mark.parametrize("images", [1, 8])
def test_segment_record_flags_follow_the_edge_kernel_ranges(images):
    """A detection whose edges are split between two waves' tile ranges of edge_fwd_w (or that has none) must have its
    segment-max records start from zero (flag 1), every other one is written by a plain store (flag 0): the flags
    (edge_geometry, through efw_owner) and the ranges the kernel walks (efw_begin) have to describe the same partition --
    at 8 images the layered one, at 1 image the equal one.  Checked against a restatement of the ranges in Python."""
    from gossipnet_amd.network import DeviceBatch
    net, _ = make_pair(80, 2)
    dev = torch.device("cuda", 0)
    b = DeviceBatch([make_image(2000, 80, seed=i, preset="dense") for i in range(images)], dev)
    net.run(b, training=False)
    torch.cuda.synchronize()
    E, N = int(net.num_edges), int(net.num_dets)
    tiles = (E + 31) // 32
    waves = max(1, min(3 * 256, (tiles + 3) // 4)) * 4
    begins = np.array([_efw_begin(g, tiles, waves) for g in range(waves + 1)], dtype=np.int64)
    assert begins[0] == 0 and begins[-1] == tiles and np.all(np.diff(begins) >= 0)
    assert (tiles >= 6 * waves) == (images == 8), "the batch takes the layered ranges, a single image the equal ones"
    rp = net.row_ptr.cpu().numpy().astype(np.int64)
    owner = lambda t: np.searchsorted(begins, t, side="right") - 1           # the wave whose range holds tile t
    eb, ee = rp[:-1], rp[1:]
    want = (ee == eb) | (owner(eb >> 5) != owner(np.maximum(ee - 1, 0) >> 5))
    got = net._view(net._buf.scratch_i, N, torch.int32).cpu().numpy()
    assert np.array_equal(got != 0, want)
    if images == 8:      # the layers really differ in size
        per = np.diff(begins)
        assert per[:waves // 24].mean() > 1.15 * per[2 * waves // 24:3 * waves // 24].mean()
