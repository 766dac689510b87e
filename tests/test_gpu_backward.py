"""GPU parity of loss + backward (through the C ABI) against the CPU oracle.

The loss is piecewise smooth (ReLU masks, segment-max arg-max): two fp32 implementations whose
forward values differ by 1e-7 can sit on different sides of a kink, which changes single gradient
entries by O(1).  The acceptance test therefore PINS the smooth piece: the oracle's backward pass takes
the ReLU masks and the segment-max winner sets of the HIP forward pass (tests/util.gpu_pins ->
GnetOracle.forward(pins=...)), so that both sides differentiate the same piece, and then EVERY parameter
tensor of EVERY seed must agree to 1e-5 of its max |gradient| (TF semantics: ReluGrad = g * (out > 0),
_SegmentMinOrMaxGrad tie split; network.py:387-388).  The unpinned comparison is kept as a diagnostic:
every one of its > 2e-5 outliers has to disappear under pinning."""
import numpy as np
import pytest
import torch

from oracle import gnet_oracle as go
from tests.util import make_pair, rel_err, make_image, grad_errors, gpu_pins
from oracle.pins import mask_disagreements, winner_records_exact

pytestmark = pytest.mark.gpu

TIGHT = 2e-5      # unpinned agreement that counts as "no kink was crossed"
PINNED = 1e-5     # acceptance bar of the mask-pinned comparison, per parameter tensor


def pinned_errors(net, orc, batch, c, b, image=None):
    """Per-tensor gradient errors against the oracle differentiating the HIP forward's smooth piece."""
    _, gpin = orc.forward_backward(batch, pins=gpu_pins(net, image))
    return grad_errors(net, gpin, c, b)


def check_outputs(net, ref):
    assert np.array_equal(net.det_anno_iou.cpu().numpy(), ref["det_anno_iou"]), "det_anno_iou bit-exact"
    assert np.array_equal(net.det_gt_matching.cpu().numpy(), ref["det_gt_matching"]), "assignments bit-exact"
    assert np.array_equal(net.labels.cpu().numpy(), ref["labels"])
    assert rel_err(net.weights.cpu().numpy(), ref["weights"].numpy()) < 1e-6
    assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < 1e-5
    assert abs(float(net.loss) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))
    assert abs(float(net.loss_normed) - float(ref["loss_normed"])) <= 1e-5 * max(1.0, abs(float(ref["loss_normed"])))


KINK = 2e-6       # a mask may differ from the oracle's own only where the oracle is this close (x layer scale) to the kink.
                  # Measured on MI355X (profiles/r03_kinks.txt): <= 2.7e-7 over every test of this file, the headline image
                  # (33 differing entries of 3.7e8) included; a wrong winner rule is off by the size of the activations.


# kink-free seeds (of 6) measured on MI355X per case, minus one (a change of a kernel's summation order may move a unit
# across its kink); the count is deterministic for fixed kernels.  Filled from the run recorded in profiles/r04_kinks.txt.
# History: round 4 lowered two floors -- (200, 1, 1, 0.01) 4 -> 2 and (150, 80, 3, 0.01) 4 -> 3 -- when pw_feats/fc1 went from an
# fma chain with the bias added last to tc[c] + tn[n] + a K = 8 MFMA product (another rounding order: more units of the 0.01-bias
# cases land within 1e-7 of their kink on the other side); round 5's pw_fwd2 keeps that order for fc1 and changes fc2 / fc3's
# (bias added to the finished sum, eight K = 32 partial sums for fc3): the floors held.  The count is a REGRESSION GUARD for "most
# seeds need no pinning at all"; round 5's edge_fwd_w forms its fp32 products as six bf16 products of three-term splits (another
# rounding of the same sums): measured 6 6 6 6 5 6 6 5 5 6 4 5 4 5 4 4 in the order of the table -- three floors went UP (64,80,1,.01: 3 -> 4;
# 64,80,2,.01: 2 -> 3; 200,1,1,.01: 2 -> 3), two down (64,80,2,.5: 5 -> 4; 150,80,3,.5: 5 -> 3), worst kink distance 2.4e-7 as before.
# The acceptance criteria are the two asserts on every seed -- worst kink distance <= KINK and
# pinned gradients <= PINNED -- which no kernel change has ever loosened.
KINK_FREE_MIN = {(6, 1, 1, 0.01): 5, (6, 1, 1, 0.5): 5, (20, 1, 1, 0.01): 5, (20, 1, 1, 0.5): 5, (33, 1, 2, 0.01): 4, (33, 1, 2, 0.5): 5,
                 (64, 1, 2, 0.01): 5, (64, 1, 2, 0.5): 4, (64, 80, 1, 0.01): 4, (64, 80, 1, 0.5): 5, (64, 80, 2, 0.01): 3,
                 (64, 80, 2, 0.5): 4, (200, 1, 1, 0.01): 3, (200, 1, 1, 0.5): 4, (150, 80, 3, 0.01): 3, (150, 80, 3, 0.5): 3}


def kink_report(net, ref, image=None):
    """(n_diff, worst, where) of oracle.pins.mask_disagreements for the HIP forward pass against the oracle's own."""
    return mask_disagreements(gpu_pins(net, image), ref)


@pytest.mark.parametrize("bias", [0.01, 0.5])
@pytest.mark.parametrize("n,c,b", [(6, 1, 1), (20, 1, 1), (33, 1, 2), (64, 1, 2), (64, 80, 1), (64, 80, 2), (200, 1, 1),
                                   (150, 80, 3)])
def test_backward_parity_small(n, c, b, bias):
    """bias 0.01 = the experiments' init (ReLU pre-activations near 0 do occur); bias 0.5 keeps most units active.
    Three assertions per seed, none of which takes the device's word for the smooth piece:
      1. the device's ReLU masks and SegmentMax winner sets equal the oracle's OWN, except at entries where the oracle's
         own pre-activation (or its distance from the segment maximum) is within KINK of the kink -- a winner rule that
         picked a non-maximal edge or dropped a tied one fails here;
      2. when no entry differs at all, the UNPINNED gradient comparison must be tight for every tensor;
      3. on the device's piece (the oracle differentiating the masks of 1.) every tensor agrees to 1e-5.
    Most seeds are kink-free (measured on MI355X: see the assertion at the end)."""
    cw = np.linspace(0.5, 1.5, c + 1).astype(np.float32)
    net, orc = make_pair(c, b, class_weights=cw, bias=bias)
    net.keep_edge_activations = True
    n_free, worst_all = 0, 0.0
    for seed in range(6):
        batch = make_image(n, c, seed=seed)
        ref, gref = orc.forward_backward(batch, keep=True)
        net.run(batch)
        torch.cuda.synchronize()
        check_outputs(net, ref)
        assert not np.isnan(net.grads.cpu().numpy()).any()
        n_diff, worst, where = kink_report(net, ref)
        assert worst <= KINK, (seed, n_diff, worst, where)
        worst_all = max(worst_all, worst)
        unpinned = grad_errors(net, gref, c, b)
        if n_diff == 0:
            assert max(unpinned.values()) <= TIGHT, (seed, max(unpinned.items(), key=lambda kv: kv[1]))
            n_free += 1
        pinned = pinned_errors(net, orc, batch, c, b)
        assert max(pinned.values()) <= PINNED, (seed, max(pinned.items(), key=lambda kv: kv[1]))
    print("kink-free seeds: %d / 6, worst distance from a kink at a differing mask entry %.2e  (n=%d c=%d b=%d bias=%g)"
          % (n_free, worst_all, n, c, b, bias))
    assert n_free >= KINK_FREE_MIN.get((n, c, b, bias), 0), "kink-free seeds %d" % n_free


@pytest.mark.parametrize("c", [80, 1])
def test_tiny_steps_follow_the_stated_gradient_rule(c):
    """Steps of two to four mutually overlapping detections: the largest entry of a head gradient is a cancelled sum there
    (+0.2542 - 0.2516), so one ulp of a summand is ~1e-5 of the result.  The criterion, stated once (oracle/pins.py fp64_rule,
    DESIGN.md 2): every tensor <= PINNED of the fp32 oracle on the pinned piece (an absolute floor of 5e-7 under the
    normalisation, as tools/fuzz_parity.py) -- or, for steps of at most four detections ONLY, the device no further from the
    oracle's fp64 twin than the fp32 oracle itself is, plus PINNED.  A step of five or more detections never takes the
    exception (asserted on the rule itself)."""
    from oracle.pins import fp64_rule, TINY_STEP_DETS
    net, orc = make_pair(c, 2, class_weights=np.linspace(0.5, 1.5, c + 1).astype(np.float32))
    net.keep_edge_activations = True
    o64 = go.GnetOracle(c, 2, params={k: v.detach().numpy() for k, v in orc.params.items()}, dtype=torch.float64,
                        class_weights=orc.class_weights.numpy())
    rng = np.random.default_rng(5)
    settled, worst = 0, 0.0
    for case in range(24):
        n = int(rng.integers(2, TINY_STEP_DETS + 1))
        im = make_image(n, c, seed=int(rng.integers(1 << 30)))
        im["det_classes"][:] = 1; im["gt_classes"][:] = 1
        im["dets"] = (im["dets"][:1] + np.abs(rng.normal(0, 0.5, (n, 4)))).astype(np.float32)      # everything overlaps everything
        im["dets"][:, 2:] += 8.0
        net.run(im)
        torch.cuda.synchronize()
        ref, _ = orc.forward_backward(im, keep=True)
        check_outputs(net, ref)
        n_diff, w, where = kink_report(net, ref)
        assert w <= KINK, (case, n_diff, w, where)
        pins = gpu_pins(net)
        _, g32 = orc.forward_backward(im, pins=pins)
        g64 = None
        for name, _shape in go.param_spec(c, 2):
            g = net.gradients[name].detach().cpu().numpy().reshape(-1).astype(np.float64)
            gr = np.asarray(g32[name], np.float64).reshape(-1)
            err = float(np.abs(g - gr).max() / (np.abs(gr).max() + 5e-2))
            worst = max(worst, err)
            if err <= PINNED:
                continue
            if g64 is None:
                _, g64 = o64.forward_backward(im, pins=pins)
            ok, e_dev, e_f32 = fp64_rule(g, gr, g64[name], n, PINNED, floor=5e-2)
            assert ok, (case, n, name, err, e_dev, e_f32)
            settled += 1
    print("tiny steps (c = %d): worst pinned error %.2e; %d tensor(s) settled by the fp64 rule" % (c, worst, settled))
    assert settled <= 2, "the exception is rare even among tiny steps"
    z = np.zeros(3)
    assert fp64_rule(z, z, z, TINY_STEP_DETS + 1)[0] is False and fp64_rule(z, z, z, TINY_STEP_DETS)[0]


def first_winner_only(sel, c_idx):
    """The winner sets a WRONG implementation would use: of every (detection, column)'s tied edges only the first."""
    sel = np.asarray(sel)
    out = np.zeros_like(sel)
    e, j = np.nonzero(sel)
    key = c_idx[e].astype(np.int64) * sel.shape[1] + j
    _, first = np.unique(key, return_index=True)             # (e ascending within equal keys: nonzero() is row-major)
    out[e[first], j[first]] = True
    return out


@pytest.mark.parametrize("c,bias", [(80, 0.5), (1, 0.5), (80, 0.01)])
def test_tied_columns_split_evenly_across_distinct_edges(c, bias):
    """SegmentMax ties between edges with DIFFERENT inputs (TF _SegmentMinOrMaxGrad: every tied edge receives
    grad / count; network.py:387-388).  Two columns of pw_fc2 get zero weights and a positive bias in both blocks, so
    that EVERY edge of a detection ties on them (h2[:, j] = b2[j] exactly) while the pw_fc1 rows behind the edges all
    differ: d W2[:, j] = sum_e h1[e] * dp[c_e, j] / degree(c_e) under the even split, but h1[first edge] * dp[c, j] under
    a first-winner rule -- the parameter gradients now tell the two apart (with duplicated detections they cannot).
    The oracle differentiates ITS OWN winner sets (nothing read back from the device's tie records)."""
    b, n = 2, 48
    tied = (5, 40)                                              # one column in each 32-column half of the MFMA tile
    net, orc = make_pair(c, b, bias=bias)
    net.keep_edge_activations = True
    with torch.no_grad():
        for blk in (1, 2):
            for j in tied:
                net.variables["gnet/block%d/pw_fc2/weights" % blk][:, j] = 0.0
                net.variables["gnet/block%d/pw_fc2/biases" % blk][j] = 0.25 + 0.5 * blk
                orc.params["gnet/block%d/pw_fc2/weights" % blk].data[:, j] = 0.0
                orc.params["gnet/block%d/pw_fc2/biases" % blk].data[j] = 0.25 + 0.5 * blk
    names = ["gnet/block%d/pw_fc2/weights" % k for k in (1, 2)] + ["gnet/block%d/pw_fc2/biases" % k for k in (1, 2)]
    for seed in range(4):
        batch = make_image(n, c, seed=seed)
        ref, gref = orc.forward_backward(batch, keep=True)
        c_idx = ref["neighbor_pair_idxs"][:, 0]
        deg = np.bincount(c_idx, minlength=n)
        assert (deg > 1).sum() >= n // 2, "the image must have neighbours"
        for blk in (0, 1):                                      # the oracle sees the ties: every edge, both columns
            assert ref["pins"]["sel"][blk][:, list(tied)].all()
        net.run(batch)
        torch.cuda.synchronize()
        check_outputs(net, ref)
        # the device's tie records against numpy reductions of the values its own kernel saw: exact
        for blk in (1, 2):
            H = winner_records_exact(net, blk)
            assert rel_err(H, ref["pre"]["sel"][blk - 1]) < 1e-5
            pm = net.debug_view("blk_pm", n * 64, dtype=torch.int64, index=blk).view(n, 64).cpu().numpy()
            assert np.array_equal((pm & 0xffffffff)[:, list(tied)], np.stack([deg, deg], 1)), "tie count = degree"
        n_diff, worst, where = kink_report(net, ref)
        assert worst <= KINK, (seed, n_diff, worst, where)
        # (1) ReLU masks from the device, winner sets the ORACLE'S OWN: every tensor, in particular the tied columns
        pins = gpu_pins(net)
        pins["sel"] = ref["pins"]["sel"]
        _, g_own_sel = orc.forward_backward(batch, pins=pins)
        errs = grad_errors(net, g_own_sel, c, b)
        assert max(errs.values()) <= PINNED, (seed, max(errs.items(), key=lambda kv: kv[1]))
        dev = {k: net.gradients[k].cpu().numpy() for k in names}
        for k in names[:2]:
            for j in tied:
                col, want = dev[k][:, j], g_own_sel[k][:, j]
                assert np.abs(col - want).max() <= PINNED * np.abs(g_own_sel[k]).max(), (seed, k, j)
        # (2) fully unpinned whenever no mask differs
        if n_diff == 0:
            unp = grad_errors(net, gref, c, b)
            assert max(unp.values()) <= TIGHT, (seed, max(unp.items(), key=lambda kv: kv[1]))
        # (3) the test discriminates: a first-winner rule gives a DIFFERENT d W2 in the tied columns (and would fail (1))
        wrong = dict(pins)
        wrong["sel"] = [first_winner_only(sel_, c_idx) for sel_ in ref["pins"]["sel"]]
        _, g_first = orc.forward_backward(batch, pins=wrong)
        gap = max(np.abs(g_first[k][:, j] - g_own_sel[k][:, j]).max() / np.abs(g_own_sel[k]).max() for k in names[:2] for j in tied)
        assert gap > 100 * PINNED, "first-winner and even split must differ here (else the test proves nothing): %g" % gap
        bad = max(np.abs(dev[k][:, j] - g_first[k][:, j]).max() / np.abs(g_own_sel[k]).max() for k in names[:2] for j in tied)
        assert bad > 100 * PINNED


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 65, 95, 129, 257])
def test_tile_tails_forward_and_pinned_backward(n):
    """Detection counts around the 32-row tile size of the node kernels (and below one tile, one detection): every
    output and every gradient tensor against the oracle."""
    for c in (1, 80):
        net, orc = make_pair(c, 2)
        net.keep_edge_activations = True
        batch = make_image(n, c, seed=100 + n)
        ref, gref = orc.forward_backward(batch)
        net.run(batch)
        torch.cuda.synchronize()
        check_outputs(net, ref)
        pinned = pinned_errors(net, orc, batch, c, 2)
        assert max(pinned.values()) <= PINNED, (n, c, max(pinned.items(), key=lambda kv: kv[1]))


@pytest.mark.parametrize("n,c,b,seed", [(300, 80, 16, 0), (300, 80, 16, 1), (1000, 1, 16, 0)])
def test_backward_parity_pinned_16_blocks(n, c, b, seed):
    net, orc = make_pair(c, b)
    net.keep_edge_activations = True
    batch = make_image(n, c, seed=seed)
    ref = orc.forward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    check_outputs(net, ref)
    pinned = pinned_errors(net, orc, batch, c, b)
    assert max(pinned.values()) <= PINNED, max(pinned.items(), key=lambda kv: kv[1])


def test_headline_config_single_image():
    """BASELINE configs[2] at full size: N = 2000, 80 classes, 16 blocks, one image (the reference's step
    shape, train.py:115).  Integer results bit-exact; fp32 tensors <= 1e-5; gradients <= 1e-5 on the pinned piece."""
    n, c, b = 2000, 80, 16
    net, orc = make_pair(c, b)
    net.keep_edge_activations = True
    batch = make_image(n, c, seed=0)
    ref = orc.forward(batch, keep=True)
    net.run(batch)
    torch.cuda.synchronize()
    pairs = net.neighbor_pair_idxs.cpu().numpy()
    assert np.array_equal(pairs, ref["neighbor_pair_idxs"])
    assert np.array_equal(net.edge_iou.cpu().numpy(), ref["det_det_iou"][pairs[:, 0], pairs[:, 1]])
    check_outputs(net, ref)
    assert rel_err(net.pw_feats.cpu().numpy(), ref["pw_feats"].detach().numpy()) < 1e-5
    bf = net.block_feats
    for k in range(1, b + 1):
        assert rel_err(bf[k].cpu().numpy(), ref["block_feats"][k].detach().numpy()) < 1e-5, "block %d" % k
    n_diff, worst, where = kink_report(net, ref)
    print("headline image: %d mask entries differ from the oracle's own, worst distance %.2e at %s" % (n_diff, worst, where))
    assert worst <= KINK, (n_diff, worst, where)
    for blk in (1, 8, 16):
        H = winner_records_exact(net, blk)
        assert rel_err(H, ref["pre"]["sel"][blk - 1]) < 1e-5
    pinned = pinned_errors(net, orc, batch, c, b)
    assert max(pinned.values()) <= PINNED, max(pinned.items(), key=lambda kv: kv[1])


@pytest.mark.parametrize("preset", ["dense", "coco_like"])
def test_headline_bench_batch_8_images(preset):
    """The batch bench.py times: 8 images x N = 2000, C = 80, B = 16 as one block-diagonal graph
    (dense preset: E ~ 1.44 M, the XCD-aware range mapping and the 32-bit byte offsets are live at this size;
    coco_like preset, bench.py's other_configs line: E/N ~ 34 -- fewer tiles per wave, the equal (not layered)
    edge_fwd_w ranges, a different winner fraction; reference experiments/coco_multiclass/conf.yaml, 16 stacked
    blocks network.py:344-409).  Per image: edges / det_anno_iou / assignments bit-exact, activations and logits
    <= 1e-5; the batch gradient <= 1e-5 against the sum of the per-image oracle gradients on the pinned piece."""
    n, c, b, k_img = 2000, 80, 16, 8
    net, orc = make_pair(c, b)
    net.keep_edge_activations = True
    imgs = [make_image(n, c, seed=i, preset=preset) for i in range(k_img)]
    net.run(imgs)
    torch.cuda.synchronize()
    print("%s preset: 8 x N=2000, E = %d (E/N %.1f)" % (preset, net.num_edges, net.num_edges / (n * k_img)))
    pairs_all = net.neighbor_pair_idxs.cpu().numpy()
    rp = net.row_ptr.cpu().numpy()
    pred = net.prediction.cpu().numpy(); assign = net.det_gt_matching.cpu().numpy(); labels = net.labels.cpu().numpy()
    pw = net.pw_feats.cpu().numpy(); bf = [None] + [x.cpu().numpy() for x in net.block_feats[1:]]
    anno = net.det_anno_iou
    losses = net.image_losses[:, 0].cpu().numpy()
    gsum = None
    for i, im in enumerate(imgs):
        d0, d1 = i * n, (i + 1) * n
        e0, e1 = rp[d0], rp[d1]
        out, g = orc.forward_backward(im, pins=gpu_pins(net, image=i))
        assert np.array_equal(pairs_all[e0:e1] - d0, out["neighbor_pair_idxs"]), "image %d" % i
        assert np.array_equal(anno[i].cpu().numpy(), out["det_anno_iou"])
        assert np.array_equal(assign[d0:d1], out["det_gt_matching"]) and np.array_equal(labels[d0:d1], out["labels"])
        assert rel_err(pw[e0:e1], out["pw_feats"].detach().numpy()) < 1e-5
        for k in range(1, b + 1):
            assert rel_err(bf[k][d0:d1], out["block_feats"][k].detach().numpy()) < 1e-5
        assert rel_err(pred[d0:d1], out["prediction"].detach().numpy()) < 1e-5
        assert abs(losses[i] - float(out["loss"])) <= 1e-5 * max(1.0, abs(float(out["loss"])))
        gsum = g if gsum is None else {k_: gsum[k_] + g[k_] for k_ in g}
    errs = grad_errors(net, gsum, c, b)
    assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])


def test_normalize_loss_and_multiplier():
    from gossipnet_amd.config import cfg
    net, orc = make_pair(1, 1, normalize_loss=True)
    batch = make_image(20, 1, seed=0)
    ref, gref = orc.forward_backward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    check_outputs(net, ref)
    assert max(grad_errors(net, gref, 1, 1).values()) < TIGHT


def test_batch_gradient_is_sum_of_image_gradients():
    """Size-independent property: a block-diagonal batch = independent images."""
    net, orc = make_pair(80, 2)
    imgs = [make_image(n, 80, seed=s) for n, s in ((90, 0), (40, 1), (130, 2))]
    total = torch.zeros_like(net.grads)
    losses = []
    for im in imgs:
        net.run(im)
        total += net.grads
        losses.append(float(net.loss))
    net.run(imgs)
    torch.cuda.synchronize()
    assert np.allclose(net.image_losses[:, 0].cpu().numpy(), losses, rtol=1e-6, atol=1e-6)
    scale = float(total.abs().max())
    assert float((net.grads - total).abs().max()) <= 2e-5 * scale


def test_no_gt_image():
    """n_gt = 0: every detection is a negative with weight 1 (det_matching.cc, network.py:286-293)."""
    net, orc = make_pair(80, 1)
    net.keep_edge_activations = True
    batch = make_image(30, 80, seed=0)
    batch["gt_boxes"] = np.zeros((0, 4), np.float32)
    batch["gt_crowd"] = np.zeros(0, bool)
    batch["gt_classes"] = np.zeros(0, np.int32)
    ref, gref = orc.forward_backward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    assert net.det_gt_matching.cpu().tolist() == [-1] * 30
    assert abs(float(net.loss) - float(ref["loss"])) < 1e-4
    assert max(pinned_errors(net, orc, batch, 80, 1).values()) <= PINNED


def test_weight_reg_gradient():
    net, orc = make_pair(1, 1)
    net.weight_reg = 0.0005
    batch = make_image(12, 1, seed=1)
    net.run(batch)
    g1 = net.grads.clone()
    net.weight_reg = None
    net.run(batch)
    diff = (g1 - net.grads)
    expect = 0.0005 * net.params * net._reg_mask
    # (gradients are bitwise reproducible, test_gradients_bitwise_reproducible: diff is the regulariser only,
    #  up to the rounding of g + wd*w - g)
    assert float((diff - expect).abs().max()) < 2e-6 * float(net.grads.abs().max())


def test_weight_reg_multi_image_batch():
    """The l2 regulariser enters ONCE per step (train.py:231-238: data loss + sum of l2 terms) whatever the number
    of images and the data-loss scale: grads = grad_scale * sum_i g_i + weight_reg * w on the regularised weights."""
    c, b = 80, 2
    net, orc = make_pair(c, b)
    net.keep_edge_activations = True
    imgs = [make_image(n, c, seed=s_) for n, s_ in ((60, 0), (45, 1), (80, 2))]
    net.weight_reg = 0.0005
    net.grad_scale = 1.0 / 3
    net.run(imgs)
    torch.cuda.synchronize()
    gsum = None
    for i, im in enumerate(imgs):
        _, g = orc.forward_backward(im, pins=gpu_pins(net, image=i))
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    want = {}
    for k, v in gsum.items():
        w = orc.params[k].detach().numpy()
        reg = 0.0005 * w if (k.endswith("weights") and "/predict/" not in k) else 0.0
        want[k] = v / 3.0 + reg
    errs = grad_errors(net, want, c, b)
    assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])
    assert abs(float(net.regularization_loss()) - 0.5 * 0.0005 * sum(float((orc.params[k] ** 2).sum()) for k in want
                                                                     if k.endswith("weights") and "/predict/" not in k)) < 1e-4


def test_pw_feat_multiplyer():
    """cfg.gnet.pw_feat_multiplyer scales every _geometry_feats column (network.py:199-200)."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet
    c, b = 80, 2
    experiment_cfg()
    cfg.gnet.num_blocks = b
    cfg.gnet.pw_feat_multiplyer = 2.5
    params = go.init_params(c, b)
    net = Gnet(c)
    net.keep_edge_activations = True
    net.load_params(params)
    orc = go.GnetOracle(c, b, params=params, pw_feat_multiplyer=2.5)
    batch = make_image(120, c, seed=4)
    ref = orc.forward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    check_outputs(net, ref)
    assert rel_err(net.pw_feats.cpu().numpy(), ref["pw_feats"].detach().numpy()) < 1e-5
    pinned = pinned_errors(net, orc, batch, c, b)
    assert max(pinned.values()) <= PINNED
    experiment_cfg()


def test_exact_ties_from_duplicate_detections():
    """Duplicate detections give bit-identical edge activations: the segment max then has exact positive
    ties and TF splits the gradient evenly among them.  Exercises the tie counting of the
    streaming (max, count) combine, including its deferred tie repair across waves."""
    net, orc = make_pair(80, 2, bias=0.5)
    net.keep_edge_activations = True
    for seed in range(3):
        base = make_image(60, 80, seed=seed)
        rep = np.repeat(np.arange(60), 3)                       # every detection three times
        rng = np.random.default_rng(seed)
        rng.shuffle(rep)
        batch = dict(base)
        for k in ("dets", "det_scores", "det_classes"):
            batch[k] = base[k][rep]
        ref, gref = orc.forward_backward(batch, keep=True)
        net.run(batch)
        torch.cuda.synchronize()
        pm = net.debug_view("blk_pm", 180 * 64, dtype=torch.int64, index=1).cpu().numpy()
        assert (((pm & 0xffffffff) > 1) & ((pm >> 32) > 0)).any(), "test must contain positive ties"
        assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < 1e-5
        for blk in (1, 2):                                  # tie records vs the values the kernel saw: exact
            winner_records_exact(net, blk)
        n_diff, worst, where = kink_report(net, ref)        # ... and vs the oracle's own winner sets
        assert worst <= KINK, (seed, n_diff, worst, where)
        pinned = pinned_errors(net, orc, batch, 80, 2)
        assert max(pinned.values()) <= PINNED, (seed, max(pinned.items(), key=lambda kv: kv[1]))


def test_batch_with_empty_image_and_image_without_gt():
    """Ragged batch: an image with zero detections, one with zero GT, one ordinary (edge cases of the
    block-diagonal batch; the reference feeds one image at a time and skips images without detections,
    train.py:141-142)."""
    net, orc = make_pair(80, 2)
    net.keep_edge_activations = True
    empty = {"dets": np.zeros((0, 4), np.float32), "det_scores": np.zeros(0, np.float32),
             "det_classes": np.zeros(0, np.int32), "gt_boxes": np.zeros((0, 4), np.float32),
             "gt_crowd": np.zeros(0, bool), "gt_classes": np.zeros(0, np.int32)}
    nogt = make_image(40, 80, seed=3)
    nogt["gt_boxes"] = np.zeros((0, 4), np.float32); nogt["gt_crowd"] = np.zeros(0, bool)
    nogt["gt_classes"] = np.zeros(0, np.int32)
    normal = make_image(70, 80, seed=4)
    net.run([empty, nogt, normal])
    torch.cuda.synchronize()
    losses = net.image_losses[:, 0].cpu().numpy()
    assert losses[0] == 0.0
    r1, g1 = orc.forward_backward(nogt, pins=gpu_pins(net, image=1))
    r2, g2 = orc.forward_backward(normal, pins=gpu_pins(net, image=2))
    assert abs(losses[1] - float(r1["loss"])) < 1e-4 and abs(losses[2] - float(r2["loss"])) < 1e-4
    assert np.array_equal(net.det_gt_matching.cpu().numpy()[:40], np.full(40, -1))
    assert np.array_equal(net.det_gt_matching.cpu().numpy()[40:], r2["det_gt_matching"])
    gsum = {k: g1[k] + g2[k] for k in g1}
    assert max(grad_errors(net, gsum, 80, 2).values()) <= PINNED
    # a batch of only an empty image
    net.run([empty])
    assert float(net.grads.abs().max()) == 0.0 and net.num_edges == 0


def test_gradients_bitwise_reproducible():
    """No float atomics and a fixed summation order everywhere (static chunk assignment of the sparse edge
    stage, per-workgroup partials summed in index order): repeated runs give bit-identical outputs."""
    net, _ = make_pair(80, 4)
    batch = [make_image(500, 80, seed=21), make_image(120, 80, seed=22)]
    runs = []
    for _ in range(3):
        net.run(batch)
        torch.cuda.synchronize()
        runs.append((net.grads.clone(), net.prediction.clone(), net.image_losses.clone()))
    for g, p, l in runs[1:]:
        assert torch.equal(g, runs[0][0]) and torch.equal(p, runs[0][1]) and torch.equal(l, runs[0][2])


def test_alternating_batches_without_host_syncs_match_isolated_runs():
    """A step uses two streams (neighbour count, matching candidates, reverse-edge permutation and winner lists on
    the side stream) and recycles one workspace: steps of different shapes issued back to back, with no host
    synchronisation between them, must give bit for bit what each batch gives on a fresh network."""
    from gossipnet_amd.network import Gnet
    net, _ = make_pair(80, 4)
    batches = [[make_image(700, 80, seed=31), make_image(90, 80, seed=32)],
               [make_image(260, 80, seed=33)],
               [make_image(40, 80, seed=34), make_image(900, 80, seed=35), make_image(300, 80, seed=36)]]
    want = []
    for b in batches:
        ref = Gnet(80)
        ref.params.copy_(net.params)
        ref.run(b)
        torch.cuda.synchronize()
        want.append((ref.grads.clone(), ref.prediction.clone(), ref.det_gt_matching.clone()))
        del ref
    got = []
    for i in [0, 1, 2, 1, 0, 2, 2, 0]:
        net.run(batches[i])                       # no synchronize: the next step's launches queue behind this one
        got.append((i, net.grads.clone(), net.prediction.clone(), net.det_gt_matching.clone()))
    torch.cuda.synchronize()
    for i, g, p, m in got:
        assert torch.equal(g, want[i][0]) and torch.equal(p, want[i][1]) and torch.equal(m, want[i][2]), i


@pytest.mark.parametrize("imfeat_dim", [-1, 64])
def test_imfeats_start_features(imfeat_dim):
    """Image-feature variant (network.py:223-240): block_feats[0] = reduce_imfeats(flatten(crop_windows(imfeats, dets)))
    from a caller-supplied feature map; forward <= 1e-5, gradients of EVERY tensor (both reduce_imfeats FCs included)
    <= 1e-5 on the pinned piece."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet
    c, b, ch = 80, 2, 32
    imf = {"channels": ch, "imfeat_dim": imfeat_dim, "crop": 7, "stride": 16}
    experiment_cfg()
    cfg.gnet.num_blocks = b
    cfg.gnet.imfeats = True
    cfg.gnet.imfeat_dim = imfeat_dim
    params = go.init_params(c, b, imfeat=imf)
    net = Gnet(c, imfeat_channels=ch, imfeat_stride=16)
    net.keep_edge_activations = True
    net.load_params(params)
    orc = go.GnetOracle(c, b, params=params, imfeat=imf)
    rng = np.random.default_rng(0)
    batch = make_image(90, c, seed=7)
    batch["imfeats"] = rng.normal(size=(1, 30, 40, ch)).astype(np.float32)
    ref = orc.forward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    assert np.array_equal(net.roifeats.cpu().numpy(), ref["roifeats"])
    check_outputs(net, ref)
    assert rel_err(net.block_feats[0].cpu().numpy(), ref["block_feats"][0].detach().numpy()) < 1e-5
    _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
    errs = grad_errors(net, gpin, c, b, imfeat=imf)
    assert any(k.startswith("gnet/reduce_imfeats/") for k in errs)
    assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])
    experiment_cfg()


def test_imfeats_rank_without_images_contributes_zero():
    """The image-feature variant on a data-parallel rank whose shard is empty: the placeholder image has no feature map; the
    step runs, the loss is 0 and every gradient -- the reduce_imfeats tensors included, right after a step that left them
    non-zero -- is exactly 0 (the other ranks would otherwise wait in the all-reduce for a rank that raised)."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet
    c, b, ch = 80, 2, 32
    experiment_cfg()
    cfg.gnet.num_blocks = b
    cfg.gnet.imfeats = True
    cfg.gnet.imfeat_dim = 64
    net = Gnet(c, imfeat_channels=ch, imfeat_stride=16)
    batch = make_image(60, c, seed=3)
    batch["imfeats"] = np.random.default_rng(1).normal(size=(1, 30, 40, ch)).astype(np.float32)
    net.run(batch); torch.cuda.synchronize()
    g_full = net.grads.clone()
    assert net.gradients["gnet/reduce_imfeats/fully_connected/weights"].abs().max().item() > 0
    net.run([]); torch.cuda.synchronize()
    assert float(net.loss) == 0.0 and net.grads.abs().max().item() == 0.0
    net.run(batch); torch.cuda.synchronize()
    assert torch.equal(net.grads, g_full)
    experiment_cfg()


def test_neighbor_feats():
    """cfg.gnet.neighbor_feats=True (network.py:356-365): the neighbour half of build_context comes from a second reduce
    FC `reduce_dim_neighbor`; forward <= 1e-5, gradients of every tensor <= 1e-5 on the pinned piece."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet
    c, b = 80, 3
    experiment_cfg()
    cfg.gnet.num_blocks = b
    cfg.gnet.neighbor_feats = True
    params = go.init_params(c, b, neighbor_feats=True)
    net = Gnet(c)
    net.keep_edge_activations = True
    net.load_params(params)
    orc = go.GnetOracle(c, b, params=params, neighbor_feats=True)
    for seed in (0, 1):
        batch = make_image(150, c, seed=seed)
        ref = orc.forward(batch)
        net.run(batch)
        torch.cuda.synchronize()
        check_outputs(net, ref)
        for k in range(1, b + 1):
            assert rel_err(net.block_feats[k].cpu().numpy(), ref["block_feats"][k].detach().numpy()) < 1e-5
        _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
        errs = grad_errors(net, gpin, c, b, neighbor_feats=True)
        assert any("reduce_dim_neighbor" in k for k in errs)
        assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])
    # inference mode agrees with training mode
    p1 = net.prediction.cpu().numpy().copy()
    net.run({k: batch[k] for k in ("dets", "det_scores", "det_classes")})
    assert np.array_equal(net.prediction.cpu().numpy(), p1)
    experiment_cfg()
