"""CPU ORACLE for the Gnet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path (gossipnet_amd/) never does: it fails
loudly when the HIP library is missing.

PARITY UNPINNED: the reference ships no golden vectors or tests for this path
 and its arithmetic runs inside un-vendored TensorFlow 0.12
(not installable here; no network).  This file restates the reference graph
op-for-op; it is pinned only by the hand-derived known-answer tests of
SURVEY.md §8c (tests/test_oracle_kat.py) and by an fp64 twin of itself.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Geometry is numpy fp32 executed op by op in the reference's
order (no FMA contraction, IEEE division) so that neighbour indices are
bit-exact; the learned layers are torch-CPU so that autograd provides the
backward oracle, with TF's SegmentMax tie-splitting gradient restated by hand.
"""
import math
import os
import numpy as np
import torch

# ---- hyper-parameters: nms_net/config.py:58-79 overridden by experiments/*/conf.yaml
SHORTCUT_DIM = 128       # config.py:59
REDUCED_DIM = 32         # config.py:61
PAIRFEAT_DIM = 64        # config.py:62
PWFEAT_DIM = 256         # config.py:74
PWFEAT_NARROW_DIM = 32   # experiments/*/conf.yaml pwfeat_narrow_dim
NUM_PWFEAT_FC = 3        # experiments/*/conf.yaml num_pwfeat_fc
PREDICT_FC_DIM = 128     # config.py:68
NEIGHBOR_THRESH = 0.2    # config.py:58


def pw_feat_dim(num_classes):
    """network.py:452-453: 2*C' score columns + 7 geometry columns."""
    cp = num_classes if num_classes > 1 else 1
    return 2 * cp + 7


def imfeat_param_spec(imfeat):
    """reduce_imfeats FCs (network.py:223-240).  imfeat = dict(channels, imfeat_dim, crop (7), stride (16)).  TF names
    the first fully_connected of the scope `fully_connected`, a second one `fully_connected_1`."""
    d_in = imfeat.get("crop", 7) ** 2 * imfeat["channels"]
    spec, names = [], ["gnet/reduce_imfeats/fully_connected/", "gnet/reduce_imfeats/fully_connected_1/"]
    if imfeat.get("imfeat_dim", -1) > 0:
        spec += [(names[0] + "weights", (d_in, imfeat["imfeat_dim"])), (names[0] + "biases", (imfeat["imfeat_dim"],))]
        d_in = imfeat["imfeat_dim"]
        names = names[1:]
    spec += [(names[0] + "weights", (d_in, SHORTCUT_DIM)), (names[0] + "biases", (SHORTCUT_DIM,))]
    return spec


def pw_mlp_dims(num_classes, num_pwfeat_fc, narrow=PWFEAT_NARROW_DIM):
    """_pw_feats_fc (network.py:324-342, called only when num_pwfeat_fc > 0, :217-221): num_pwfeat_fc - 1 layers of
    pwfeat_dim, then one of pwfeat_narrow_dim.  With num_pwfeat_fc = 0 (the reference's default, config.py:73) the list is
    just the raw feature width: the blocks' pw_fc1 reads the 2C'+7 columns of _geometry_feats."""
    d = pw_feat_dim(num_classes)
    return [d] + [PWFEAT_DIM] * max(num_pwfeat_fc - 1, 0) + ([narrow] if num_pwfeat_fc > 0 else [])


def param_spec(num_classes, num_blocks, imfeat=None, neighbor_feats=False, num_pwfeat_fc=NUM_PWFEAT_FC):
    """Ordered (TF variable name, shape) list; FC weights are [in, out]
    (tf.contrib.layers.fully_connected).  Scopes: network.py:167,218,260,267,
    334,341,347,354,385,397,405; SURVEY.md §8f.  imfeat: the reduce_imfeats variables follow."""
    if imfeat is not None:
        return param_spec(num_classes, num_blocks, None, neighbor_feats, num_pwfeat_fc) + imfeat_param_spec(imfeat)
    spec = []
    dims = pw_mlp_dims(num_classes, num_pwfeat_fc)
    for i in range(num_pwfeat_fc):
        spec.append(("gnet/pw_feats/fc%d/weights" % (i + 1), (dims[i], dims[i + 1])))
        spec.append(("gnet/pw_feats/fc%d/biases" % (i + 1), (dims[i + 1],)))
    for b in range(1, num_blocks + 1):
        p = "gnet/block%d/" % b
        spec += [
            (p + "reduce_dim/weights", (SHORTCUT_DIM, REDUCED_DIM)), (p + "reduce_dim/biases", (REDUCED_DIM,)),
            (p + "pw_fc1/weights", (dims[-1] + 2 * REDUCED_DIM, PAIRFEAT_DIM)), (p + "pw_fc1/biases", (PAIRFEAT_DIM,)),
            (p + "pw_fc2/weights", (PAIRFEAT_DIM, PAIRFEAT_DIM)), (p + "pw_fc2/biases", (PAIRFEAT_DIM,)),
            (p + "fc1/weights", (PAIRFEAT_DIM, PAIRFEAT_DIM)), (p + "fc1/biases", (PAIRFEAT_DIM,)),
            (p + "fc2/weights", (PAIRFEAT_DIM, SHORTCUT_DIM)), (p + "fc2/biases", (SHORTCUT_DIM,)),
        ]
        if neighbor_feats:   # network.py:356-365
            spec += [(p + "reduce_dim_neighbor/weights", (SHORTCUT_DIM, REDUCED_DIM)), (p + "reduce_dim_neighbor/biases", (REDUCED_DIM,))]
    for i in (1, 2):
        p = "gnet/predict/fc%d/fully_connected/" % i
        spec += [(p + "weights", (PREDICT_FC_DIM, PREDICT_FC_DIM)), (p + "biases", (PREDICT_FC_DIM,))]
    p = "gnet/predict/logits/fully_connected/"
    spec += [(p + "weights", (PREDICT_FC_DIM, 1)), (p + "biases", (1,))]
    return spec


def init_params(num_classes, num_blocks, seed=42, bias_init=0.01, imfeat=None, neighbor_feats=False, num_pwfeat_fc=NUM_PWFEAT_FC):
    """xavier-uniform weights (network.py:203-205, limit sqrt(6/(fan_in+fan_out))),
    constant biases (network.py:215).  TF's RNG stream cannot be reproduced; the
    seed only fixes OUR stream."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_spec(num_classes, num_blocks, imfeat, neighbor_feats, num_pwfeat_fc):
        if name.endswith("weights"):
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(torch.float32).numpy()
        else:
            out[name] = np.full(shape, bias_init, dtype=np.float32)
    return out


# ------------------------------------------------------------------ geometry
def xyxy_to_boxdata(a):
    """network.py:463-472 -> (x1, y1, w, h, x2, y2, area) as [N,1] columns."""
    a = np.asarray(a)
    x1, y1, x2, y2 = a[:, 0:1], a[:, 1:2], a[:, 2:3], a[:, 3:4]
    w = x2 - x1
    h = y2 - y1
    return (x1, y1, w, h, x2, y2, w * h)


def intersection(a, b):
    """network.py:491-511."""
    x1 = np.maximum(a[0].reshape(-1, 1), b[0].reshape(1, -1))
    y1 = np.maximum(a[1].reshape(-1, 1), b[1].reshape(1, -1))
    x2 = np.minimum(a[4].reshape(-1, 1), b[4].reshape(1, -1))
    y2 = np.minimum(a[5].reshape(-1, 1), b[5].reshape(1, -1))
    zero = a[0].dtype.type(0.0)
    w = np.maximum(zero, x2 - x1)
    h = np.maximum(zero, y2 - y1)
    return w * h


def iou(a, b, crowd=None):
    """network.py:475-488: inter / ((a_area + b_area) - inter); crowd columns use
    inter / a_area (a = detections)."""
    a_area = a[6].reshape(-1, 1)
    b_area = b[6].reshape(1, -1)
    inter = intersection(a, b)
    with np.errstate(divide="ignore", invalid="ignore"):
        union = (a_area + b_area) - inter
        res = inter / union
        if crowd is None:
            return res
        ioa = inter / a_area
    crowd = np.asarray(crowd, dtype=bool).reshape(1, -1)
    return np.where(np.broadcast_to(crowd, res.shape), ioa, res)


def preprocess(dets, det_classes, gt_boxes, gt_crowd, gt_classes, num_classes,
               thresh=NEIGHBOR_THRESH, dtype=np.float32):
    """network.py:168-195: boxdata, det_anno_iou (class-masked when multiclass,
    :177-187), det_det_iou, neighbour pairs = where(iou >= thresh) row-major."""
    dets = np.asarray(dets, dtype=dtype)
    gt_boxes = np.asarray(gt_boxes, dtype=dtype).reshape(-1, 4)
    db = xyxy_to_boxdata(dets)
    gb = xyxy_to_boxdata(gt_boxes)
    det_anno_iou = iou(db, gb, gt_crowd)
    det_det_iou = iou(db, db)
    if num_classes > 1:
        same = np.asarray(det_classes).reshape(-1, 1) == np.asarray(gt_classes).reshape(1, -1)
        det_anno_iou = np.where(same, det_anno_iou, dtype(0.0))
    # the threshold is a python float converted to the tensor dtype (network.py:192-193)
    pairs = np.argwhere(det_det_iou >= dtype(thresh))  # row-major: c ascending, n ascending
    return db, det_anno_iou.astype(dtype), det_det_iou, pairs.astype(np.int64)


def geometry_feats(db, det_det_iou, det_scores, det_classes, pairs, num_classes, dtype=np.float32):
    """network.py:411-454, columns [c_score(C'), n_score(C'), iou, x_dist, y_dist,
    l2_dist, w_diff, h_diff, aspect_diff]."""
    c, n = pairs[:, 0], pairs[:, 1]
    N = db[0].shape[0]
    det_scores = np.asarray(det_scores, dtype=dtype)
    if num_classes > 1:
        sc = np.zeros((N, num_classes), dtype=dtype)          # scatter_nd :413-419
        sc[np.arange(N), np.asarray(det_classes) - 1] = det_scores
    else:
        sc = det_scores.reshape(-1, 1)
    c_score, n_score = sc[c], sc[n]
    ious = det_det_iou[c, n].reshape(-1, 1)
    x1, y1, w, h = db[0], db[1], db[2], db[3]
    two = dtype(2.0)
    c_w, c_h = w[c], h[c]
    c_scale = (c_w + c_h) / two
    c_cx = x1[c] + c_w / two
    c_cy = y1[c] + c_h / two
    n_w, n_h = w[n], h[n]
    n_cx = x1[n] + n_w / two
    n_cy = y1[n] + n_h / two
    x_dist = n_cx - c_cx
    y_dist = n_cy - c_cy
    l2_dist = np.sqrt(x_dist * x_dist + y_dist * y_dist) / c_scale
    x_dist = x_dist / c_scale
    y_dist = y_dist / c_scale
    log2 = dtype(np.log(2.0))
    w_diff = np.log(n_w / c_w) / log2
    h_diff = np.log(n_h / c_h) / log2
    aspect_diff = (np.log(n_w / n_h) - np.log(c_w / c_h)) / log2
    return np.concatenate([c_score, n_score, ious, x_dist, y_dist, l2_dist,
                           w_diff, h_diff, aspect_diff], axis=1).astype(dtype)


# ------------------------------------------------------------------ learned layers
class _SegmentMax(torch.autograd.Function):
    """tf.segment_max over sorted ids (network.py:387-388) with TF's gradient
    (_SegmentMinOrMaxGrad): sel = (x == out[ids]); cnt = segment_sum(sel);
    dx = sel ? (dout / cnt)[ids] : 0 -- ties share the gradient evenly.
    A segment id without rows yields 0."""

    @staticmethod
    def forward(ctx, x, ids, n_seg):
        out = torch.zeros(n_seg, x.shape[1], dtype=x.dtype)
        idx = ids.view(-1, 1).expand_as(x)
        out = out.scatter_reduce(0, idx, x, reduce="amax", include_self=False)
        ctx.save_for_backward(x, ids, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, ids, out = ctx.saved_tensors
        sel = x == out[ids]
        cnt = torch.zeros_like(out).index_add_(0, ids, sel.to(x.dtype))
        weighted = g / cnt
        return torch.where(sel, weighted[ids], torch.zeros_like(x)), None, None


class _PinnedRelu(torch.autograd.Function):
    """relu whose BACKWARD uses a caller-supplied mask instead of (out > 0).  Test device: the loss is
    piecewise smooth, and two fp32 implementations whose pre-activations differ by 1e-7 can sit on different
    sides of a ReLU kink.  Feeding the masks of the implementation under test makes both sides differentiate
    the SAME smooth piece (TF's ReluGrad is g * (out > 0); the mask is that predicate evaluated on the other
    implementation's `out`)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


class _PinnedSegmentMaxRelu(torch.autograd.Function):
    """relu + tf.segment_max (network.py:385-388) whose BACKWARD routes the gradient to the caller-supplied
    winner set `sel` [E, F] (True where the edge attains a POSITIVE segment maximum; ties all True) with TF's
    even split dx = sel ? (dout / segment_sum(sel))[ids] : 0.  Forward values are this oracle's own."""

    @staticmethod
    def forward(ctx, x, ids, n_seg, sel):
        h = torch.relu(x)
        out = torch.zeros(n_seg, x.shape[1], dtype=x.dtype)
        idx = ids.view(-1, 1).expand_as(h)
        out = out.scatter_reduce(0, idx, h, reduce="amax", include_self=False)
        ctx.save_for_backward(ids, sel)
        ctx.n_seg = n_seg
        return out

    @staticmethod
    def backward(ctx, g):
        ids, sel = ctx.saved_tensors
        self_f = sel.to(g.dtype)
        cnt = torch.zeros(ctx.n_seg, sel.shape[1], dtype=g.dtype).index_add_(0, ids, self_f)
        weighted = g / torch.clamp(cnt, min=1.0)
        return weighted[ids] * self_f, None, None, None


def _fc(x, params, scope, relu, stats=None, pin=None, pre=None):
    y = x @ params[scope + "/weights"] + params[scope + "/biases"]
    if pre is not None:
        pre.append(y.detach().numpy())
    if relu and stats is not None and y.numel():
        stats["relu_margin"] = min(stats.get("relu_margin", float("inf")), float(y.detach().abs().min()))
    if relu and pin is not None:
        return _PinnedRelu.apply(y, pin)
    return torch.relu(y) if relu else y


def _segmax_gap(h, ids, n_seg):
    """Smallest positive gap between the largest and the second largest value of a segment
    (conditioning of the arg-max; exact ties are handled identically by every implementation)."""
    h = h.detach()
    idx = ids.view(-1, 1).expand_as(h)
    top = torch.zeros(n_seg, h.shape[1], dtype=h.dtype).scatter_reduce(0, idx, h, reduce="amax", include_self=False)
    rest = torch.where(h == top[ids], torch.full_like(h, -1.0), h)
    second = torch.full((n_seg, h.shape[1]), -1.0, dtype=h.dtype).scatter_reduce(0, idx, rest, reduce="amax", include_self=True)
    gap = (top - second)[(top > 0) & (second >= 0)]
    return float(gap.min()) if gap.numel() else float("inf")


def sigmoid_xent(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits (network.py:301-302):
    max(x,0) - x*z + log(1 + exp(-|x|)), written with select(x >= 0, ...) as TF
    does, so that autograd yields w*(sigmoid(x) - z) also at x == 0."""
    zeros = torch.zeros_like(x)
    cond = x >= zeros
    relu_x = torch.where(cond, x, zeros)
    neg_abs = torch.where(cond, -x, x)
    return relu_x - x * z + torch.log(1.0 + torch.exp(neg_abs))


def detection_matching_py(ious, score, ignore):
    """matching_module/det_matching.cc:95-159.  Order: score descending, ties ->
    higher index first (a stable ascending sort reversed);
    GT order: non-crowd first, index ascending inside each group."""
    ious = np.asarray(ious)
    n_det, n_gt = ious.shape[0], len(ignore)
    score = np.asarray(score)
    det_order = np.argsort(score, kind="stable")[::-1]
    gt_order = np.argsort(np.asarray(ignore, dtype=np.int8), kind="stable")
    labels = np.zeros(n_det, np.float32)
    weights = np.ones(n_det, np.float32)
    assign = np.full(n_det, -1, np.int32)
    matched = np.zeros(n_gt, bool)
    half = ious.dtype.type(0.5)
    for det in det_order:
        best, match = half, -1
        for gt in gt_order:
            if matched[gt] and not ignore[gt]:
                continue
            if match > -1 and ignore[gt]:
                break
            if ious[det, gt] < best:
                continue
            best, match = ious[det, gt], gt
        if match > -1:
            matched[match] = True
            labels[det] = 1
            assign[det] = match
            if ignore[match]:
                weights[det] = 0
    return labels, weights, assign


class GnetOracle:
    """Forward + loss + autograd backward of network.py:167-314 on CPU."""

    def __init__(self, num_classes, num_blocks=16, params=None, class_weights=None,
                 dtype=torch.float32, thresh=NEIGHBOR_THRESH, normalize_loss=False,
                 loss_multiplyer=1.0, bias_init=0.01, matching_fn=None, pw_feat_multiplyer=1.0, imfeat=None,
                 neighbor_feats=False, num_pwfeat_fc=NUM_PWFEAT_FC):
        self.num_classes = num_classes
        self.num_blocks = num_blocks
        self.dtype = dtype
        self.npdtype = np.float32 if dtype == torch.float32 else np.float64
        self.thresh = thresh
        self.normalize_loss = normalize_loss
        self.loss_multiplyer = loss_multiplyer
        self.pw_feat_multiplyer = pw_feat_multiplyer    # config.py:77, network.py:199-200
        self.imfeat = imfeat                            # image-feature variant (network.py:223-240), see imfeat_param_spec
        self.neighbor_feats = neighbor_feats            # config.py:72, network.py:356-365
        self.num_pwfeat_fc = num_pwfeat_fc              # config.py:73, network.py:217-221 (0: no pairwise-feature MLP)
        if params is None:
            params = init_params(num_classes, num_blocks, bias_init=bias_init, imfeat=imfeat, neighbor_feats=neighbor_feats,
                                 num_pwfeat_fc=num_pwfeat_fc)
        self.params = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)
                       for k, v in params.items()}
        if class_weights is None:
            class_weights = np.ones(num_classes + 1, np.float32)   # network.py:282-284
        self.class_weights = torch.tensor(np.asarray(class_weights), dtype=dtype)
        self.matching_fn = matching_fn or detection_matching_py

    def forward(self, batch, with_loss=True, keep=False, stats=None, pins=None):
        """stats (optional dict) collects the conditioning of the non-smooth points:
        relu_margin = min |pre-activation| over every ReLU, max_gap = min top-2 gap of segment_max.
        pins (optional dict, tests only) fixes the smooth piece the BACKWARD pass differentiates -- the ReLU
        masks and segment-max winner sets of another implementation's forward pass (bool arrays):
          "pw": [3] masks of pw_feats/fc1..3;  "r"/"h1"/"sel"/"q"/"x": [num_blocks] masks of reduce_dim,
          pw_fc1, the segment-max winners (of relu(pw_fc2)), fc1 and the block output.
        Forward values are unaffected."""
        P = self.params
        npd = self.npdtype
        C = self.num_classes
        dets = np.asarray(batch["dets"], dtype=npd)
        N = dets.shape[0]
        gt_boxes = np.asarray(batch.get("gt_boxes", np.zeros((0, 4))), dtype=npd).reshape(-1, 4)
        gt_crowd = np.asarray(batch.get("gt_crowd", np.zeros(0, bool)), dtype=bool)
        gt_classes = np.asarray(batch.get("gt_classes", np.zeros(0, np.int32)), dtype=np.int32)
        db, det_anno_iou, det_det_iou, pairs = preprocess(
            dets, batch["det_classes"], gt_boxes, gt_crowd, gt_classes, C, self.thresh, npd)
        raw = geometry_feats(db, det_det_iou, batch["det_scores"], batch["det_classes"], pairs, C, npd)
        raw = raw * npd(self.pw_feat_multiplyer)             # network.py:199-200
        out = {"det_anno_iou": det_anno_iou, "neighbor_pair_idxs": pairs, "raw_pw_feats": raw,
               "num_dets": N}
        if keep:
            out["det_det_iou"] = det_det_iou
        c_idx = torch.from_numpy(pairs[:, 0].copy())
        n_idx = torch.from_numpy(pairs[:, 1].copy())
        f = torch.from_numpy(raw).to(self.dtype)          # stop_gradient (:454)
        # _pw_feats_fc network.py:324-342
        pin = (lambda key, i: None) if pins is None else (lambda key, i: torch.as_tensor(np.asarray(pins[key][i])))
        own = {"pw": [], "r": [], "rn": [], "h1": [], "sel": [], "q": [], "x": [], "im": []}   # this forward's own smooth piece (keep=True)
        # ... and the pre-activations behind every mask of it (keep=True): how far each unit is from its kink.  "sel"
        # holds the pw_fc2 pre-activations (the segment maximum is taken on their rectified values).
        pre = {k_: [] for k_ in own}
        rec = (lambda key: pre[key]) if (keep and pins is None) else (lambda key: None)
        note = (lambda key, t: own[key].append((t.detach() > 0).numpy())) if keep else (lambda key, t: None)
        note_im = lambda t: note("im", t)
        for i in range(1, self.num_pwfeat_fc + 1):       # network.py:217-221: only when num_pwfeat_fc > 0
            f = _fc(f, P, "gnet/pw_feats/fc%d" % i, True, stats, pin("pw", i - 1), rec("pw"))
            note("pw", f)
        pw = f
        out["pw_feats"] = pw
        if self.imfeat is None:
            x = torch.zeros(N, SHORTCUT_DIM, dtype=self.dtype)   # network.py:241-246
        else:
            # network.py:223-240: crop_windows (enlarge_windows :78-86, to_frcn_coords :97-100, RoiPool 7x7 at 1/stride)
            # -> flatten -> fully_connected (ReLU) [-> fully_connected_1 (ReLU)]
            from . import native
            two = npd(2.0)
            x1, y1, w_, h_, x2, y2 = db[0], db[1], db[2], db[3], db[4], db[5]
            cx, cy = (x1 + x2) / two, (y1 + y2) / two
            nw2, nh2 = w_ * npd(1.0), h_ * npd(1.0)              # w * (0.5 + padding), padding = 0.5
            boxes = np.concatenate([np.zeros_like(cx), cx - nw2, cy - nh2, cx + nw2, cy + nh2], 1).astype(np.float32)
            fmap = np.asarray(batch["imfeats"], np.float32)
            fmap = fmap.reshape((1,) + fmap.shape[-3:])
            crop = self.imfeat.get("crop", 7)
            roifeats, _ = native.roi_pool(fmap, boxes, crop, crop, 1.0 / self.imfeat.get("stride", 16))
            out["roifeats"] = roifeats
            x = torch.from_numpy(roifeats.reshape(N, -1)).to(self.dtype)
            scopes = sorted({n_.rsplit("/", 1)[0] for n_ in P if n_.startswith("gnet/reduce_imfeats/")})
            for li, sc in enumerate(scopes):                    # fully_connected, fully_connected_1
                x = _fc(x, P, sc, True, stats, pin("im", li), rec("im"))
                note_im(x)
        block_feats = [x]
        is_id = (c_idx == n_idx).view(-1, 1)
        for b in range(1, self.num_blocks + 1):             # _block network.py:344-409
            s = "gnet/block%d/" % b
            r = _fc(x, P, s + "reduce_dim", True, stats if b > 1 else None, pin("r", b - 1), rec("r"))
            note("r", r)
            cf = r[c_idx]
            if self.neighbor_feats:                      # network.py:356-365: a second reduce FC for the neighbour side
                rnb = _fc(x, P, s + "reduce_dim_neighbor", True, stats if b > 1 else None, pin("rn", b - 1), rec("rn"))
                note("rn", rnb)
            else:
                rnb = r
            nf = torch.where(is_id, torch.zeros((), dtype=self.dtype), rnb[n_idx])
            h = torch.cat([pw, cf, nf], 1)
            h = _fc(h, P, s + "pw_fc1", True, stats, pin("h1", b - 1), rec("h1"))
            note("h1", h)
            if pins is None:
                h = _fc(h, P, s + "pw_fc2", True, stats, None, rec("sel"))
                if stats is not None:
                    stats["max_gap"] = min(stats.get("max_gap", float("inf")), _segmax_gap(h, c_idx, N))
                p = _SegmentMax.apply(h, c_idx, N)
                if keep:
                    own["sel"].append(((h == p[c_idx]) & (h > 0)).detach().numpy())
            else:
                h = _fc(h, P, s + "pw_fc2", False)
                p = _PinnedSegmentMaxRelu.apply(h, c_idx, N, pin("sel", b - 1))
            q = _fc(p, P, s + "fc1", True, stats, pin("q", b - 1), rec("q"))
            note("q", q)
            y = _fc(q, P, s + "fc2", False)
            if stats is not None and N:
                stats["relu_margin"] = min(stats.get("relu_margin", float("inf")), float((x + y).detach().abs().min()))
            if keep and pins is None:
                pre["x"].append((x + y).detach().numpy())
            x = torch.relu(x + y) if pins is None else _PinnedRelu.apply(x + y, pin("x", b - 1))
            note("x", x)
            block_feats.append(x)
        out["block_feats"] = block_feats
        if keep and pins is None:
            out["pins"] = own
            out["pre"] = pre
        h = x                                                 # head network.py:258-273
        h = _fc(h, P, "gnet/predict/fc1/fully_connected", False)
        h = _fc(h, P, "gnet/predict/fc2/fully_connected", False)
        pred = _fc(h, P, "gnet/predict/logits/fully_connected", False).reshape(-1)
        out["prediction"] = pred
        if not with_loss:
            return out
        # loss network.py:275-314
        labels, weights, assign = self.matching_fn(
            det_anno_iou.astype(np.float32), pred.detach().to(torch.float32).numpy(), gt_crowd)
        out["labels"], out["det_gt_matching"] = labels, assign
        if gt_crowd.shape[0] > 0:
            a0 = np.maximum(assign, 0)
            det_crowd = gt_crowd[a0]
            det_class = gt_classes[a0]
        else:
            det_crowd = np.zeros(N, bool)
            det_class = np.zeros(N, np.int32)
        det_class = np.where((assign >= 0) & ~det_crowd, det_class, 0)
        w = torch.from_numpy(weights).to(self.dtype) * self.class_weights[torch.from_numpy(det_class.astype(np.int64))]
        out["weights"] = w
        z = torch.from_numpy(labels).to(self.dtype)
        wl = sigmoid_xent(pred, z) * w
        out["loss_unnormed"] = wl.sum()
        out["loss_normed"] = wl.mean() if N > 0 else wl.sum()
        out["loss"] = (out["loss_normed"] if self.normalize_loss else out["loss_unnormed"]) * self.loss_multiplyer
        return out

    def forward_backward(self, batch, stats=None, pins=None, keep=False):
        for p in self.params.values():
            p.grad = None
        out = self.forward(batch, stats=stats, pins=pins, keep=keep)
        out["loss"].backward()
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy()
                 for k, p in self.params.items()}
        return out, grads


def flatten(params_or_grads, num_classes, num_blocks, imfeat=None, neighbor_feats=False, num_pwfeat_fc=NUM_PWFEAT_FC):
    """Flat fp32 vector in param_spec order (the layout of include/gossipnet_hip.h)."""
    return np.concatenate([np.asarray(params_or_grads[n], dtype=np.float32).reshape(-1)
                           for n, _ in param_spec(num_classes, num_blocks, imfeat, neighbor_feats, num_pwfeat_fc)])
