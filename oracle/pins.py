"""TEST INFRASTRUCTURE (like everything under oracle/): helpers that compare the HIP backward pass with the
oracle on the SAME smooth piece of the piecewise-smooth loss.  Imported by tests/ and __graft_entry__.smoke()."""
import numpy as np
import torch

from . import gnet_oracle as go


def grad_errors(net, gref, c, b, imfeat=None, neighbor_feats=False, num_pwfeat_fc=go.NUM_PWFEAT_FC):
    """Per-tensor max |g_hip - g_ref| / max |g_ref| (TF variable name -> error)."""
    errs = {}
    for name, shape in go.param_spec(c, b, imfeat, neighbor_feats, num_pwfeat_fc):
        g = net.gradients[name].detach().cpu().numpy().reshape(-1)       # (by name: the flat buffer may hold alignment padding)
        gr = np.asarray(gref[name], np.float64).reshape(-1)
        m = np.abs(gr).max() if gr.size else 0.0
        errs[name] = float(np.abs(g - gr).max() / m) if m > 0 else float(np.abs(g).max())
    return errs


# ---- the ONE exception to "pinned gradients <= 1e-5 of the fp32 oracle's"  ---------------------------------------
# A step of at most TINY_STEP_DETS detections can have parameter tensors whose LARGEST gradient entry is itself a cancelled sum
# (two or three overlapping detections: a head bias gradient is +0.2542 - 0.2516): one ulp of a summand is 1e-5 of the result, and
# the fp32 oracle is then no better a yardstick than the device.  For such a step -- and only there -- a tensor above the bar is
# settled against the oracle's fp64 twin on the same pinned piece:
#     device's error against fp64  <=  the fp32 oracle's own error against fp64  +  PINNED (1e-5),
# both measured like the bar itself (max |difference| / (max |fp64 gradient| + floor)).  Everything larger keeps the hard bar.
TINY_STEP_DETS = 4


def fp64_rule(g_dev, g_f32, g_f64, n_dets, bar=1e-5, floor=0.0):
    """(ok, e_dev, e_f32) for one parameter tensor of a step of n_dets detections; ok is False for any step above
    TINY_STEP_DETS detections whatever the errors are."""
    g_dev, g_f32, g_f64 = (np.asarray(x, np.float64).reshape(-1) for x in (g_dev, g_f32, g_f64))
    den = np.abs(g_f64).max() + floor if g_f64.size else 1.0
    if den == 0.0:
        den = 1.0
    e_dev, e_f32 = float(np.abs(g_dev - g_f64).max() / den), float(np.abs(g_f32 - g_f64).max() / den)
    return (n_dets <= TINY_STEP_DETS and e_dev <= e_f32 + bar), e_dev, e_f32


def device_winner_sets(net, b):
    """[E,64] bool (device tensor): the winner set the HIP backward routes block b's SegmentMax gradient through -- the
    recorded arg-max edge of every (detection, column) with a positive maximum plus, on detections flagged as tied, the
    extra winners of the tied columns (csrc/backward_edge.hip winners_mark / winners_ties)."""
    E, N, B = int(net.num_edges), int(net.num_dets), net.num_blocks
    dv = net.debug_view
    wl_stride = ((E + 64 + 63) // 64) * 64            # edge_geom(): xm_stride = wl_stride, tf_stride
    tf_stride = ((N + 32 + 63) // 64) * 64
    xmask = dv("xmask", B * wl_stride, dtype=torch.int64).view(B, wl_stride)
    tflag = dv("tflag", B * tf_stride, dtype=torch.uint8).view(B, tf_stride)
    edge_c = net.neighbor_pair_idxs[:, 0]
    shifts = torch.arange(64, device=xmask.device, dtype=torch.int64).view(1, 64)
    cols = torch.arange(64, device=xmask.device).view(1, 64).expand(N, 64)
    parg = dv("blk_parg", N * 64, dtype=torch.int64, index=b).view(N, 64)
    valid = (parg >> 32) != 0
    sel = torch.zeros(E, 64, dtype=torch.bool, device=parg.device)
    sel[(parg & 0xffffffff)[valid], cols[valid]] = True
    tied_rows = tflag[b - 1, :N][edge_c] != 0                    # edges of flagged detections
    extra = ((xmask[b - 1, :E].view(-1, 1) >> shifts) & 1) != 0
    return sel | (extra & tied_rows.view(-1, 1))


def gpu_pins(net, image=None):
    """The smooth piece the HIP backward differentiated: ReLU masks (out > 0 on the HIP forward's own
    activations, TF ReluGrad) and the segment-max winner sets (device_winner_sets, ties included) -- in the
    layout GnetOracle.forward(pins=...) takes.  The Gnet must have been run with
    net.keep_edge_activations = True (block pw_fc1 activations kept in HBM).  image = index inside a
    multi-image batch (rows / edges of that image only)."""
    E, N, B = int(net.num_edges), int(net.num_dets), net.num_blocks
    dv = net.debug_view
    d0, d1, e0, e1 = 0, N, 0, E
    if image is not None:
        db = net._dbatch
        d0, d1 = int(db.det_off_h[image]), int(db.det_off_h[image + 1])
        rp = net.row_ptr.cpu().numpy()
        e0, e1 = int(rp[d0]), int(rp[d1])
    cpu = lambda t: t.cpu().numpy()
    pins = {"pw": [], "r": [], "rn": [], "h1": [], "sel": [], "q": [], "x": [], "im": []}
    if net._buf.pw_h1:          # (num_pwfeat_fc = 0: no pairwise-feature MLP, no masks of it)
        pins["pw"] = [cpu(dv("pw_h1", E * 256).view(E, 256)[e0:e1] > 0), cpu(dv("pw_h2", E * 256).view(E, 256)[e0:e1] > 0),
                      cpu(net.pw_feats[e0:e1] > 0)]
    if getattr(net, "_imfeats", False):
        pins["im"] = [cpu(a_[d0:d1] > 0) for a_ in net._imfeat_acts]
    bf = net.block_feats
    for b in range(1, B + 1):
        pins["r"].append(cpu(dv("blk_r", N * 32, index=b).view(N, 32)[d0:d1] > 0))
        if net._buf.blk_rnb[b]:
            pins["rn"].append(cpu(dv("blk_rnb", N * 32, index=b).view(N, 32)[d0:d1] > 0))
        pins["h1"].append(cpu(dv("blk_h1", E * 64, index=b).view(E, 64)[e0:e1] > 0))
        pins["sel"].append(cpu(device_winner_sets(net, b)[e0:e1]))
        pins["q"].append(cpu(dv("blk_q", N * 64, index=b).view(N, 64)[d0:d1] > 0))
        pins["x"].append(cpu(bf[b][d0:d1] > 0))
    return pins


def mask_disagreements(dev, out):
    """Where do the HIP forward pass and the oracle's OWN forward pass sit on different sides of a kink, and how close to
    the kink is the oracle there?  dev = gpu_pins(net[, image]); out = GnetOracle.forward(batch, keep=True) of the same
    image (its own masks out["pins"] and the pre-activations behind them out["pre"]).

    Returns (n_diff, worst, where): the number of mask entries that differ, and the largest distance from the kink --
    measured in the ORACLE's arithmetic, in units of the layer's scale max(1, max |pre-activation|) -- over those
    entries.  A ReLU entry's distance is |pre-activation|.  A winner-set entry (e, j) the device selects and the oracle
    does not is max(segment max - y[e,j], -y[e,j]) away (how far the edge is from attaining a positive maximum); one the
    oracle selects and the device does not is y[e,j] away when the device selected nothing in that (detection, column)
    (its maximum was not positive), 0 when the device selected other edges (those are measured by the first rule).
    Two fp32 implementations that agree to 1e-5 can only disagree at distances of that order; a wrong winner rule
    (dropped ties, a non-maximal edge) shows up as a distance of the order of the activations themselves."""
    own, pre = out["pins"], out["pre"]
    c_idx = np.asarray(out["neighbor_pair_idxs"][:, 0])
    N = int(out["num_dets"])
    n_diff, worst, where = 0, 0.0, None
    for key in ("pw", "im", "r", "rn", "h1", "q", "x"):
        for i, (a, b, y) in enumerate(zip(dev[key], own[key], pre[key])):
            d = np.asarray(a) != np.asarray(b)
            if d.any():
                w = float(np.abs(y[d]).max() / max(1.0, float(np.abs(y).max())))
                n_diff += int(d.sum())
                if w > worst:
                    worst, where = w, (key, i)
    for i, (a, b, y) in enumerate(zip(dev["sel"], own["sel"], pre["sel"])):
        a, b = np.asarray(a), np.asarray(b)
        d = a != b
        if not d.any():
            continue
        scale = max(1.0, float(np.abs(y).max()))
        top = np.full((N, y.shape[1]), -np.inf, y.dtype)
        np.maximum.at(top, c_idx, y)
        dev_any = np.zeros((N, y.shape[1]), bool)
        np.logical_or.at(dev_any, c_idx, a)
        e, j = np.nonzero(d)
        ye, te = y[e, j], top[c_idx[e], j]
        dist = np.where(a[e, j], np.maximum(np.maximum(te - ye, -ye), 0.0),       # device selects, oracle does not
                        np.where(dev_any[c_idx[e], j], 0.0, np.maximum(ye, 0.0)))   # oracle selects, device does not
        w = float(dist.max() / scale)
        n_diff += int(d.sum())
        if w > worst:
            worst, where = w, ("sel", i)
    return n_diff, worst, where


def winner_records_exact(net, b):
    """Checks block b's SegmentMax records of the HIP forward / backward-preparation against plain numpy reductions of the
    pw_fc2 pre-activations the kernel itself saw (blk_h2, kept with net.keep_edge_activations): maxima bit for bit, tie
    counts, the recorded arg-max edge, the tie flags, the full winner sets (ties included), the winner bitmap and the
    ascending winner list.  Returns the dumped pre-activations [E,64] (to be compared with the oracle's)."""
    E, N, B = int(net.num_edges), int(net.num_dets), net.num_blocks
    dv = net.debug_view
    H = dv("blk_h2", E * 64, index=b).view(E, 64).cpu().numpy()
    rp = net.row_ptr.cpu().numpy().astype(np.int64)
    assert (np.diff(rp) > 0).all(), "every detection has its self pair"
    c_idx = np.repeat(np.arange(N), np.diff(rp))
    pm = dv("blk_pm", N * 64, dtype=torch.int64, index=b).view(N, 64).cpu().numpy()
    pa = dv("blk_parg", N * 64, dtype=torch.int64, index=b).view(N, 64).cpu().numpy()
    R = np.maximum(H, np.float32(0.0))
    top = np.maximum.reduceat(R, rp[:-1], axis=0)
    assert np.array_equal((pm >> 32).astype(np.uint32), top.view(np.uint32)), "segment maxima, bit for bit"
    assert np.array_equal(pm >> 32, pa >> 32), "both records carry the same maximum"
    pos = top > 0
    att = (R == top[c_idx]) & pos[c_idx]                        # attains a positive maximum
    cnt = np.add.reduceat(att.astype(np.int64), rp[:-1], axis=0)
    assert np.array_equal((pm & 0xffffffff)[pos], cnt[pos]), "tie counts of the positive maxima"
    assert ((pm & 0xffffffff)[~pos] >= 1).all()
    arg = (pa & 0xffffffff).astype(np.int64)
    jj = np.broadcast_to(np.arange(64), (N, 64))
    cc = np.broadcast_to(np.arange(N)[:, None], (N, 64))
    assert (arg[pos] >= rp[cc[pos]]).all() and (arg[pos] < rp[cc[pos] + 1]).all(), "an edge of that detection"
    assert np.array_equal(H[arg[pos], jj[pos]], top[pos]), "the recorded edge attains the maximum, bit for bit"
    tf_stride = ((N + 32 + 63) // 64) * 64
    tflag = dv("tflag", B * tf_stride, dtype=torch.uint8).view(B, tf_stride)[b - 1, :N].cpu().numpy()
    assert np.array_equal(tflag != 0, ((cnt > 1) & pos).any(1)), "tie flags"
    sel = device_winner_sets(net, b).cpu().numpy()
    assert np.array_equal(sel, att), "winner sets (all tied edges included), exactly"
    n_words = (E + 63) // 64
    bm_stride = ((n_words + 1 + 255) // 256) * 256
    ewin = dv("ewin", (B + 1) * bm_stride, dtype=torch.int64).view(B + 1, bm_stride)[b - 1, :n_words].cpu().numpy()
    bits = np.unpackbits(ewin.view(np.uint8), bitorder="little")[:E].astype(bool)
    rows = att.any(1)
    assert np.array_equal(bits, rows), "winner bitmap"
    wl_stride = ((E + 64 + 63) // 64) * 64
    want = np.nonzero(rows)[0]
    wlist = dv("wlist", B * wl_stride, dtype=torch.int32).view(B, wl_stride)[b - 1, :want.size].cpu().numpy()
    assert np.array_equal(wlist, want), "ascending winner list"
    return H
