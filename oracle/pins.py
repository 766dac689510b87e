"""TEST INFRASTRUCTURE (like everything under oracle/): helpers that compare the HIP backward pass with the
oracle on the SAME smooth piece of the piecewise-smooth loss.  Imported by tests/ and __graft_entry__.smoke()."""
import numpy as np
import torch

from . import gnet_oracle as go


def grad_errors(net, gref, c, b, imfeat=None, neighbor_feats=False):
    """Per-tensor max |g_hip - g_ref| / max |g_ref| (TF variable name -> error)."""
    errs = {}
    for name, shape in go.param_spec(c, b, imfeat, neighbor_feats):
        g = net.gradients[name].detach().cpu().numpy().reshape(-1)       # (by name: the flat buffer may hold alignment padding)
        gr = np.asarray(gref[name], np.float64).reshape(-1)
        m = np.abs(gr).max() if gr.size else 0.0
        errs[name] = float(np.abs(g - gr).max() / m) if m > 0 else float(np.abs(g).max())
    return errs


def gpu_pins(net, image=None):
    """The smooth piece the HIP backward differentiated: ReLU masks (out > 0 on the HIP forward's own
    activations, TF ReluGrad) and the segment-max winner sets (per-edge column masks of winners_mark, ties
    included) -- in the layout GnetOracle.forward(pins=...) takes.  The Gnet must have been run with
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
    pins = {"pw": [cpu(dv("pw_h1", E * 256).view(E, 256)[e0:e1] > 0), cpu(dv("pw_h2", E * 256).view(E, 256)[e0:e1] > 0),
                   cpu(net.pw_feats[e0:e1] > 0)],
            "r": [], "rn": [], "h1": [], "sel": [], "q": [], "x": [], "im": []}
    if getattr(net, "_imfeats", False):
        pins["im"] = [cpu(a_[d0:d1] > 0) for a_ in net._imfeat_acts]
    # winner sets: the recorded arg-max edge of every (detection, column) with a positive maximum, plus -- for
    # detections flagged as tied -- the extra winners of the tied columns (csrc/backward_edge.hip winners_mark)
    wl_stride = ((E + 64 + 63) // 64) * 64            # edge_geom(): xm_stride = wl_stride, tf_stride
    tf_stride = ((N + 32 + 63) // 64) * 64
    xmask = dv("xmask", B * wl_stride, dtype=torch.int64).view(B, wl_stride)
    tflag = dv("tflag", B * tf_stride, dtype=torch.uint8).view(B, tf_stride)
    edge_c = net.neighbor_pair_idxs[:, 0]
    shifts = torch.arange(64, device=xmask.device, dtype=torch.int64).view(1, 64)
    cols = torch.arange(64, device=xmask.device).view(1, 64).expand(N, 64)

    def winner_sets(b):
        parg = dv("blk_parg", N * 64, dtype=torch.int64, index=b).view(N, 64)
        valid = (parg >> 32) != 0
        sel = torch.zeros(E, 64, dtype=torch.bool, device=parg.device)
        sel[(parg & 0xffffffff)[valid], cols[valid]] = True
        tied_rows = tflag[b - 1, :N][edge_c] != 0                    # edges of flagged detections
        extra = ((xmask[b - 1, :E].view(-1, 1) >> shifts) & 1) != 0
        return sel | (extra & tied_rows.view(-1, 1))

    bf = net.block_feats
    for b in range(1, B + 1):
        pins["r"].append(cpu(dv("blk_r", N * 32, index=b).view(N, 32)[d0:d1] > 0))
        if net._buf.blk_rnb[b]:
            pins["rn"].append(cpu(dv("blk_rnb", N * 32, index=b).view(N, 32)[d0:d1] > 0))
        pins["h1"].append(cpu(dv("blk_h1", E * 64, index=b).view(E, 64)[e0:e1] > 0))
        pins["sel"].append(cpu(winner_sets(b)[e0:e1]))
        pins["q"].append(cpu(dv("blk_q", N * 64, index=b).view(N, 64)[d0:d1] > 0))
        pins["x"].append(cpu(bf[b][d0:d1] > 0))
    return pins
