"""The mask-pinned backward of the oracle (GnetOracle.forward(pins=...), used by the GPU gradient tests):
pinned to its OWN smooth piece it must reproduce its ordinary autograd gradient, and flipping a pinned mask
entry must change the gradient (the pins are really what the backward pass uses)."""
import numpy as np
import pytest

from gossipnet_amd.synthetic import make_image
from oracle import gnet_oracle as go


@pytest.mark.parametrize("n,c,b", [(40, 1, 2), (120, 80, 3)])
def test_pinned_to_own_piece_equals_autograd(n, c, b):
    orc = go.GnetOracle(c, b)
    batch = make_image(n, c, seed=3)
    out, g = orc.forward_backward(batch, keep=True)
    out2, g2 = orc.forward_backward(batch, pins=out["pins"])
    assert float(out2["loss"]) == float(out["loss"])
    for k in g:
        assert np.abs(g[k] - g2[k]).max() <= 1e-6 * max(1e-30, np.abs(g[k]).max()), k


def test_pins_drive_the_backward():
    orc = go.GnetOracle(1, 1)
    batch = make_image(30, 1, seed=0)
    out, g = orc.forward_backward(batch, keep=True)
    pins = {k: [a.copy() for a in v] for k, v in out["pins"].items()}
    pins["q"][0][:] = False                     # kill fc1's ReLU in the backward pass only
    out2, g2 = orc.forward_backward(batch, pins=pins)
    assert float(out2["loss"]) == float(out["loss"])
    assert np.abs(g2["gnet/block1/fc1/weights"]).max() == 0.0
    assert np.abs(g2["gnet/block1/fc2/weights"] - g["gnet/block1/fc2/weights"]).max() == 0.0


def test_tied_winners_share_the_gradient():
    """Duplicate detections: exact positive ties; the pinned winner sets carry every tied edge and the split is
    TF's (grad / number selected)."""
    base = make_image(20, 1, seed=1)
    rep = np.repeat(np.arange(20), 2)
    batch = dict(base)
    for k in ("dets", "det_scores", "det_classes"):
        batch[k] = base[k][rep]
    orc = go.GnetOracle(1, 1, bias_init=0.5)
    out, g = orc.forward_backward(batch, keep=True)
    sel = out["pins"]["sel"][0]
    c_idx = out["neighbor_pair_idxs"][:, 0]
    cnt = np.zeros((40, 64)); np.add.at(cnt, c_idx, sel)
    assert (cnt > 1).any()
    _, g2 = orc.forward_backward(batch, pins=out["pins"])
    for k in g:
        assert np.abs(g[k] - g2[k]).max() <= 1e-6 * max(1e-30, np.abs(g[k]).max()), k
