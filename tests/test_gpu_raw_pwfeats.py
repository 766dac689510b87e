"""The reference's DEFAULT hyper-parameters: cfg.gnet.num_pwfeat_fc = 0 (nms_net/config.py:73) -- no pairwise-feature MLP,
`pw_feats` is the raw _geometry_feats matrix [E, 2C'+7] (network.py:197-221) and every block's pw_fc1 is [2C'+7+64, 64]
(network.py:367-385).  HIP path through the C ABI against the CPU oracle: integers bit-exact, activations / logits / losses
<= 1e-5, every parameter gradient <= 1e-5 on the pinned smooth piece (tests/test_gpu_backward.py explains the pinning), at the
small shapes, the SURVEY 8c fixture shape (300, 80, 16) and the headline image (2000, 80, 16)."""
import numpy as np
import pytest
import torch

from tests.util import make_pair, rel_err, make_image, grad_errors, gpu_pins
from tests.test_gpu_backward import check_outputs, KINK, PINNED, TIGHT
from oracle.pins import mask_disagreements, winner_records_exact

pytestmark = pytest.mark.gpu


def forward_checks(net, ref, blocks):
    assert np.array_equal(net.neighbor_pair_idxs.cpu().numpy(), ref["neighbor_pair_idxs"]), "neighbour pairs bit-exact"
    raw = net.pw_feats.cpu().numpy()
    assert raw.shape == ref["raw_pw_feats"].shape
    assert rel_err(raw, ref["raw_pw_feats"]) < 2e-6, "Gnet.pw_feats = the raw feature columns"
    assert np.array_equal(raw[:, :raw.shape[1] - 7], ref["raw_pw_feats"][:, :raw.shape[1] - 7]), "score columns + IoU bit-exact"
    bf = net.block_feats
    for k in range(1, blocks + 1):
        assert rel_err(bf[k].cpu().numpy(), ref["block_feats"][k].detach().numpy()) < 1e-5, "block_feats[%d]" % k
    assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < 1e-5


@pytest.mark.parametrize("n,c,b,bias", [(6, 1, 1, 0.01), (33, 1, 2, 0.5), (64, 80, 2, 0.01), (150, 80, 3, 0.5), (200, 1, 1, 0.0)])
def test_raw_pairwise_features_small(n, c, b, bias):
    cw = np.linspace(0.5, 1.5, c + 1).astype(np.float32)
    net, orc = make_pair(c, b, class_weights=cw, bias=bias, num_pwfeat_fc=0)
    net.keep_edge_activations = True
    for seed in range(4):
        batch = make_image(n, c, seed=seed)
        ref, gref = orc.forward_backward(batch, keep=True)
        net.run(batch)
        torch.cuda.synchronize()
        forward_checks(net, ref, b)
        check_outputs(net, ref)
        n_diff, worst, where = mask_disagreements(gpu_pins(net), ref)
        assert worst <= KINK, (seed, n_diff, worst, where)
        for blk in range(1, b + 1):
            winner_records_exact(net, blk)
        if n_diff == 0:
            unpinned = grad_errors(net, gref, c, b, num_pwfeat_fc=0)
            assert max(unpinned.values()) <= TIGHT, (seed, max(unpinned.items(), key=lambda kv: kv[1]))
        _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
        pinned = grad_errors(net, gpin, c, b, num_pwfeat_fc=0)
        assert max(pinned.values()) <= PINNED, (seed, max(pinned.items(), key=lambda kv: kv[1]))


@pytest.mark.parametrize("n,c,b", [(300, 80, 16), (2000, 80, 16), (1000, 1, 16)])
def test_raw_pairwise_features_16_blocks(n, c, b):
    net, orc = make_pair(c, b, num_pwfeat_fc=0)
    net.keep_edge_activations = True
    batch = make_image(n, c, seed=3)
    ref = orc.forward(batch, keep=True)
    net.run(batch)
    torch.cuda.synchronize()
    forward_checks(net, ref, b)
    check_outputs(net, ref)
    n_diff, worst, where = mask_disagreements(gpu_pins(net), ref)
    assert worst <= KINK, (n_diff, worst, where)
    _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
    pinned = grad_errors(net, gpin, c, b, num_pwfeat_fc=0)
    assert max(pinned.values()) <= PINNED, max(pinned.items(), key=lambda kv: kv[1])


def test_raw_batch_of_images_and_multiplier():
    """A block-diagonal batch (gradient = sum over the images), pw_feat_multiplyer != 1, inference mode = training logits."""
    c, b = 80, 3
    net, orc = make_pair(c, b, bias=0.5, num_pwfeat_fc=0, pw_feat_multiplyer=0.7)
    net.keep_edge_activations = True
    imgs = [make_image(n, c, seed=10 + i) for i, n in enumerate((40, 1, 97))]
    total = None
    net.run(imgs)
    torch.cuda.synchronize()
    logits = net.prediction.cpu().numpy().copy()
    off = 0
    for i, im in enumerate(imgs):
        ref = orc.forward(im, keep=True)
        nd = im["dets"].shape[0]
        assert rel_err(logits[off:off + nd], ref["prediction"].detach().numpy()) < 1e-5
        _, g = orc.forward_backward(im, pins=gpu_pins(net, i))
        total = g if total is None else {k: total[k] + g[k] for k in g}
        off += nd
    errs = grad_errors(net, total, c, b, num_pwfeat_fc=0)
    assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])
    grads = net.grads.clone()
    net.run(imgs)
    torch.cuda.synchronize()
    assert torch.equal(grads, net.grads), "bitwise reproducible"
    net.run([{k: im[k] for k in ("dets", "det_scores", "det_classes")} for im in imgs])
    torch.cuda.synchronize()
    assert rel_err(net.prediction.cpu().numpy(), logits) < 1e-6


def test_raw_with_neighbor_feats():
    """num_pwfeat_fc = 0 together with cfg.gnet.neighbor_feats (network.py:356-365): the neighbour score term rides on the rn
    table of the SECOND reduce FC's features; its variables follow fc2 in every block's group."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet
    from oracle import gnet_oracle as go
    c, b = 80, 3
    experiment_cfg(num_pwfeat_fc=0, pwfeat_narrow_dim=64, num_blocks=b, neighbor_feats=True)
    params = go.init_params(c, b, neighbor_feats=True, num_pwfeat_fc=0)
    net = Gnet(c)
    net.keep_edge_activations = True
    net.load_params(params)
    orc = go.GnetOracle(c, b, params=params, neighbor_feats=True, num_pwfeat_fc=0)
    for seed in (0, 1):
        batch = make_image(150, c, seed=seed)
        ref = orc.forward(batch, keep=True)
        net.run(batch)
        torch.cuda.synchronize()
        forward_checks(net, ref, b)
        check_outputs(net, ref)
        _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
        errs = grad_errors(net, gpin, c, b, neighbor_feats=True, num_pwfeat_fc=0)
        assert any("reduce_dim_neighbor" in k for k in errs)
        assert max(errs.values()) <= PINNED, max(errs.items(), key=lambda kv: kv[1])
    experiment_cfg()


def test_raw_exact_ties_from_duplicate_detections():
    """Duplicate detections: exact positive ties of the segment maximum (TF splits the gradient evenly) -- the tie repair
    recomputes pw_fc1 / pw_fc2 from the padded geometry operand and the self-pair rows of rn."""
    net, orc = make_pair(80, 2, bias=0.5, num_pwfeat_fc=0)
    net.keep_edge_activations = True
    for seed in range(2):
        base = make_image(60, 80, seed=seed)
        rep = np.repeat(np.arange(60), 3)
        np.random.default_rng(seed).shuffle(rep)
        batch = dict(base)
        for k in ("dets", "det_scores", "det_classes"):
            batch[k] = base[k][rep]
        ref = orc.forward(batch, keep=True)
        net.run(batch)
        torch.cuda.synchronize()
        pm = net.debug_view("blk_pm", 180 * 64, dtype=torch.int64, index=1).cpu().numpy()
        assert (((pm & 0xffffffff) > 1) & ((pm >> 32) > 0)).any(), "test must contain positive ties"
        assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < 1e-5
        for blk in (1, 2):
            winner_records_exact(net, blk)
        n_diff, worst, where = mask_disagreements(gpu_pins(net), ref)
        assert worst <= KINK, (seed, n_diff, worst, where)
        _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
        pinned = grad_errors(net, gpin, 80, 2, num_pwfeat_fc=0)
        assert max(pinned.values()) <= PINNED, (seed, max(pinned.items(), key=lambda kv: kv[1]))


def test_raw_train_steps_and_checkpoint_names(tmp_path):
    """The training step around the path (train.py:64-77, 316-320) and a checkpoint keyed by the TF variable names with the
    reference's default hyper-parameters: no gnet/pw_feats/* variables, pw_fc1 is [2C'+7+64, 64]."""
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer, train_step
    from gossipnet_amd import checkpoint
    net, _ = make_pair(80, 2, num_pwfeat_fc=0)
    net.weight_reg = cfg.train.weight_decay
    opt = Optimizer(net)
    batch = make_image(150, 80, seed=0)
    losses = [float(train_step(net, opt, batch, 1e-3)) for _ in range(8)]
    assert losses[-1] < losses[0]
    names = [nm for nm, _ in net._spec]
    assert not any(nm.startswith("gnet/pw_feats/") for nm in names)
    assert tuple(net.variables["gnet/block1/pw_fc1/weights"].shape) == (2 * 80 + 7 + 64, 64)
    path = checkpoint.save(net, str(tmp_path / "gnet-8"), global_step=8, optimizer=opt)
    before = net.params.clone()
    net.params.zero_()
    checkpoint.load(net, path, optimizer=opt)
    assert torch.equal(net.params, before)
