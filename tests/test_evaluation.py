"""Host-side harness around the path: AP, detection pickle, checkpoints."""
import os
import pickle

import numpy as np
import pytest

from gossipnet_amd import evaluation as ev


def test_average_precision_kats():
    # perfect ranking: every recall threshold sees precision 1 -> AP 100
    assert ev.average_precision(np.array([1, 1, 0, 0]), 2) == pytest.approx(100.0)
    # hand-derived (train.py:189-206): labels [1,0,1], 2 objects: recall [0,.5,.5,1,1,2], envelope precision
    # [1,1,2/3,2/3,0,0]; thresholds 0..0.5 -> 1 (51 points), 0.51..1.0 -> 2/3 (50 points)
    assert ev.average_precision(np.array([1, 0, 1]), 2) == pytest.approx((51 * 1.0 + 50 * 2 / 3) / 101 * 100, rel=1e-6)
    # nothing found
    assert ev.average_precision(np.array([0, 0]), 3) == pytest.approx(100.0 / 101, rel=1e-6)


def test_compute_aps_groups_by_class():
    roidb = [{"gt_crowd": np.array([False, False, True]), "gt_classes": np.array([1, 2, 1])}]
    scores = np.array([0.9, 0.8, 0.7, 0.6]); classes = np.array([1, 2, 1, 2]); labels = np.array([1, 0, 0, 1])
    m, multi, per = ev.compute_aps(scores, classes, labels, roidb)
    assert per[0] == pytest.approx(100.0) and len(per) == 2
    assert m == pytest.approx(np.mean(per))
    assert multi == pytest.approx(ev.average_precision(np.array([1, 0, 0, 1]), 2))


def test_save_dets_format(tmp_path):
    d = [{"id": 7, "dets": np.array([[0, 0, 1, 1], [1, 1, 2, 2.]], np.float32), "det_classes": np.array([1, 2]),
          "det_scores": np.array([0.5, -1.0], np.float32)}]
    f = str(tmp_path / "dets.pkl")
    ev.save_dets(["__background__", "person", "dog"], {"person": 1, "dog": 18}, d, f)
    dets, image_ids, cat_ids = pickle.load(open(f, "rb"))
    assert cat_ids == [-1, 1, 18] and image_ids == [7]
    assert dets[0][0].shape == (0, 5) and dets[1][0].tolist() == [[0, 0, 1, 1, 0.5]] and dets[2][0][0, 4] == -1.0


@pytest.mark.gpu
def test_val_run_rescore_and_checkpoint_roundtrip(tmp_path):
    import torch
    from gossipnet_amd import checkpoint as ck
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer
    from tests.util import make_pair, make_image
    net, orc = make_pair(80, 2)
    roidb = [dict(make_image(80 + 10 * i, 80, seed=i), id=i, im_scale=1.0) for i in range(3)]
    m, multi, per = ev.val_run(net, roidb)
    assert 0.0 <= multi <= 100.0 and len(per) >= 1
    # same labels/scores through the oracle give the same AP
    sc, lb, cl = [], [], []
    for r in roidb:
        o = orc.forward(r)
        mask = o["weights"].numpy() > 0
        sc.append(o["prediction"].detach().numpy()[mask]); lb.append(o["labels"][mask]); cl.append(r["det_classes"][mask])
    m2, multi2, _ = ev.compute_aps(np.concatenate(sc), np.concatenate(cl), np.concatenate(lb), roidb)
    assert multi == pytest.approx(multi2, abs=1e-3) and m == pytest.approx(m2, abs=1e-3)
    dets, sec, nd = ev.rescore(net, roidb)
    assert len(dets) == 3 and sec > 0 and nd == pytest.approx(90.0)
    # checkpoint keyed by TF names, with Adam slots
    cfg.train.optimizer = "adam"
    opt = Optimizer(net)
    net.run(roidb[0]); opt.apply_gradients(1e-3)
    path = ck.save(net, ck.checkpoint_name(5, str(tmp_path)), global_step=5, optimizer=opt)
    before = net.params.clone(); mb = opt.m.clone()
    net.params.zero_(); opt.m.zero_()
    assert ck.load(net, path, optimizer=opt) == 5
    assert torch.equal(net.params, before) and torch.equal(opt.m, mb)
    z = np.load(path)
    assert "gnet/block2/pw_fc1/weights" in z.files and "gnet/predict/logits/fully_connected/biases/Adam_1" in z.files
