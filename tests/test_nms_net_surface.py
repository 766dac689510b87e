"""The reference's import surface (`nms_net.*`, reference train.py:19-21, nms_net/network.py:12-14) resolves to this
implementation, name by name; plus the GPU-free host logic next to it (learning-rate table, flat-buffer layout)."""
import importlib

import numpy as np
import pytest


def test_every_name_a_caller_of_the_reference_imports():
    import gossipnet_amd
    from nms_net import cfg                                          # reference train.py:19
    from nms_net.network import Gnet                                 # train.py:20
    from nms_net.config import cfg_from_file                         # train.py:21
    from nms_net.roi_pooling_layer import roi_pooling_op, roi_pooling_op_grad      # nms_net/network.py:13
    from nms_net import matching_module                              # nms_net/network.py:14
    from nms_net.dataset import ShuffledDataset, load_roi            # train.py:22, test.py:22
    from nms_net.class_weights import class_equal_weights            # train.py:23
    from nms_net import tools                                        # test.py:20
    assert callable(load_roi) and callable(class_equal_weights) and callable(tools.Timer) and ShuffledDataset.next_batch
    assert cfg is gossipnet_amd.config.cfg and cfg_from_file is gossipnet_amd.config.cfg_from_file
    assert Gnet is gossipnet_amd.network.Gnet
    assert callable(matching_module.detection_matching)
    assert callable(roi_pooling_op.roi_pool) and callable(roi_pooling_op.roi_pool_grad)
    assert matching_module.detection_matching is importlib.import_module("gossipnet_amd.matching_module").detection_matching
    # the gradient registration module exists and binds roi_pool_grad as the backward of roi_pool
    assert roi_pooling_op_grad.RoiPoolFunction.backward is not None
    assert roi_pooling_op_grad.roi_pool_output_shapes((1, 38, 63, 1024), (2000, 5), 7, 7) == [(2000, 7, 7, 1024)] * 2
    # the Gnet surface of SURVEY 8b
    assert Gnet.name == 'gnet' and set(Gnet.get_batch_spec(80)) >= {'dets', 'det_scores', 'det_classes', 'gt_boxes', 'gt_crowd', 'gt_classes'}
    assert set(Gnet.get_batch_spec(80, is_training=False)) == {'dets', 'det_scores', 'det_classes'}


def test_class_equal_weights_known_answer():
    """nms_net/class_weights.py:12-22 + imdb/tools.py:113-124 traced by hand: C = 2, pos_weight 0.1; one image with
    gt classes [1, 1, 2] and 10 detections, one image with gt [2] and no detections: counts start at 1 ->
    [1 + 7, 1 + 2, 1 + 2] = [8, 3, 3], 14 samples; expected [0.9, 0.05, 0.05] -> 14 * e / counts."""
    from gossipnet_amd.config import reset_cfg
    from nms_net.class_weights import class_equal_weights, get_class_counts
    reset_cfg()                   # the reference's defaults: pos_weight 0.1 (config.py:41)
    imdb = {"num_classes": 2, "roidb": [{"gt_classes": np.array([1, 1, 2]), "det_classes": np.ones(10, np.int32)},
                                        {"gt_classes": np.array([2])}]}
    assert get_class_counts(imdb).tolist() == [8, 3, 3]
    w = class_equal_weights(imdb)
    assert np.allclose(w, [14 * 0.9 / 8, 14 * 0.05 / 3, 14 * 0.05 / 3], rtol=1e-6)


def test_datasets_feed_rois_without_touching_the_imdb():
    """nms_net/dataset.py:17-112: load_roi copies the record and scales boxes by im_scale only with images; the shuffled
    set redraws a permutation that cannot fill the next step; batch_size k returns k images (one per step in the reference)."""
    from nms_net.dataset import ShuffledDataset, TestDataset, load_roi
    roidb = [{"dets": np.full((2, 4), float(i), np.float32), "gt_boxes": np.ones((1, 4), np.float32), "id": i} for i in range(5)]
    imdb = {"roidb": roidb, "num_classes": 1}
    r = load_roi(False, roidb[2])
    assert r["im_scale"] == 1.0 and r is not roidb[2] and "im_scale" not in roidb[2]
    with pytest.raises(ValueError):
        load_roi(True, roidb[2])
    r = load_roi(True, dict(roidb[2], imfeats=np.zeros((1, 2, 2, 4), np.float32), im_scale=1.5))
    assert np.array_equal(r["dets"], roidb[2]["dets"] * 1.5) and np.array_equal(roidb[2]["dets"], np.full((2, 4), 2.0, np.float32))
    ts = TestDataset(imdb, 1, False)
    assert len(ts) == 5 and [ts.next_batch()["id"] for _ in range(5)] == [0, 1, 2, 3, 4]
    sd = ShuffledDataset(imdb, 1, False, rng=np.random.default_rng(0))
    first = [sd.next_batch()["id"] for _ in range(5)]
    assert sorted(first) == [0, 1, 2, 3, 4]
    sd = ShuffledDataset(imdb, 2, False, rng=np.random.default_rng(0))
    seen = [tuple(r["id"] for r in sd.next_batch()) for _ in range(4)]
    assert all(len(s) == 2 and s[0] != s[1] for s in seen) and len({i for s in seen[:2] for i in s}) == 4


def test_timer_running_average():
    from nms_net.tools import Timer
    calls = []
    t = Timer(sync=lambda: calls.append(1))
    t.tic(); d = t.toc(average=False)
    t.tic(); avg = t.toc()
    assert t.calls == 2 and len(calls) == 4 and d >= 0 and abs(avg - t.total_time / 2) < 1e-12


def test_learning_rate_table_semantics():
    """reference train.py:26-37 fed consecutive iterations: entry k's rate up to and including its iteration, the last
    rate ever after (hand-traced: cursor 0 returns 0.1 at iterations 1, 2 and moves at 2; 0.01 at 3, 4; then the last)."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.train import LearningRate
    experiment_cfg()
    cfg.train.lr_multi_step = [(2, 0.1), (4, 0.01)]
    lr = LearningRate()
    assert [lr.get_lr(i) for i in range(0, 8)] == [0.1, 0.1, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01]
    # stateless: a run resumed at iteration 3 is on the second rate at once
    assert LearningRate().get_lr(3) == 0.01
    experiment_cfg()
    lr = LearningRate()          # the shipped experiment's table (experiments/coco_multiclass/conf.yaml:5)
    assert lr.get_lr(800000) == 0.0001 and lr.get_lr(800001) == 0.00001 and lr.get_lr(10 ** 7) == 0.00001
    from gossipnet_amd.config import reset_cfg
    reset_cfg()
    lr = LearningRate()          # config.py default table
    assert lr.get_lr(10000) == 0.001 and lr.get_lr(10001) == 0.0001 and lr.get_lr(80001) == 1e-7 and lr.get_lr(10 ** 7) == 1e-7


def test_flat_buffer_offsets_core_packed_imfeats_aligned():
    """Core variables are packed in the C ABI's order (581 793 floats for C=80, B=16); the reduce_imfeats tensors behind
    them start on 16-byte boundaries (fc.hip reads float4) and the padding is zero."""
    from gossipnet_amd.config import cfg, experiment_cfg
    from gossipnet_amd.network import Gnet, param_spec
    experiment_cfg()
    net = Gnet(80, device="cpu")
    offs = net.tensor_offsets()
    sizes = [int(np.prod(s)) for _, s in param_spec(80, 16)]
    assert offs == list(np.concatenate([[0], np.cumsum(sizes)]))
    assert offs[-1] == 581793 == net.params.numel()
    cfg.gnet.imfeats = True
    cfg.gnet.imfeat_dim = 64
    net = Gnet(80, device="cpu", imfeat_channels=32)
    offs = net.tensor_offsets()
    names = [nm for nm, _ in net._spec]
    im = [o for nm, o in zip(names, offs) if nm.startswith("gnet/reduce_imfeats/")]
    assert len(im) == 4 and all(o % 4 == 0 for o in im) and im[0] == 581796 and offs[-1] % 4 == 0
    used = np.zeros(offs[-1], bool)
    for (nm, shape), o in zip(net._spec, offs):
        k = int(np.prod(shape))
        assert not used[o:o + k].any()
        used[o:o + k] = True
    assert float(net.params[~used].abs().sum()) == 0.0 and int((~used).sum()) == offs[-1] - sum(int(np.prod(s)) for _, s in net._spec)
    named = net.flat_to_named(net.params)
    assert named["gnet/reduce_imfeats/fully_connected/weights"].shape == (7 * 7 * 32, 64)
    assert named["gnet/reduce_imfeats/fully_connected_1/biases"].data_ptr() == net.variables["gnet/reduce_imfeats/fully_connected_1/biases"].data_ptr()
    experiment_cfg()


def test_stale_library_is_detected_and_unhashed_library_is_loaded(tmp_path, monkeypatch):
    """_lib.load(): a recorded hash that differs from the sources forces a rebuild attempt; a missing record does not."""
    import warnings
    from gossipnet_amd import _lib, build
    assert build._up_to_date()
    monkeypatch.setattr(_lib, "_lib", None)
    calls = []
    monkeypatch.setattr(build, "build", lambda *a, **k: calls.append(1))
    monkeypatch.setattr(build, "source_hash", lambda: "different")
    _lib.load()
    assert calls == [1]                                   # stale -> build() called
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", _lib.LIB_PATH)  # same library ...
    real_open = open

    def no_hash(path, *a, **k):
        if str(path).endswith(".srchash"):
            raise OSError("no record")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", no_hash)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _lib.load()
    assert calls == [1] and any("no source hash" in str(x.message) for x in w)    # ... without a record: loaded, not rebuilt



@pytest.mark.gpu
def test_trainable_variables_follow_the_reference_rule():
    """network.py:317-322: gnet / resnet variables minus the ignored prefixes (the frozen trunk layers of get_resnet)."""
    import torch
    from gossipnet_amd.config import experiment_cfg
    from gossipnet_amd.network import Gnet
    experiment_cfg()
    net = Gnet(3, device=torch.device("cuda", 0))
    names = [nm for nm, _ in net._spec]
    assert net.trainable_names == names and len(net.trainable_variables) == len(names)
    net._ignore_prefixes = ["gnet/block01/"]
    net._filter_trainable()
    assert net.trainable_names == [nm for nm in names if not nm.startswith("gnet/block01/")]
    assert all(v is net.variables[nm] for v, nm in zip(net.trainable_variables, net.trainable_names))
