"""The C-ABI library loads on a CPU-only host and exports every symbol include/gossipnet_hip.h declares
(no compute calls here: there is no GPU).  Also checks the host-side, GPU-free parts of the API."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gossipnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gossipnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:gnet|det_matching|roi_pool)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), n
    assert set(_lib.EXPORTS) <= set(names)
    assert b"gfx950" in lib.gnet_version()


def test_abi_guard_sizes_and_version():
    """The ctypes mirrors of the four structs have the library's sizes and field offsets, the header's version is the
    binding's, and load() refuses a library that reports anything else."""
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "gossipnet_hip.h")).read()
    assert int(re.search(r"#define GNET_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION == lib.gnet_abi_version()
    sizes = (C.c_size_t * 8)()
    assert lib.gnet_abi_sizes(sizes) == len(_lib.KCLASSES)
    assert list(sizes) == _lib.abi_mirror()
    assert sizes[3] == C.sizeof(_lib.gnet_buffers) > 5000 and sizes[0] == 17 * 4
    assert lib.gnet_abi_sizes(None) == _lib.ERR_INVALID

    class Fn(object):                          # stands in for a ctypes function object (check_abi sets restype / argtypes)
        def __init__(self, f):
            self.f = f

        def __call__(self, *a):
            return self.f(*a)

    def sizes_shifted(out):                    # a library built from a header with one more pointer in gnet_buffers
        for i, v in enumerate(_lib.abi_mirror()):
            out[i] = v + (8 if i >= 3 else 0)
        return len(_lib.KCLASSES)

    class Fake(object):
        pass
    fake = Fake()
    fake.gnet_abi_version, fake.gnet_abi_sizes = Fn(lambda: _lib.ABI_VERSION), Fn(sizes_shifted)
    with pytest.raises(_lib.GnetError, match="layout"):
        _lib.check_abi(fake)
    fake.gnet_abi_version, fake.gnet_abi_sizes = Fn(lambda: _lib.ABI_VERSION - 1), Fn(lambda out: lib.gnet_abi_sizes(out))
    with pytest.raises(_lib.GnetError, match="ABI version"):
        _lib.check_abi(fake)
    with pytest.raises(_lib.GnetError, match="predates"):
        _lib.check_abi(Fake())


def test_param_count_and_unsupported_config():
    lib = _lib.load()
    ok = _lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 32, 3, 128, 3, 2, 2, 1.0, 0)
    assert lib.gnet_param_count(C.byref(ok)) == 581793          # multiclass, B = 16
    one = _lib.gnet_config(1, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 32, 3, 128, 3, 2, 2, 1.0, 0)
    assert lib.gnet_param_count(C.byref(one)) == 541345
    # the reference's default hyper-parameters (config.py:73-75: num_pwfeat_fc = 0, pwfeat_narrow_dim = 64): no pairwise-feature
    # MLP, every block's pw_fc1 reads the 2C'+7 raw feature columns -- 16 x ((167 + 64) x 64 + 26 976 - 96 x 64) + head 33 153
    raw = _lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 64, 0, 128, 3, 2, 2, 1.0, 0)
    blk = 128 * 32 + 32 + (167 + 64) * 64 + 64 + 64 * 64 + 64 + 64 * 64 + 64 + 64 * 128 + 128
    assert lib.gnet_param_count(C.byref(raw)) == 16 * blk + 33153
    raw1 = _lib.gnet_config(1, 2, 0.2, 0, 1.0, 128, 32, 64, 256, 64, 0, 128, 3, 2, 2, 1.0, 0)
    assert lib.gnet_param_count(C.byref(raw1)) == 2 * (blk - (167 - 9) * 64) + 33153
    for bad in (_lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 64, 3, 128, 3, 2, 2, 1.0, 0),     # a 64-wide narrow layer
                _lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 32, 2, 128, 3, 2, 2, 1.0, 0),     # a two-layer pw-MLP
                _lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 64, 64, 256, 32, 3, 128, 3, 2, 2, 1.0, 0)):    # reduced_dim 64
        assert lib.gnet_param_count(C.byref(bad)) == _lib.ERR_UNSUPPORTED


def test_workspace_query_and_plan_argument_checks():
    lib = _lib.load()
    cfg = _lib.gnet_config(80, 16, 0.2, 0, 1.0, 128, 32, 64, 256, 32, 3, 128, 3, 2, 2, 1.0, 0)
    sh = _lib.gnet_shape(1, 2000, 80, 158724, 160000)
    train = lib.gnet_workspace_bytes(C.byref(cfg), C.byref(sh), 1)
    infer = lib.gnet_workspace_bytes(C.byref(cfg), C.byref(sh), 0)
    assert train > infer > 0
    buf = _lib.gnet_buffers()
    assert lib.gnet_plan(C.byref(cfg), C.byref(sh), 1, None, train, C.byref(buf)) == _lib.ERR_INVALID
    fake = C.c_void_p(1 << 20)   # aligned, never dereferenced by gnet_plan
    assert lib.gnet_plan(C.byref(cfg), C.byref(sh), 1, fake, train - 1, C.byref(buf)) == _lib.ERR_WORKSPACE
    assert lib.gnet_plan(C.byref(cfg), C.byref(sh), 1, fake, train, C.byref(buf)) == _lib.OK
    assert buf.row_ptr == 1 << 20 and buf.arena and buf.pw_h1 and buf.block_feats[16] and not buf.block_feats[17]


def test_param_spec_matches_oracle_and_layout():
    from gossipnet_amd.config import experiment_cfg
    from gossipnet_amd.network import param_spec
    from oracle import gnet_oracle as go
    from gossipnet_amd.config import cfg, reset_cfg
    experiment_cfg()
    for c, b in ((80, 16), (1, 1), (3, 5)):
        assert param_spec(c, b) == go.param_spec(c, b)
    reset_cfg()                                  # the reference's defaults: no pairwise-feature MLP
    assert cfg.gnet.num_pwfeat_fc == 0 and cfg.gnet.pwfeat_narrow_dim == 64 and cfg.gnet.bias_const_init == 0.0   # config.py:73-77
    for c, b in ((80, 16), (1, 2)):
        assert param_spec(c, b) == go.param_spec(c, b, num_pwfeat_fc=0)
        assert ("gnet/block1/pw_fc1/weights", (2 * c + 7 + 64, 64)) in param_spec(c, b)
    experiment_cfg()


def test_config_merge_rules(tmp_path):
    from gossipnet_amd.config import cfg, cfg_from_file, experiment_cfg
    experiment_cfg()
    p = tmp_path / "conf.yaml"
    p.write_text("gnet:\n  num_blocks: 1\n  bias_const_init: 0.1\ntrain:\n  imdb: coco_2014_train\n")
    cfg_from_file(str(p))                       # out-of-scope keys of the reference configs are ignored
    assert cfg.gnet.num_blocks == 1 and cfg.gnet.bias_const_init == 0.1
    with pytest.raises(KeyError):
        cfg_from_file(str(p), strict=True)      # config.py:91-92 semantics
    bad = tmp_path / "bad.yaml"
    bad.write_text("gnet:\n  num_blocks: 'one'\n")
    with pytest.raises(ValueError):
        cfg_from_file(str(bad))                 # config.py:95-103: type mismatch
    experiment_cfg()
    assert cfg.gnet.num_blocks == 16


def test_reference_experiment_configs_load():
    """The values of the two shipped experiments select the compiled configuration."""
    from gossipnet_amd.config import cfg, cfg_from_file, experiment_cfg
    import tempfile
    experiment_cfg()
    text = "gnet:\n  bias_const_init: 0.1\n  neighbor_feats: false\n  num_blocks: 1\n  num_pwfeat_fc: 3\n  pwfeat_narrow_dim: 32\nrandom_seed: 42\n"
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(text)
    cfg_from_file(f.name)
    assert cfg.gnet.num_blocks == 1 and cfg.gnet.num_pwfeat_fc == 3
    experiment_cfg()


def test_experiment_cfg_equals_the_reference_conf_files():
    """experiment_cfg(<name>) == reset + cfg_from_file(the reference's experiments/<name>/conf.yaml), key for key (EXPERIMENTS
    restates every key of the two files; this container has them, the GPU box does not: skipped there)."""
    import copy
    import os
    from gossipnet_amd.config import cfg, cfg_from_file, experiment_cfg, reset_cfg
    root = "/root/reference/experiments"
    if not os.path.isdir(root):
        pytest.skip("the reference tree is not on this machine")

    def norm(d):
        return {k: (norm(v) if isinstance(v, dict) else ([list(x) for x in v] if isinstance(v, (list, tuple)) else v)) for k, v in d.items()}
    for name in ("coco_multiclass", "coco_person"):
        experiment_cfg(name)
        a = norm(copy.deepcopy(dict(cfg)))
        reset_cfg()
        cfg_from_file(os.path.join(root, name, "conf.yaml"))
        assert a == norm(copy.deepcopy(dict(cfg))), name
    experiment_cfg()


def test_device_batch_offsets_on_cpu():
    from gossipnet_amd.network import DeviceBatch
    from gossipnet_amd.synthetic import make_image
    imgs = [make_image(30, 80, seed=0), make_image(1, 80, seed=1), make_image(75, 80, seed=2)]
    db = DeviceBatch(imgs, "cpu")
    assert db.det_off_h.tolist() == [0, 30, 31, 106]
    m = [len(i["gt_boxes"]) for i in imgs]
    assert db.gt_off_h.tolist() == [0, m[0], m[0] + m[1], sum(m)]
    assert db.anno_off_h.tolist() == [0, 30 * m[0], 30 * m[0] + m[1], 30 * m[0] + m[1] + 75 * m[2]]
    assert db.n_det == 106 and db.has_gt
    with pytest.raises(_lib.InvalidArgumentError):
        DeviceBatch({"dets": np.zeros((3, 4)), "det_scores": np.zeros(2), "det_classes": np.zeros(3)}, "cpu")


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gossipnet_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gossipnet_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"(import|from)\s+oracle|oracle[./_]|liboracle", text), os.path.join(dirpath, fn)
