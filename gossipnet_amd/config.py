"""Global `cfg` with the reference's defaults and strict YAML merge.

Mirrors nms_net/config.py:10-121 (values are part of the hot-path contract).
EasyDict is not installed here; `AttrDict` gives the same attribute access.
"""
import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _defaults():
    cfg = AttrDict()
    cfg.random_seed = 42
    cfg.prefetch_q_size = 20
    cfg.log_dir = "./log"
    cfg.imfeat_crop_width = 7
    cfg.imfeat_crop_height = 7
    t = cfg.train = AttrDict()
    t.optimizer = "adam"
    t.momentum = 0.9
    t.weight_decay = 0.0005
    t.num_iter = 100000
    t.lr_multi_step = [(10000, 0.001), (80000, 0.0001), (200000, 0.0000001)]
    t.gradient_clipping = -1.0
    t.pos_weight = 0.1
    t.max_num_detections = -1
    t.normalize_loss = False
    t.loss_multiplyer = 1.0
    g = cfg.gnet = AttrDict()
    g.neighbor_thresh = 0.2
    g.shortcut_dim = 128
    g.num_blocks = 16
    g.reduced_dim = 32
    g.pairfeat_dim = 2 * 32
    g.gt_match_thresh = 0.5
    g.num_block_pw_fc = 2
    g.num_block_fc = 2
    g.num_predict_fc = 3
    g.block_dim = 2 * 32
    g.predict_fc_dim = 128
    g.imfeats = False
    g.load_imfeats = False
    g.imfeat_dim = -1
    g.neighbor_feats = False
    # the reference's own defaults (config.py:73-77): no pairwise-feature MLP.  Both shipped experiments override them
    # (experiments/*/conf.yaml: num_pwfeat_fc 3, pwfeat_narrow_dim 32, bias_const_init 0.01 / 0.1) -- see experiment_cfg().
    g.num_pwfeat_fc = 0
    g.pwfeat_dim = 256
    g.pwfeat_narrow_dim = 64
    g.weight_init = "xavier"
    g.bias_const_init = 0.0
    g.freeze_n_imfeat_layers = 3            # (config.py:78: trunk layers whose variables are not trained; the trunk is the caller's here)
    g.pw_feat_multiplyer = 1.0
    return cfg


cfg = _defaults()


def _merge_a_into_b(a, b):
    """config.py:82-112: unknown key -> KeyError, type mismatch -> ValueError."""
    for k, v in a.items():
        if k not in b:
            raise KeyError("{} is not a valid config key".format(k))
        if isinstance(b[k], dict):
            if not isinstance(v, dict):
                raise ValueError("Type mismatch for config key: {}".format(k))
            _merge_a_into_b(v, b[k])
            continue
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], (list, tuple)) and isinstance(v, (list, tuple)):
                pass
            elif isinstance(b[k], float) and isinstance(v, int) and not isinstance(v, bool):
                v = float(v)
            else:
                raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(b[k]), type(v), k))
        b[k] = v


def cfg_from_file(filename, strict=False):
    """config.py:115-121.  Keys of the reference's config that only steer out-of-scope
    subsystems (imdb names, detector, save_iter ...) are ignored unless strict=True."""
    with open(filename, "r") as f:
        y = yaml.safe_load(f)
    _merge_a_into_b(_prune(y, cfg, strict), cfg)


def _prune(a, b, strict=False):
    """The part of a (a loaded conf.yaml) whose keys exist in b (the cfg of this build)."""
    out = {}
    for k, v in a.items():
        if k not in b:
            if strict:
                raise KeyError("{} is not a valid config key".format(k))
            continue
        out[k] = _prune(v, b[k], strict) if isinstance(v, dict) and isinstance(b[k], dict) else v
    return out


def reset_cfg():
    """Back to the reference's defaults (nms_net/config.py:10-79)."""
    d = _defaults()
    cfg.clear()
    cfg.update(d)


# every key the two shipped experiments set in their conf.yaml (experiments/coco_multiclass/conf.yaml:1-24,
# experiments/coco_person/conf.yaml:1-22), values as written there -- what a run started with
# `cfg_from_file(experiments/<name>/conf.yaml)` sees on top of the defaults (keys that only steer out-of-scope subsystems --
# imdb / detector names, val_iter, save_iter, only_class -- are dropped by the same pruning cfg_from_file applies)
EXPERIMENTS = {
    "coco_multiclass": {"train": {"optimizer": "adam", "weight_decay": 0.0005, "num_iter": 2000000,
                                  "lr_multi_step": [[800000, 0.0001], [2000000, 0.00001]], "detector": "FRCN_train",
                                  "imdb": "coco_2014_train", "val_imdb": "coco_2014_minival", "val_iter": 20000, "save_iter": 20000,
                                  "max_num_detections": 600, "pos_weight": 0.3},
                        "gnet": {"imfeats": False, "neighbor_feats": False, "num_pwfeat_fc": 3, "bias_const_init": 0.01,
                                 "imfeat_dim": 1024, "pwfeat_narrow_dim": 32}},
    "coco_person": {"gnet": {"bias_const_init": 0.1, "neighbor_feats": False, "num_blocks": 1, "num_pwfeat_fc": 3,
                             "pwfeat_narrow_dim": 32},
                    "random_seed": 42,
                    "train": {"detector": "FRCN_train", "imdb": "coco_2014_train", "lr_multi_step": [[1000000, 0.0001], [2000000, 1.0e-05]],
                              "max_num_detections": 600, "num_iter": 2000000, "only_class": "person", "optimizer": "adam",
                              "pos_weight": 0.1, "save_iter": 20000, "val_imdb": "coco_2014_minival", "val_iter": 20000,
                              "weight_decay": 0.0005}},
}


def experiment_cfg(name="coco_multiclass", **gnet_overrides):
    """reset_cfg() + the overrides of a shipped experiment's conf.yaml (+ cfg.gnet overrides of the caller): the configuration
    BASELINE.json's numbers are quoted on.  Tests, bench.py and the tools start from here; `cfg` itself starts from the
    reference's defaults, as `from nms_net import cfg` does (num_pwfeat_fc = 0, pwfeat_narrow_dim = 64, bias 0.0 -- a parameter
    layout WITHOUT the pairwise-feature MLP: a checkpoint of one of the shipped experiments needs experiment_cfg(<its name>) or
    cfg_from_file(<its conf.yaml>) before the Gnet is built, as with the reference)."""
    reset_cfg()
    _merge_a_into_b(_prune(EXPERIMENTS[name], cfg), cfg)
    for k, v in gnet_overrides.items():
        if k not in cfg.gnet:
            raise KeyError("{} is not a valid cfg.gnet key".format(k))
        cfg.gnet[k] = v
    return cfg
