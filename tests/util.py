import numpy as np
import torch

from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.synthetic import make_image
from oracle import gnet_oracle as go


def make_pair(num_classes, num_blocks, seed_params=42, class_weights=None, normalize_loss=False, bias=0.01, num_pwfeat_fc=3,
              pw_feat_multiplyer=1.0):
    """(Gnet on cuda:0, GnetOracle) sharing the same parameters.  num_pwfeat_fc = 3: the shipped experiments' configuration;
    0: the reference's default (no pairwise-feature MLP, pwfeat_narrow_dim at its default 64, ignored)."""
    from gossipnet_amd.network import Gnet
    experiment_cfg()
    cfg.gnet.num_blocks = num_blocks
    cfg.gnet.bias_const_init = bias
    cfg.train.normalize_loss = normalize_loss
    cfg.gnet.pw_feat_multiplyer = pw_feat_multiplyer
    if num_pwfeat_fc != 3:
        cfg.gnet.num_pwfeat_fc = num_pwfeat_fc
        cfg.gnet.pwfeat_narrow_dim = 64
    params = go.init_params(num_classes, num_blocks, seed=seed_params, bias_init=bias, num_pwfeat_fc=num_pwfeat_fc)
    net = Gnet(num_classes, class_weights=class_weights)
    net.load_params(params)
    orc = go.GnetOracle(num_classes, num_blocks, params=params, class_weights=class_weights,
                        normalize_loss=normalize_loss, num_pwfeat_fc=num_pwfeat_fc, pw_feat_multiplyer=pw_feat_multiplyer)
    return net, orc


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b))))) if a.size else 0.0


from oracle.pins import grad_errors, gpu_pins  # noqa: E402,F401
