"""gossipnet_amd -- MI355X-native GossipNet (learned NMS) hot path behind the reference's nms_net API.

    from gossipnet_amd import cfg, Gnet
    from gossipnet_amd.matching_module import detection_matching
    from gossipnet_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad
"""
from .config import cfg, cfg_from_file  # noqa: F401


def __getattr__(name):
    if name == "Gnet":
        from .network import Gnet
        return Gnet
    raise AttributeError(name)
