"""nms_net -- the reference's import surface, served by the MI355X-native implementation.

Callers of the reference do `from nms_net import cfg`, `from nms_net.network import Gnet`,
`from nms_net.config import cfg_from_file` (reference train.py:19-21, test.py) and, inside the network,
`from nms_net.roi_pooling_layer import roi_pooling_op, roi_pooling_op_grad` and `from nms_net import matching_module`
(reference nms_net/network.py:12-14).  Every one of those names resolves here to the object of the same name in
`gossipnet_amd` (HIP kernels behind include/gossipnet_hip.h; no TensorFlow, no CPU fallback): putting this repository
root in front of the reference's on sys.path switches the hot path without touching the caller.

This package holds no logic of its own -- only the names (reference nms_net/__init__.py:2 exports `cfg`).
"""
from gossipnet_amd.config import cfg  # noqa: F401
