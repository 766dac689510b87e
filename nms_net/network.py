"""nms_net.network (reference nms_net/network.py:121-322 `Gnet`, :78-118 the crop helpers)."""
from gossipnet_amd.network import (DeviceBatch, Gnet, crop_windows, enlarge_windows,  # noqa: F401
                                   to_frcn_coords)
