"""nms_net.roi_pooling_layer.roi_pooling_op (reference roi_pooling_op.py:4-7): `roi_pool`, `roi_pool_grad`."""
from gossipnet_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad  # noqa: F401
