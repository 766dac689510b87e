"""nms_net.roi_pooling_layer.roi_pooling_op_grad (reference roi_pooling_op_grad.py:7-43): importing it makes
`roi_pool` differentiable with respect to its data input (the reference registers the gradient on import)."""
from gossipnet_amd.roi_pooling_layer.roi_pooling_op_grad import RoiPoolFunction, roi_pool_output_shapes  # noqa: F401
