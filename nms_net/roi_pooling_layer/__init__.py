"""nms_net.roi_pooling_layer (reference nms_net/roi_pooling_layer/: roi_pooling_op, roi_pooling_op_grad)."""
