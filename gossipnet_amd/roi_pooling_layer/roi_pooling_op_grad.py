"""Shape function and gradient registration of RoiPool -- mirror of nms_net/roi_pooling_layer/roi_pooling_op_grad.py:
`_roi_pool_shape` (:7-21, tf.RegisterShape("RoiPool")) and `_roi_pool_grad` (:23-43, ops.RegisterGradient("RoiPool")).
The autograd node below is what `roi_pooling_op.roi_pool` returns tensors from; its backward calls
`roi_pooling_op.roi_pool_grad` and returns [data_grad, None] like the reference's registration."""
import torch

from . import roi_pooling_op


def roi_pool_output_shapes(data_shape, rois_shape, pooled_height, pooled_width):
    """roi_pooling_op_grad.py:7-21: both outputs are [num_rois, pooled_height, pooled_width, channels]."""
    out = (int(rois_shape[0]), int(pooled_height), int(pooled_width), int(data_shape[3]))
    return [out, out]


class RoiPoolFunction(torch.autograd.Function):
    """forward = op "RoiPool" (top_data, argmax); backward = op "RoiPoolGrad" on the data input, no gradient for the
    rois (roi_pooling_op_grad.py:41-43) nor for the attributes."""

    @staticmethod
    def forward(ctx, bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
        top, argmax = roi_pooling_op.roi_pool_raw(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale)
        ctx.save_for_backward(bottom_data, bottom_rois, argmax)
        ctx.attrs = (pooled_height, pooled_width, spatial_scale)
        ctx.mark_non_differentiable(argmax)
        return top, argmax

    @staticmethod
    def backward(ctx, grad_top, _grad_argmax):
        data, rois, argmax = ctx.saved_tensors
        ph, pw, sc = ctx.attrs
        data_grad = roi_pooling_op.roi_pool_grad(data, rois, argmax, grad_top, ph, pw, sc)
        return data_grad, None, None, None, None
