"""Shape function of roi_pooling_op_grad.py:7-21; the gradient itself is wired in roi_pooling_op._RoiPool."""


def roi_pool_shape(data_shape, rois_shape, pooled_height, pooled_width):
    out = [rois_shape[0], pooled_height, pooled_width, data_shape[3]]
    return [out, out]
