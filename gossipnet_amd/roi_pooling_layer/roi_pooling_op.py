"""roi_pool / roi_pool_grad -- mirror of nms_net/roi_pooling_layer/roi_pooling_op.py:4-7 (ops "RoiPool" and
"RoiPoolGrad", roi_pooling_op.cc:35-54).  NHWC, argmax = index within the image."""
import ctypes as C

import torch

from .. import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _check(bottom_data, bottom_rois, pooled_height, pooled_width):
    if pooled_height < 0:
        raise _lib.InvalidArgumentError("Need pooled_height >= 0, got %d" % pooled_height)   # roi_pooling_op.cc:64-66
    if pooled_width < 0:
        raise _lib.InvalidArgumentError("Need pooled_width >= 0, got %d" % pooled_width)
    if bottom_data.dim() != 4:
        raise _lib.InvalidArgumentError("data must be 4-dimensional")                        # :88-89
    if bottom_rois.dim() != 2:
        raise _lib.InvalidArgumentError("rois must be 2-dimensional")                        # :92-93
    if not bottom_data.is_cuda:
        raise _lib.GnetError("roi_pool has only a device kernel: pass CUDA/HIP tensors")


def roi_pool_raw(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
    _check(bottom_data, bottom_rois, pooled_height, pooled_width)
    lib = _lib.load()
    data = bottom_data.detach().contiguous().float()
    rois = bottom_rois.detach().contiguous().float()
    B, H, W, Cc = data.shape
    R = rois.shape[0]
    top = torch.empty(R, pooled_height, pooled_width, Cc, dtype=torch.float32, device=data.device)
    argmax = torch.empty(R, pooled_height, pooled_width, Cc, dtype=torch.int32, device=data.device)
    _lib.check(lib.roi_pool_fwd_f32(_p(data), B, H, W, Cc, _p(rois), R, pooled_height, pooled_width,
                                    float(spatial_scale), _p(top), _p(argmax), _stream(data.device)), "roi_pool_fwd_f32")
    return top, argmax


def roi_pool_grad(bottom_data, bottom_rois, argmax, grad, pooled_height, pooled_width, spatial_scale, deterministic=True):
    """output = roi_pool_grad(bottom_data, bottom_rois, argmax, grad, ...) (roi_pooling_op.cc:45-54).
    deterministic=True sums in the CPU kernel's order (bit-exact, reproducible); False = float-atomic scatter."""
    _check(bottom_data, bottom_rois, pooled_height, pooled_width)
    if argmax.dim() != 4:
        raise _lib.InvalidArgumentError("argmax_data must be 4-dimensional")                 # :343-344
    if grad.dim() != 4:
        raise _lib.InvalidArgumentError("out_backprop must be 4-dimensional")                # :346-347
    lib = _lib.load()
    B, H, W, Cc = bottom_data.shape
    rois = bottom_rois.detach().contiguous().float()
    am = argmax.detach().contiguous().to(torch.int32)
    g = grad.detach().contiguous().float()
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=bottom_data.device)
    fn = lib.roi_pool_bwd_f32 if deterministic else lib.roi_pool_bwd_atomic_f32
    _lib.check(fn(_p(g), _p(am), _p(rois), B, H, W, Cc, rois.shape[0], pooled_height, pooled_width,
                  float(spatial_scale), _p(out), _stream(out.device)), "roi_pool_bwd_f32")
    return out


def roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
    """(top_data, argmax) = roi_pool(bottom_data [B,H,W,C], bottom_rois [R,5], ...) (roi_pooling_op.cc:35-43).
    Differentiable with respect to bottom_data through the gradient registered in roi_pooling_op_grad
    (the reference registers it when that module is imported, roi_pooling_op_grad.py:23; here it is always bound)."""
    from .roi_pooling_op_grad import RoiPoolFunction
    return RoiPoolFunction.apply(bottom_data, bottom_rois, int(pooled_height), int(pooled_width), float(spatial_scale))
