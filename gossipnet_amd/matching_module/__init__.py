"""detection_matching -- mirror of nms_net/matching_module/__init__.py:9-13 (op "DetectionMatching",
det_matching.cc:16-33).  Not differentiable (tf.NotDifferentiable, __init__.py:10): the returned
tensors carry no autograd edges.  Runs on the device through det_matching_f32 (no CPU kernel)."""
import ctypes as C

import torch

from .. import _lib

__all__ = 'detection_matching'


def detection_matching(iou, score, ignore):
    """(labels f32 [N], weights f32 [N], assignment i32 [N]) = detection_matching(iou [N,M], score [N], ignore [M]).

    Shape checks follow det_matching.cc:76-93 (InvalidArgument)."""
    if iou.dim() != 2:
        raise _lib.InvalidArgumentError("DetectionMatching expects a 2-D vector as input 1.")
    if score.dim() != 1:
        raise _lib.InvalidArgumentError("DetectionMatching expects a 1-D vector as input 2.")
    if ignore.dim() != 1:
        raise _lib.InvalidArgumentError("DetectionMatching expects a 1-D vector as input 3.")
    if iou.shape[0] != score.shape[0]:
        raise _lib.InvalidArgumentError("DetectionMatching expects dim 1 of input 1 and dim 1 of input 2 to be the same "
                                        "(%d != %d)" % (iou.shape[0], score.shape[0]))
    if iou.shape[1] != ignore.shape[0]:
        raise _lib.InvalidArgumentError("DetectionMatching expects dim 2 of input 1 and dim 1 of input 3 to be the same "
                                        "(%d != %d)" % (iou.shape[1], ignore.shape[0]))
    if not iou.is_cuda:
        raise _lib.GnetError("detection_matching has only a device kernel: pass CUDA/HIP tensors")
    lib = _lib.load()
    dev = iou.device
    iou = iou.detach().contiguous().float()
    score = score.detach().contiguous().float().to(dev)
    ign = ignore.detach().to(dev).to(torch.uint8).contiguous()
    n, m = iou.shape
    labels = torch.empty(n, dtype=torch.float32, device=dev)
    weights = torch.empty(n, dtype=torch.float32, device=dev)
    assign = torch.empty(n, dtype=torch.int32, device=dev)
    nbytes = lib.det_matching_workspace_bytes(n, m)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    p = ws.data_ptr()
    p = (p + 255) // 256 * 256
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.det_matching_f32(C.c_void_p(iou.data_ptr()), C.c_void_p(score.data_ptr()), C.c_void_p(ign.data_ptr()),
                                    n, m, C.c_void_p(labels.data_ptr()), C.c_void_p(weights.data_ptr()),
                                    C.c_void_p(assign.data_ptr()), C.c_void_p(p), nbytes, s), "det_matching_f32")
    return labels, weights, assign
