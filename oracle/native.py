"""ctypes loader for oracle/native_oracle.c -- TEST INFRASTRUCTURE ONLY (see its header)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    src = os.path.join(_HERE, "native_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def det_matching(ious, score, ignore):
    ious = np.ascontiguousarray(ious, np.float32)
    score = np.ascontiguousarray(score, np.float32)
    ign = np.ascontiguousarray(np.asarray(ignore, dtype=bool).astype(np.uint8))
    n, m = score.shape[0], ign.shape[0]
    ious = ious.reshape(n, m)
    labels = np.empty(n, np.float32)
    weights = np.empty(n, np.float32)
    assign = np.empty(n, np.int32)
    lib().oracle_det_matching(_p(ious), _p(score), _p(ign), n, m, _p(labels), _p(weights), _p(assign))
    return labels, weights, assign


def roi_pool(data, rois, ph, pw, scale):
    data = np.ascontiguousarray(data, np.float32)
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 5)
    B, H, W, C = data.shape
    R = rois.shape[0]
    top = np.empty((R, ph, pw, C), np.float32)
    arg = np.empty((R, ph, pw, C), np.int32)
    lib().oracle_roi_pool_fwd(_p(data), B, H, W, C, _p(rois), R, ph, pw, ctypes.c_float(scale), _p(top), _p(arg))
    return top, arg


def roi_pool_grad(data_shape, rois, argmax, grad, ph, pw, scale):
    B, H, W, C = data_shape
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 5)
    argmax = np.ascontiguousarray(argmax, np.int32)
    grad = np.ascontiguousarray(grad, np.float32)
    out = np.empty((B, H, W, C), np.float32)
    lib().oracle_roi_pool_bwd(_p(grad), _p(argmax), _p(rois), B, H, W, C, rois.shape[0], ph, pw,
                              ctypes.c_float(scale), _p(out))
    return out
