"""Dense FC layers on the HIP library (csrc/fc.hip): the `reduce_imfeats` stack of the image-feature variant
(nms_net/network.py:223-240).  Plain tensors in, plain tensors out; PyTorch only owns the memory."""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class FcWorkspace(object):
    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, lib, M, K, N):
        need = int(lib.gnet_fc_workspace_bytes(M, K, N))
        if self.buf is None or self.buf.numel() < need:
            self.buf = None
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.buf


def fc_forward(x, w, b, relu, ws):
    """y = act(x @ w + b); x [M,K], w [K,N] ([in,out], tf.contrib.layers.fully_connected), b [N]."""
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    buf = ws.get(lib, M, K, N)
    s = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(lib.gnet_fc_forward(_p(x), _p(w), _p(b), M, K, N, int(bool(relu)), _p(y), _p(buf), buf.numel(), s), "gnet_fc_forward")
    return y


def fc_backward(x, w, y, dy, relu, dw, db, ws, need_dx):
    """Gradients of y = act(x @ w + b) given dy; dw / db are written in place (views of the flat gradient buffer)."""
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=x.device) if need_dx else None
    buf = ws.get(lib, M, K, N)
    s = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(lib.gnet_fc_backward(_p(x), _p(w), _p(y), _p(dy), M, K, N, int(bool(relu)), _p(dw), _p(db), _p(dx), _p(buf),
                                    buf.numel(), s), "gnet_fc_backward")
    return dx
