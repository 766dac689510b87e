"""The arithmetic the FC kernels run on: every fp32 product as six bf16 products of exact three-term splits
(csrc/common.hpp split3_pk / mma6 -- edge_fwd_w, pw_fwd2, pw_bwd_main, edge_bwd_w's h1, winners_ties), measured through the
LIBRARY's own primitives (gnet_debug_gemm, csrc/debug.hip) on the real operands of the headline image -- pairwise features, rectified
activations, the network's weight matrices -- against fp64, beside v_mfma_f32_32x32x2_f32 on the same operands.

What is asserted:
  * hi + mid + lo == x EXACTLY for every finite operand value of magnitude >= 2^-100 (and for 0), each term a bf16; below that the
    remainders are fp32 denormals, which the vector ALU flushes: the split is then off by less than 2^-125 in absolute terms;
  * the six-product result is as close to fp64 as the fp32 MFMA's: its worst entry within RATIO (2x) of the fp32 MFMA's worst entry,
    both below 1e-6 of sum |a||b| (measured on MI355X, round 6: 3.5e-7 against 2.6e-7 at K = 32 on the pairwise features, 5.1e-7
    against 5.5e-7 at K = 64 on operands spread over e^+-16 -- the three dropped cross terms all carry the product's sign, a bias
    of ~2^-25 of sum |a||b| that the fp32 chain's roundings do not have);
  * small, negative, zero and mixed-magnitude operands stay within that bound."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RATIO = 2.0


def debug_gemm(a, b, mode, want_terms=False):
    from gossipnet_amd import _lib
    lib = _lib.load()
    a = a.contiguous().float(); b = b.contiguous().float()
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, device=a.device)
    terms = torch.zeros(3, M, K, device=a.device) if want_terms else None
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.gnet_debug_gemm(vp(a), vp(b), M, K, N, mode, vp(c), vp(terms), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "gnet_debug_gemm")
    torch.cuda.synchronize()
    return (c, terms) if want_terms else c


def errors(a, b, c):
    """(max |c - a.b| over the entries, the same relative to the row's sum |a||b|) with the product in fp64."""
    a64, b64 = a.double(), b.double()
    ref = a64 @ b64
    scale = (a64.abs() @ b64.abs()).clamp_min(1e-300)
    d = (c.double() - ref).abs()
    return float(d.max()), float((d / scale).max())


def check_terms(a, terms):
    t = terms.double()
    d = (t[0] + t[1] + t[2] - a.double()).abs()
    big = (a.abs() >= 2.0 ** -100) | (a == 0)
    assert float(d[big].max()) == 0.0 if bool(big.any()) else True, "hi + mid + lo must equal the operand exactly"
    assert float(d.max()) < 2.0 ** -125, "below 2^-100 the remainders are denormals (flushed): off by less than 2^-125"
    bits = terms.view(torch.int32)
    assert int((bits & 0xffff).abs().max()) == 0, "every term is a bf16 (the lower 16 bits of its fp32 image are zero)"
    # magnitudes: mid below 2^-7 of hi's, lo below 2^-15 (two truncations of 8 significant bits each)
    hi, mid, lo = t[0].abs(), t[1].abs(), t[2].abs()
    assert bool((mid <= hi * 2.0 ** -7).all()) and bool((lo <= hi * 2.0 ** -15).all())


def check_product(a, b, what):
    c6, terms = debug_gemm(a, b, 0, want_terms=True)
    c32 = debug_gemm(a, b, 1)
    check_terms(a, terms)
    abs6, rel6 = errors(a, b, c6)
    abs32, rel32 = errors(a, b, c32)
    print("%-34s [%d x %d x %d]  six bf16 products: max err %.3e (%.3e of sum|a||b|)   fp32 MFMA: %.3e (%.3e)"
          % (what, a.shape[0], a.shape[1], b.shape[1], abs6, rel6, abs32, rel32))
    assert rel6 <= 1e-6 and rel32 <= 1e-6, (what, rel6, rel32)
    assert rel6 <= RATIO * rel32 + 1e-8, (what, rel6, rel32)
    return rel6, rel32


def pad_rows(x, mult=32):
    r = (-x.shape[0]) % mult
    return torch.cat([x, torch.zeros(r, x.shape[1], device=x.device, dtype=x.dtype)]) if r else x


def test_six_products_on_the_headline_image_operands():
    """P (pairwise features), a block's rectified pw_fc1 activations and the pw-MLP's rectified h1 of the headline image
    (N = 2000, C = 80, dense preset), times the weight matrices they meet in the kernels."""
    from tests.util import make_pair, make_image
    net, _ = make_pair(80, 2)
    net.keep_edge_activations = True
    net.run(make_image(2000, 80, seed=0))
    torch.cuda.synchronize()
    E = int(net.num_edges)
    dv = net.debug_view
    P = net.pw_feats.detach().clone()                                   # [E,32] >= 0
    h1b = dv("blk_h1", E * 64, index=1).view(E, 64).clone()             # block 1: relu(pw_fc1) [E,64]
    h1p = dv("pw_h1", E * 256).view(E, 256).clone()                     # pw-MLP fc1 activations [E,256]
    v = net.variables
    w1 = v["gnet/block1/pw_fc1/weights"][:32].detach().clone()          # the pairwise rows [32,64]
    w2 = v["gnet/block1/pw_fc2/weights"].detach().clone()               # [64,64]
    pw2 = v["gnet/pw_feats/fc2/weights"].detach().clone()               # [256,256]
    assert E > 100000 and float(P.max()) > 0 and float(h1b.max()) > 0
    check_product(pad_rows(P), w1, "P . pw_fc1[:32] (edge_fwd_w layer 1)")
    check_product(pad_rows(h1b), w2, "relu(h1) . pw_fc2 (edge_fwd_w layer 2)")
    rows = pad_rows(h1p[: 32 * 1024])
    check_product(rows, pw2, "pw h1 . pw_feats/fc2 (pw_fwd2)")
    # the weight-gradient shape: contraction over the edges (h1^T . d) -- K = 4096 rows of real activations
    d = torch.randn(4096, 64, device=h1p.device, generator=torch.Generator(device=h1p.device).manual_seed(1)) * 1e-3
    check_product(h1p[:4096].t().contiguous(), d, "pw h1^T . d (weight-gradient shape)")


@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e3, 1e-20, 1e-36])
def test_six_products_on_synthetic_operands(scale):
    """Signed, zero-rich and mixed-magnitude operands (values over e^+-16 around the scale); at 1e-36 the operands sit at the bottom
    of the fp32 range, where remainders and products are denormals on both pipes: finite results and the absolute split bound only."""
    g = torch.Generator(device="cuda:0").manual_seed(7)
    a = torch.randn(256, 64, device="cuda:0", generator=g)
    a = torch.where(torch.rand(256, 64, device="cuda:0", generator=g) < 0.3, torch.zeros_like(a), a)      # rectified-like zeros
    a = a * torch.exp(4.0 * torch.randn(256, 64, device="cuda:0", generator=g))                              # magnitudes over e^+-8
    b = (torch.rand(64, 64, device="cuda:0", generator=g) - 0.5) * 0.43                                      # xavier-uniform of a 64 x 64 layer
    a = (a * scale).float()
    c6, terms = debug_gemm(a, b, 0, want_terms=True)
    check_terms(a, terms)
    c32 = debug_gemm(a, b, 1)
    abs6, rel6 = errors(a, b, c6)
    abs32, rel32 = errors(a, b, c32)
    print("scale %g: six bf16 products %.3e of sum|a||b| (fp32 MFMA %.3e)" % (scale, rel6, rel32))
    if scale >= 1e-20:
        assert rel6 <= 1e-6 and rel6 <= RATIO * rel32 + 1e-8, (rel6, rel32)
    else:
        # products near the bottom of the fp32 range: the accumulator itself rounds to denormals (both pipes); only finiteness and
        # the exact split are asserted, the measured error is printed
        assert bool(torch.isfinite(c6).all())


def test_debug_gemm_argument_checks():
    from gossipnet_amd import _lib
    a = torch.zeros(32, 16, device="cuda:0"); b = torch.zeros(16, 32, device="cuda:0")
    with pytest.raises(_lib.InvalidArgumentError):
        debug_gemm(torch.zeros(31, 16, device="cuda:0"), b, 0)
    with pytest.raises(_lib.InvalidArgumentError):
        debug_gemm(a, b, 2)
    assert bool((debug_gemm(a, b, 0) == 0).all())
