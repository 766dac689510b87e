// Numerics probe of the product arithmetic the FC kernels use (tests only: tests/test_gpu_bf16x3.py).
// gnet_debug_gemm multiplies two fp32 matrices with the LIBRARY's own primitives of common.hpp -- split3_pk / split3_8 / mma6
// (every fp32 product as six bf16 products of exact three-term splits, v_mfma_f32_32x32x16_bf16) or v_mfma_f32_32x32x2_f32 -- so that
// the arithmetic edge_fwd_w, pw_fwd2, pw_bwd_main, edge_bwd_w and winners_ties rest on can be measured on the real operands of a step
// (pairwise features, rectified activations, trained-shape weights) against fp64, next to the exactness of hi + mid + lo = x.
// One wave per 32 x 32 block of the result; nothing here is on the product path.
#include "common.hpp"

namespace {

struct DebugGemmArgs {
  const float* a; const float* b; float* c; float* terms;
  int M, K, N, mode;
};

__global__ void __launch_bounds__(64) debug_gemm(const DebugGemmArgs g) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  f32x16 acc = zero16();
  if (g.mode == 0) {
    // 32x32x16 bf16: lane (r, h) supplies A[m0 + r][k0 + 8 h + 0..7] and B[k0 + 8 h + 0..7][n0 + r]
    for (int k0 = 0; k0 < g.K; k0 += 16) {
      const float* ap = g.a + (size_t)(m0 + r) * g.K + k0 + 8 * h;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
      float bv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bv[t] = g.b[(size_t)(k0 + 8 * h + t) * g.N + n0 + r];
      const Bf3 A = split3_8(a0, a1);
      const Bf3 B = split3_8(f32x4{bv[0], bv[1], bv[2], bv[3]}, f32x4{bv[4], bv[5], bv[6], bv[7]});
      acc = mma6(acc, A, B);
      if (g.terms && blockIdx.x == 0) {
        // the three terms of the lane's eight A values, widened to fp32 (a bf16 is the upper half of an fp32)
        const size_t plane = (size_t)g.M * g.K;
        float* tp = g.terms + (size_t)(m0 + r) * g.K + k0 + 8 * h;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          tp[2 * w] = __uint_as_float(A.h[w] << 16);              tp[2 * w + 1] = __uint_as_float(A.h[w] & 0xffff0000u);
          tp[plane + 2 * w] = __uint_as_float(A.m[w] << 16);      tp[plane + 2 * w + 1] = __uint_as_float(A.m[w] & 0xffff0000u);
          tp[2 * plane + 2 * w] = __uint_as_float(A.l[w] << 16);  tp[2 * plane + 2 * w + 1] = __uint_as_float(A.l[w] & 0xffff0000u);
        }
      }
    }
  } else {
    // 32x32x2 fp32: lane (r, h) supplies A[m0 + r][k + h] and B[k + h][n0 + r]
    for (int k = 0; k < g.K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g.a[(size_t)(m0 + r) * g.K + k + h], g.b[(size_t)(k + h) * g.N + n0 + r], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) g.c[(size_t)(m0 + crow(i, h)) * g.N + n0 + r] = acc[i];
}

}  // namespace

extern "C" int gnet_debug_gemm(const float* a, const float* b, int64_t M, int64_t K, int64_t N, int mode, float* c, float* a_terms,
                               gnet_stream_t stream) {
  clear_hip_error();
  if (!a || !b || !c || M <= 0 || K <= 0 || N <= 0 || (M & 31) || (N & 31) || (K & 15) || (mode != 0 && mode != 1) ||
      M > (1ll << 21) || N > (1ll << 21) || K > (1 << 20))
    return GNET_ERR_INVALID;
  if (((uintptr_t)a & 15) || (K & 3)) return GNET_ERR_INVALID;
  DebugGemmArgs g{a, b, c, mode == 0 ? a_terms : nullptr, (int)M, (int)K, (int)N, mode};
  debug_gemm<<<dim3((unsigned)(N / 32), (unsigned)(M / 32)), 64, 0, (hipStream_t)stream>>>(g);
  return launch_status();
}
