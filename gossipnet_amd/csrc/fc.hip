// Dense fully-connected layers of the image-feature start features (network.py:223-240 "reduce_imfeats"):
//   det_imfeats = flatten(roifeats) [N, 7*7*C]  ->  FC(-> imfeat_dim, ReLU) (if imfeat_dim > 0)  ->  FC(-> 128, ReLU)
// with their backward (d W, d b, d x).  At N = 2000, C = 1024 the first layer is a [2000, 50176] x [50176, 1024]
// fp32 GEMM (205 GFLOP) -- a real MFMA kernel, not a helper:
//
//   fc_gemm<TA, TB>   C[M, N] = op(A)[M, K] . op(B)[K, N] on v_mfma_f32_32x32x2_f32 (exact fp32).  One workgroup
//                     (4 waves) per 128 x 128 output tile, every wave a 64 x 64 quadrant (4 accumulator tiles),
//                     K in steps of 16 staged through LDS in k-major order (both operand reads of an MFMA are then
//                     conflict-free 4-byte reads: lanes = consecutive rows / columns), next K-step's global loads in
//                     flight during the MFMAs.  TA / TB say whether the operand already lies k-major in memory
//                     (direct 16-byte copies) or is transposed while it is staged.  Split-K (blockIdx.z) writes
//                     partial tiles that fc_finish adds in a fixed order (no float atomics: reproducible).
//   fc_finish         y = act(sum_s partial[s] + b)
//   fc_colsum         d b = column sums of d y (fixed order)
// forward  y = x . W      : A = x [M][K] (transposed while staged), B = W [K][N] (k-major)
// d W = x^T . d y         : A = x [M][K] read as op(A)[K][M] -- k-major (the contraction runs over M), B = d y [M][N] (k-major)
// d x = d y . W^T         : A = d y [M][N] (transposed while staged), B = W [K][N] read as op(B)[N][K] (transposed while staged)
#include "common.hpp"

namespace {

constexpr int FC_T = 128;          // output tile
constexpr int FC_K = 16;           // K step
constexpr int FC_LD = FC_T + 4;    // LDS row (k-major: [FC_K][FC_LD])

// Stage one operand tile [FC_K][FC_T] (k-major in LDS).
//   KMAJOR: memory holds op(X)[k][t] = X[(k0 + k) * ld + t0 + t]: rows of 128 consecutive floats
//   else  : memory holds op(X)[k][t] = X[(t0 + t) * ld + k0 + k]: 16 consecutive floats per t (transposed while staged)
// Elements outside [kmax) x [tmax) are zero.  256 threads, 2 x float4 each.
template <bool KMAJOR>
__device__ __forceinline__ void fc_load(float4 (&r)[2], const float* __restrict__ X, long long ld, long long k0, long long kmax,
                                        long long t0, long long tmax, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;                 // 512 float4 per tile
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KMAJOR) {
      const int k = idx >> 5, t4 = idx & 31;       // 32 float4 per k row
      const long long kk = k0 + k, tt = t0 + 4 * t4;
      if (kk < kmax) {
        const float* p = X + kk * ld + tt;
        if (tt + 3 < tmax) v = *reinterpret_cast<const float4*>(p);
        else { if (tt < tmax) v.x = p[0]; if (tt + 1 < tmax) v.y = p[1]; if (tt + 2 < tmax) v.z = p[2]; }
      }
    } else {
      const int t = idx >> 2, k4 = idx & 3;        // 4 float4 per t row
      const long long tt = t0 + t, kk = k0 + 4 * k4;
      if (tt < tmax) {
        const float* p = X + tt * ld + kk;
        if (kk + 3 < kmax) v = *reinterpret_cast<const float4*>(p);
        else { if (kk < kmax) v.x = p[0]; if (kk + 1 < kmax) v.y = p[1]; if (kk + 2 < kmax) v.z = p[2]; }
      }
    }
    r[i] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void fc_store(float* s, const float4 (&r)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    if (KMAJOR) {
      const int k = idx >> 5, t4 = idx & 31;
      *reinterpret_cast<float4*>(s + k * FC_LD + 4 * t4) = r[i];
    } else {
      const int t = idx >> 2, k4 = idx & 3;
      s[(4 * k4 + 0) * FC_LD + t] = r[i].x; s[(4 * k4 + 1) * FC_LD + t] = r[i].y;
      s[(4 * k4 + 2) * FC_LD + t] = r[i].z; s[(4 * k4 + 3) * FC_LD + t] = r[i].w;
    }
  }
}

struct FcGemmArgs {
  const float* A; const float* B; float* C;      // C: [splits][M][N] partial tiles (splits == 1: the result itself)
  long long M, N, K;
  long long lda, ldb;                            // leading dimensions of A / B as they lie in memory
  long long k_per_split;
};

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ void __launch_bounds__(256, 2) fc_gemm(const FcGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float sA[2][FC_K * FC_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][FC_K * FC_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;                  // quadrant of the 128 x 128 tile
  const long long m0 = (long long)blockIdx.y * FC_T, n0 = (long long)blockIdx.x * FC_T;
  const long long kb = (long long)blockIdx.z * a.k_per_split, ke = min(a.K, kb + a.k_per_split);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  float4 ra[2], rb[2];
  fc_load<A_KMAJOR>(ra, a.A, a.lda, kb, ke, m0, a.M, tid);
  fc_load<B_KMAJOR>(rb, a.B, a.ldb, kb, ke, n0, a.N, tid);
  int it = 0;
  for (long long k0 = kb; k0 < ke; k0 += FC_K, ++it) {
    float* cA = sA[it & 1]; float* cB = sB[it & 1];
    fc_store<A_KMAJOR>(cA, ra, tid);
    fc_store<B_KMAJOR>(cB, rb, tid);
    __syncthreads();                                         // (the other buffer was last read before the previous barrier)
    if (k0 + FC_K < ke) {
      fc_load<A_KMAJOR>(ra, a.A, a.lda, k0 + FC_K, ke, m0, a.M, tid);
      fc_load<B_KMAJOR>(rb, a.B, a.ldb, k0 + FC_K, ke, n0, a.N, tid);
    }
    const float* pa = cA + half * FC_LD + 64 * wm + col;
    const float* pb = cB + half * FC_LD + 64 * wn + col;
#pragma unroll
    for (int s = 0; s < FC_K / 2; ++s) {
      const float a0 = pa[2 * s * FC_LD], a1 = pa[2 * s * FC_LD + 32];
      const float b0 = pb[2 * s * FC_LD], b1 = pb[2 * s * FC_LD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float* C = a.C + (size_t)blockIdx.z * a.M * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + 64 * wm + 32 * i + crow(r, half), n = n0 + 64 * wn + 32 * j + col;
        if (m < a.M && n < a.N) C[(size_t)m * a.N + n] = acc[i][j][r];
      }
}

// y[m][n] = act(sum over the split-K partials (ascending) + b[n])
__global__ void __launch_bounds__(256) fc_finish(const float* __restrict__ part, int splits, long long mn, long long N,
                                                 const float* __restrict__ bias, int relu, float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += part[(size_t)s * mn + i];
  if (bias) v += bias[i % N];
  y[i] = relu ? fmaxf(v, 0.f) : v;
}

// dz = dy * (y > 0) (ReLU of the layer; TF ReluGrad), in place or into dz
__global__ void __launch_bounds__(256) fc_relu_grad(const float* __restrict__ dy, const float* __restrict__ y, long long n,
                                                    float* __restrict__ dz) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// db[n] = sum_m dz[m][n]: one workgroup per 64 columns, rows split over 4 waves, folded in wave order
__global__ void __launch_bounds__(256) fc_colsum(const float* __restrict__ dz, long long M, long long N, float* __restrict__ db) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n = (long long)blockIdx.x * 64 + lane;
  float v = 0.f;
  if (n < N) for (long long m = wave; m < M; m += 4) v += dz[(size_t)m * N + n];
  red[wave][lane] = v;
  __syncthreads();
  if (wave == 0 && n < N) db[n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

int pick_splits(long long M, long long N, long long K) {
  const long long tiles = ((M + FC_T - 1) / FC_T) * ((N + FC_T - 1) / FC_T);
  long long s = 1;
  while (tiles * s < 512 && K / (s * 2) >= 8 * FC_K && s < 64) s *= 2;
  return (int)s;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" size_t gnet_fc_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  // forward: split-K partials of y; backward: dz [M][N] + split-K partials of d W (none: M is the contraction) / d x
  const size_t fwd = (size_t)pick_splits(M, N, K) * (size_t)M * (size_t)N;
  const int sw = pick_splits(K, N, M);
  const size_t bwd = (size_t)M * (size_t)N + (sw > 1 ? (size_t)sw * (size_t)K * (size_t)N : 0);
  return (fwd > bwd ? fwd : bwd) * sizeof(float) + 256;
}

extern "C" int gnet_fc_forward(const float* x, const float* w, const float* b, int64_t M, int64_t K, int64_t N, int relu,
                               float* y, void* workspace, size_t workspace_bytes, gnet_stream_t stream) {
  clear_hip_error();
  if (!x || !w || !y || M < 0 || K <= 0 || N <= 0) return GNET_ERR_INVALID;
  if ((K & 3) || (N & 3)) return GNET_ERR_UNSUPPORTED;      // 16-byte operand loads
  if (M == 0) return GNET_OK;
  if (workspace_bytes < gnet_fc_workspace_bytes(M, K, N) || !workspace) return GNET_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int splits = pick_splits(M, N, K);
  FcGemmArgs g;
  g.A = x; g.B = w; g.C = (float*)workspace; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = N;
  g.k_per_split = ((K + splits - 1) / splits + FC_K - 1) / FC_K * FC_K;
  const dim3 grid((unsigned)((N + FC_T - 1) / FC_T), (unsigned)((M + FC_T - 1) / FC_T), (unsigned)splits);
  fc_gemm<false, true><<<grid, 256, 0, s>>>(g);
  fc_finish<<<(unsigned)((M * N + 255) / 256), 256, 0, s>>>((const float*)workspace, splits, M * N, N, b, relu, y);
  return launch_status();
}

// dy: gradient wrt the layer output y (post-activation).  dw [K][N], db [N] are overwritten; dx [M][K] (may be NULL).
extern "C" int gnet_fc_backward(const float* x, const float* w, const float* y, const float* dy, int64_t M, int64_t K, int64_t N,
                                int relu, float* dw, float* db, float* dx, void* workspace, size_t workspace_bytes,
                                gnet_stream_t stream) {
  clear_hip_error();
  if (!x || !w || !y || !dy || !dw || !db || M < 0 || K <= 0 || N <= 0) return GNET_ERR_INVALID;
  if ((K & 3) || (N & 3)) return GNET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    HIP_CHECK_RET(hipMemsetAsync(dw, 0, (size_t)K * N * sizeof(float), s));
    HIP_CHECK_RET(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s));
    return GNET_OK;
  }
  if (workspace_bytes < gnet_fc_workspace_bytes(M, K, N) || !workspace) return GNET_ERR_WORKSPACE;
  float* dz = (float*)workspace;
  float* part = dz + (size_t)M * N;
  if (relu) fc_relu_grad<<<(unsigned)((M * N + 255) / 256), 256, 0, s>>>(dy, y, M * N, dz);
  const float* g_out = relu ? dz : dy;
  fc_colsum<<<(unsigned)((N + 63) / 64), 256, 0, s>>>(g_out, M, N, db);
  {
    // d W[k][n] = sum_m x[m][k] * dz[m][n]: op(A)[k][m] lies k... the contraction index m is the ROW of x: k-major
    const int splits = pick_splits(K, N, M);
    FcGemmArgs g;
    g.A = x; g.B = g_out; g.C = splits == 1 ? dw : part; g.M = K; g.N = N; g.K = M; g.lda = K; g.ldb = N;
    g.k_per_split = ((M + splits - 1) / splits + FC_K - 1) / FC_K * FC_K;
    const dim3 grid((unsigned)((N + FC_T - 1) / FC_T), (unsigned)((K + FC_T - 1) / FC_T), (unsigned)splits);
    fc_gemm<true, true><<<grid, 256, 0, s>>>(g);
    if (splits > 1) fc_finish<<<(unsigned)((K * N + 255) / 256), 256, 0, s>>>(part, splits, K * N, N, nullptr, 0, dw);
  }
  if (dx) {
    // d x[m][k] = sum_n dz[m][n] * w[k][n]: A = dz (transposed while staged), op(B)[n][k] = w[k][n] (transposed while staged)
    FcGemmArgs g;
    g.A = g_out; g.B = w; g.C = dx; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = N;
    g.k_per_split = (N + FC_K - 1) / FC_K * FC_K;
    const dim3 grid((unsigned)((K + FC_T - 1) / FC_T), (unsigned)((M + FC_T - 1) / FC_T), 1);
    fc_gemm<false, false><<<grid, 256, 0, s>>>(g);
  }
  return launch_status();
}
