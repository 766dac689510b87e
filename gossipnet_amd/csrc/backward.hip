// Backward pass of Gnet: the gradient of network.py:197-313 w.r.t. every trainable variable
// (what TF autodiff produces for train.py:64-77), hand-written for gfx950.
//
// Semantics restated from TensorFlow: ReLU grad = g * (out > 0);
// SegmentMax grad splits evenly among ties: sel = (h2 == p[c]), dh2 = sel ? (dp / cnt)[c] : 0;
// gather grads are segment sums (centre, sorted) and scatter-adds (neighbour).  No gradient flows
// into the geometry features (stop_gradient, network.py:454), the matching, or the boxes.
//
// Kernels (launch order):
//   head_bwd       predict/logits, fc2, fc1                       -> d_x (grad wrt block_feats[B])
//   per block b = B..1:
//     blk_bwd_post  shortcut ReLU, fc2, fc1                       -> d_x := dz, d_pc = dp / tie count
//     edge stage, on the edges that attain a segment maximum only (default):
//       winners_mark     arg-max record of the forward pass -> per-edge column masks (ties resolved)
//       edge_bwd_sparse  pw_fc2, pw_fc1 on 64-row tiles of winners  -> d_pw (+=), d_g1 (winner rows)
//       gather_sparse    centre / reversed-edge sums of those rows   -> d_rc, d_rn
//     or on every edge (GNET_DENSE_BWD=1): edge_bwd (recomputes pw_fc2) + gather_sums
//     blk_bwd_pre   per-node halves of pw_fc1, reduce_dim         -> d_x := dz + drpre . Wr^T
//   rowlist_*      ascending list of the edges with a non-zero d_pw row (winners of any block)
//   pw_bwd_main    pw_feats fc3, fc2 (+ d_h1 = grad wrt fc1 pre-activation), on the listed rows
//   pw_w1_nodesums + pw_w1_classrows   pw_feats fc1 (score columns via per-detection sums, 7 geometry rows)
//   reduce_partials  sums the per-workgroup partial weight gradients in a fixed order
// Weight gradients are accumulated in MFMA accumulators across a workgroup's tiles and written once
// per workgroup to an arena; reduce_partials adds them in index order.  There are no float atomics:
// every sum has a fixed order, so gradients are reproducible run to run.
#include <type_traits>
#include "common.hpp"

namespace {

constexpr int LD32 = D_E + 4;    // 36
constexpr int LD64 = D_P + 4;    // 68
constexpr int LD128 = D_S + 4;   // 132
constexpr int LD256 = D_H + 4;   // 260

template <int W, int LD>
__device__ __forceinline__ void load_tile(float* s, const float* __restrict__ g, long long row0, long long nrows,
                                          int tid, int nthreads) {
  constexpr int W4 = W / 4;
  for (int i = tid; i < 32 * W4; i += nthreads) {
    const int row = i / W4, c4 = i - row * W4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows) v = *reinterpret_cast<const float4*>(g + (size_t)(row0 + row) * W + 4 * c4);
    *reinterpret_cast<float4*>(s + row * LD + 4 * c4) = v;
  }
}

__device__ __forceinline__ float col_sum32(const float* s, int ld, int col) {
  float v = 0.f;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) v += s[r * ld + col];
  return v;
}

// write one 32x32 accumulator tile (C layout) to a row-major destination
__device__ __forceinline__ void store_acc(float* dst, int ld, const f32x16& acc, int lane) {
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[(size_t)crow(r, half) * ld + col] = acc[r];
}

// ------------------------------------------------------------------------------------------
struct HeadBwdArgs {
  int n_det;
  const float* d_logits; const float* head2; const float* head1; const float* xb;
  const float* hw1; const float* hw2; const float* hwl;   // natural layouts
  float* d_x;
  float* arena; long long stride;                          // arena row stride = total params
  long long o_hw1, o_hb1, o_hw2, o_hb2, o_hwl, o_hbl;
};

__global__ void __launch_bounds__(256) head_bwd(const HeadBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sH2[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sH1[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sX[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sD2[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sD1[32 * LD128];
  __shared__ float sDl[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  f32x16 aW2[1][4], aW1[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { aW2[0][j] = zero16(); aW1[0][j] = zero16(); }
  float gwl = 0.f, gb2 = 0.f, gb1 = 0.f, gbl = 0.f;
  const int ntiles = (a.n_det + 31) / 32;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row0 = t * 32;
    __syncthreads();
    load_tile<D_HEAD, LD128>(sH2, a.head2, row0, a.n_det, tid, 256);
    load_tile<D_HEAD, LD128>(sH1, a.head1, row0, a.n_det, tid, 256);
    load_tile<D_S, LD128>(sX, a.xb, row0, a.n_det, tid, 256);
    if (tid < 32) sDl[tid] = (row0 + tid < a.n_det) ? a.d_logits[row0 + tid] : 0.f;
    __syncthreads();
    // d head2 = dl (outer) wl
    for (int i = tid; i < 32 * D_HEAD; i += 256) {
      const int row = i >> 7, j = i & 127;
      sD2[row * LD128 + j] = sDl[row] * a.hwl[j];
    }
    if (tid < D_HEAD) {
      float v = 0.f;
      for (int r = 0; r < 32; ++r) v = fmaf(sH2[r * LD128 + tid], sDl[r], v);
      gwl += v;
    }
    if (tid == 0) { float v = 0.f; for (int r = 0; r < 32; ++r) v += sDl[r]; gbl += v; }
    __syncthreads();
    mma_xty<1, 4>(aW2, sH1 + 32 * wave, LD128, sD2, LD128, lane);           // d W(fc2) += head1^T . d head2
    if (tid < D_HEAD) gb2 += col_sum32(sD2, LD128, tid);
    {
      f32x16 acc = zero16();                                                // d head1 = d head2 . W2^T
      mma_abt<D_HEAD>(acc, sD2, LD128, a.hw2 + (size_t)(32 * wave) * D_HEAD, D_HEAD, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) sD1[crow(r, half) * LD128 + 32 * wave + col] = acc[r];
    }
    __syncthreads();
    mma_xty<1, 4>(aW1, sX + 32 * wave, LD128, sD1, LD128, lane);            // d W(fc1) += x^T . d head1
    if (tid < D_HEAD) gb1 += col_sum32(sD1, LD128, tid);
    {
      f32x16 acc = zero16();                                                // d x = d head1 . W1^T
      mma_abt<D_HEAD>(acc, sD1, LD128, a.hw1 + (size_t)(32 * wave) * D_HEAD, D_HEAD, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int node = row0 + crow(r, half);
        if (node < a.n_det) a.d_x[(size_t)node * D_S + 32 * wave + col] = acc[r];
      }
    }
  }
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    store_acc(ar + a.o_hw2 + (size_t)(32 * wave) * D_HEAD + 32 * j, D_HEAD, aW2[0][j], lane);
    store_acc(ar + a.o_hw1 + (size_t)(32 * wave) * D_HEAD + 32 * j, D_HEAD, aW1[0][j], lane);
  }
  if (tid < D_HEAD) { ar[a.o_hwl + tid] = gwl; ar[a.o_hb2 + tid] = gb2; ar[a.o_hb1 + tid] = gb1; }
  if (tid == 0) ar[a.o_hbl] = gbl;
}

// ------------------------------------------------------------------------------------------
struct BlkPostArgs {
  int n_det;
  float* d_x;                       // in: grad wrt block output; out: dz = d_x * (x_out > 0)
  const float* x_out; const float* q; const unsigned long long* pm;
  const float* w4; const float* w3; // natural [64,128], [64,64]
  float* d_pc;
  float* arena; long long stride;
  long long o_w4, o_b4, o_w3, o_b3;
};

__global__ void __launch_bounds__(256) blk_bwd_post(const BlkPostArgs a) {
  __shared__ __attribute__((aligned(16))) float sDz[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sQ[32 * LD64];
  __shared__ __attribute__((aligned(16))) float sP[32 * LD64];
  __shared__ __attribute__((aligned(16))) float sDq[32 * LD64];
  __shared__ __attribute__((aligned(16))) float sR[2 * 32 * D_P];   // K-half partials
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  f32x16 aW4[2][1], aW3[1][1];
  aW4[0][0] = zero16(); aW4[1][0] = zero16(); aW3[0][0] = zero16();
  float gb4 = 0.f, gb3 = 0.f;
  const int ntiles = (a.n_det + 31) / 32;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row0 = t * 32;
    __syncthreads();
    for (int i = tid; i < 32 * (D_S / 4); i += 256) {
      const int row = i >> 5, c4 = i & 31;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < a.n_det) {
        const size_t o = (size_t)(row0 + row) * D_S + 4 * c4;
        const float4 g = *reinterpret_cast<const float4*>(a.d_x + o);
        const float4 x = *reinterpret_cast<const float4*>(a.x_out + o);
        v = make_float4(x.x > 0.f ? g.x : 0.f, x.y > 0.f ? g.y : 0.f, x.z > 0.f ? g.z : 0.f, x.w > 0.f ? g.w : 0.f);
        *reinterpret_cast<float4*>(a.d_x + o) = v;      // dz: also the shortcut gradient
      }
      *reinterpret_cast<float4*>(sDz + row * LD128 + 4 * c4) = v;
    }
    load_tile<D_P, LD64>(sQ, a.q, row0, a.n_det, tid, 256);
    for (int i = tid; i < 32 * D_P; i += 256) {
      const int row = i >> 6, ff = i & 63;
      float v = 0.f;
      if (row0 + row < a.n_det) {
        const size_t o = (size_t)(row0 + row) * D_P + ff;
        v = __uint_as_float((unsigned)(a.pm[o] >> 32));
      }
      sP[row * LD64 + ff] = v;
    }
    __syncthreads();
    // d W4 += q^T . dz : wave w owns output column tile w
    {
      f32x16 (&acc)[2][1] = aW4;
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + h;
        const float x0 = sQ[row * LD64 + r], x1 = sQ[row * LD64 + 32 + r];
        const float y = sDz[row * LD128 + 32 * wave + r];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y, acc[1][0], 0, 0, 0);
      }
    }
    if (tid < D_S) gb4 += col_sum32(sDz, LD128, tid);
    // dq = (dz . W4^T) * (q > 0): wave = (column tile, K half)
    {
      const int nt = wave & 1, kh = wave >> 1;
      f32x16 acc = zero16();
      mma_abt<64>(acc, sDz + 64 * kh, LD128, a.w4 + (size_t)(32 * nt) * D_S + 64 * kh, D_S, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[(kh * 32 + crow(r, half)) * D_P + 32 * nt + col] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * D_P; i += 256) {
      const int row = i >> 6, ff = i & 63;
      const float v = sR[row * D_P + ff] + sR[(32 + row) * D_P + ff];
      sDq[row * LD64 + ff] = sQ[row * LD64 + ff] > 0.f ? v : 0.f;
    }
    __syncthreads();
    // d W3 += p^T . dq : wave = (mi, nj)
    {
      const int mi = wave >> 1, nj = wave & 1;
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + h;
        aW3[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sP[row * LD64 + 32 * mi + r], sDq[row * LD64 + 32 * nj + r],
                                                         aW3[0][0], 0, 0, 0);
      }
    }
    if (tid < D_P) gb3 += col_sum32(sDq, LD64, tid);
    // dp = dq . W3^T : wave = (column tile, K half)
    {
      const int nt = wave & 1, kh = wave >> 1;
      f32x16 acc = zero16();
      mma_abt<32>(acc, sDq + 32 * kh, LD64, a.w3 + (size_t)(32 * nt) * D_P + 32 * kh, D_P, lane);
      __syncthreads();   // sR of the dq step fully consumed
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[(kh * 32 + crow(r, half)) * D_P + 32 * nt + col] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * D_P; i += 256) {
      const int row = i >> 6, ff = i & 63;
      if (row0 + row < a.n_det) {
        const size_t o = (size_t)(row0 + row) * D_P + ff;
        const float dp = sR[row * D_P + ff] + sR[(32 + row) * D_P + ff];
        const unsigned cnt = (unsigned)(a.pm[o] & 0xffffffffull);
        a.d_pc[o] = dp / (float)cnt;                     // weighted_grads = grad / num_selected
      }
    }
  }
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  store_acc(ar + a.o_w4 + 32 * wave, D_S, aW4[0][0], lane);
  store_acc(ar + a.o_w4 + (size_t)32 * D_S + 32 * wave, D_S, aW4[1][0], lane);
  store_acc(ar + a.o_w3 + (size_t)(32 * (wave >> 1)) * D_P + 32 * (wave & 1), D_P, aW3[0][0], lane);
  if (tid < D_S) ar[a.o_b4 + tid] = gb4;
  if (tid < D_P) ar[a.o_b3 + tid] = gb3;
}

// ------------------------------------------------------------------------------------------
// Gather gradients (TF: segment sum for the sorted centre gather, scatter-add for the neighbour
// gather) without atomics:  d_rc[i] = sum_{e in row i} g1[e];  d_rn[i] = sum_{e in row i, n != i}
// g1[reverse(e)] (the graph is symmetric).  One wave per detection; a wave-instruction reads four
// 256-byte edge rows (16 lanes x float4 each); fixed summation order.  HBM/L2-bound: 2 x 256 B per edge.
__global__ void __launch_bounds__(256) gather_sums(const float* __restrict__ g1, const int* __restrict__ row_ptr,
                                                   const int* __restrict__ edge_n, const int* __restrict__ edge_t,
                                                   int n_det, float* __restrict__ d_rc, float* __restrict__ d_rn) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_det) return;
  const int lane = threadIdx.x & 63, sub = lane >> 4, f4 = lane & 15;
  const int eb = row_ptr[node], ee = row_ptr[node + 1];
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sn = sc;
  // 64 edges per pass: one coalesced load of the reversed-pair positions, then 16 independent row pairs per
  // quarter-wave (no index -> row dependent chain per edge; the row loads of a pass are all in flight together)
  for (int base = eb; base < ee; base += 64) {
    const int el = base + lane;
    int tt = -1;
    if (el < ee) tt = edge_n[el] != node ? edge_t[el] : -1;   // self pair: n_feats zeroed (network.py:371-374)
    const int cnt = min(64, ee - base);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int j = 4 * i + sub;                              // edge base + j, ascending per quarter-wave
      if (4 * i >= cnt) break;                                // wave-uniform
      const int t = __shfl(tt, j);
      if (j < cnt) {
        const float4 c = *reinterpret_cast<const float4*>(g1 + (size_t)(base + j) * D_P + 4 * f4);
        sc.x += c.x; sc.y += c.y; sc.z += c.z; sc.w += c.w;
        if (t >= 0) {
          const float4 v = *reinterpret_cast<const float4*>(g1 + (size_t)t * D_P + 4 * f4);
          sn.x += v.x; sn.y += v.y; sn.z += v.z; sn.w += v.w;
        }
      }
    }
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    sc.x += __shfl_xor(sc.x, o); sc.y += __shfl_xor(sc.y, o); sc.z += __shfl_xor(sc.z, o); sc.w += __shfl_xor(sc.w, o);
    sn.x += __shfl_xor(sn.x, o); sn.y += __shfl_xor(sn.y, o); sn.z += __shfl_xor(sn.z, o); sn.w += __shfl_xor(sn.w, o);
  }
  if (sub == 0) {
    *reinterpret_cast<float4*>(d_rc + (size_t)node * D_P + 4 * f4) = sc;
    *reinterpret_cast<float4*>(d_rn + (size_t)node * D_P + 4 * f4) = sn;
  }
}

struct BlkPreArgs {
  int n_det;
  int write_dx;                     // block > 1: d_x += drpre . Wr^T
  const float* d_rc; const float* d_rn; const float* r; const float* x_prev;   // x_prev NULL = zeros
  const float* w1;                  // natural [96,64]; rows 32-63 centre, 64-95 neighbour
  const float* wr;                  // natural [128,32]
  float* d_x;
  float* arena; long long stride;
  long long o_w1, o_b1, o_wr, o_br;
};

__global__ void __launch_bounds__(256) blk_bwd_pre(const BlkPreArgs a) {
  __shared__ __attribute__((aligned(16))) float sRc[32 * LD64];
  __shared__ __attribute__((aligned(16))) float sRn[32 * LD64];
  __shared__ __attribute__((aligned(16))) float sRr[32 * LD32];
  __shared__ __attribute__((aligned(16))) float sDr[32 * LD32];
  __shared__ __attribute__((aligned(16))) float sX[32 * LD128];
  __shared__ __attribute__((aligned(16))) float sPart[4 * 32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  f32x16 aWcn = zero16(), aWr = zero16();
  float gb1 = 0.f, gbr = 0.f;
  const int ntiles = (a.n_det + 31) / 32;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row0 = t * 32;
    __syncthreads();
    load_tile<D_P, LD64>(sRc, a.d_rc, row0, a.n_det, tid, 256);
    load_tile<D_P, LD64>(sRn, a.d_rn, row0, a.n_det, tid, 256);
    load_tile<D_R, LD32>(sRr, a.r, row0, a.n_det, tid, 256);
    if (a.x_prev) load_tile<D_S, LD128>(sX, a.x_prev, row0, a.n_det, tid, 256);
    else for (int i = tid; i < 32 * LD128; i += 256) sX[i] = 0.f;
    __syncthreads();
    // dr = drc . Wc^T + drn . Wn^T : wave = (term, K half)
    {
      const int term = wave & 1, kh = wave >> 1;
      f32x16 acc = zero16();
      mma_abt<32>(acc, (term ? sRn : sRc) + 32 * kh, LD64, a.w1 + (size_t)(32 + 32 * term) * D_P + 32 * kh, D_P, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) sPart[(wave * 32 + crow(r, half)) * 32 + col] = acc[r];
    }
    // d Wc += r^T . drc ; d Wn += r^T . drn : wave = (term, column tile)
    {
      const int term = wave >> 1, nj = wave & 1;
      const float* Y = term ? sRn : sRc;
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + h;
        aWcn = __builtin_amdgcn_mfma_f32_32x32x2f32(sRr[row * LD32 + r], Y[row * LD64 + 32 * nj + r], aWcn, 0, 0, 0);
      }
    }
    if (tid < D_P) gb1 += col_sum32(sRc, LD64, tid);
    __syncthreads();
    for (int i = tid; i < 32 * D_R; i += 256) {
      const int row = i >> 5, ff = i & 31;
      float v = sPart[(0 * 32 + row) * 32 + ff] + sPart[(1 * 32 + row) * 32 + ff];
      v += sPart[(2 * 32 + row) * 32 + ff];
      v += sPart[(3 * 32 + row) * 32 + ff];
      sDr[row * LD32 + ff] = sRr[row * LD32 + ff] > 0.f ? v : 0.f;    // ReLU of reduce_dim
    }
    __syncthreads();
    // d Wr += x_prev^T . drpre : wave w owns rows [32w, 32w+32) of Wr
    {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + h;
        aWr = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[row * LD128 + 32 * wave + r], sDr[row * LD32 + r], aWr, 0, 0, 0);
      }
    }
    if (tid < D_R) gbr += col_sum32(sDr, LD32, tid);
    if (a.write_dx) {
      f32x16 acc = zero16();
      mma_abt<D_R>(acc, sDr, LD32, a.wr + (size_t)(32 * wave) * D_R, D_R, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int node = row0 + crow(r, half);
        if (node < a.n_det) a.d_x[(size_t)node * D_S + 32 * wave + col] += acc[r];
      }
    }
  }
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  store_acc(ar + a.o_w1 + (size_t)(32 + 32 * (wave >> 1)) * D_P + 32 * (wave & 1), D_P, aWcn, lane);
  store_acc(ar + a.o_wr + (size_t)(32 * wave) * D_R, D_R, aWr, lane);
  if (tid < D_P) ar[a.o_b1 + tid] = gb1;
  if (tid < D_R) ar[a.o_br + tid] = gbr;
}

// ------------------------------------------------------------------------------------------
struct EdgeBwdArgs {
  int n_edge; int n_det;
  const int* edge_c;
  const float* pw;                  // [E,32] pairwise features P
  const float* h1;                  // [E+64,64] relu(pw_fc1) stored by the forward pass
  const unsigned long long* pm; const float* d_pc;
  const float* w1t; const float* w2t; const float* b2;   // transposed copies (w2t: recompute of pw_fc2)
  const float* w1; const float* w2;                       // natural layouts (input gradients)
  float* d_pw; float* d_g1;
  float* arena; long long stride;
  long long o_w1, o_w2, o_b2;
};

// edge_bwd: one workgroup (4 waves) per 64-edge tile; wave (mt, nt) owns edge rows [32mt, 32mt+32)
// and feature columns [32nt, 32nt+32) of every 64-wide tensor (h2, d h2, g1).  Per tile and wave:
// 128 MFMAs (32 L2 + 32 dW2 + 32 g1 + 16 dWp + 16 dP), two LDS tiles shared by the workgroup.
// h1 = relu(pw_fc1) is NOT recomputed: the forward pass kept it in HBM (256 B per edge and block), so this
// kernel has no rc/rn gathers and no layer-1 MFMAs; it streams the h1 and P tiles one tile ahead through
// registers.  Only pw_fc2 is recomputed (its output is compared with the stored segment maxima).
// Memory discipline (the L1 stalls on repeated requests to a line that is still in flight):
//   * centre-side gathers (segment max, tie-split gradient) are issued once per DISTINCT centre --
//     a detection's ~E/N consecutive edges share them;
//   * the old d_pw values of the read-modify-write are prefetched before the tile's stores.
// 70.6 KB of LDS -> 2 independent workgroups per CU.
constexpr int EB_T = 64;

__global__ void __launch_bounds__(256, 2) edge_bwd(const EdgeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sWpT = smem;                     // [64][36]  Wp^T  (layer 1; also the B operand of d P)
  float* sW2T = sWpT + D_P * LD32;        // [64][68]  W2^T  (layer 2; also the B operand of g1, strided)
  float* sA = sW2T + D_P * LD64;          // [64][68]  h1, later g1
  float* sB = sA + EB_T * LD64;           // [64][68]  d h2, later the K-split partials of d P
  float* sP = sB + EB_T * LD64;           // [64][36]  P tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: mt, nt and every tile offset stay in SGPRs
  for (int i = tid; i < D_P * D_E; i += 256) sWpT[(i >> 5) * LD32 + (i & 31)] = a.w1t[(i >> 5) * (D_E + 2 * D_R) + (i & 31)];
  for (int i = tid; i < D_P * D_P; i += 256) sW2T[(i >> 6) * LD64 + (i & 63)] = a.w2t[i];
  const int col = lane & 31, half = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  const float bias = a.b2[32 * nt + col];
  const unsigned* pmw = reinterpret_cast<const unsigned*>(a.pm);   // [N][64] x {count, max bits}
  f32x16 aW2 = zero16(), aWp = zero16();    // dW2 tile (mt, nt); dWp columns nt, edge half mt
  float gb2 = 0.f;
  const int ntiles = (a.n_edge + EB_T - 1) / EB_T;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(ntiles, t0 + per);
  // one-tile-ahead prefetch: indices of this wave's rows and this thread's two float4 of the P tile
  int nx_c = -1;
  float4 pf0 = make_float4(0.f, 0.f, 0.f, 0.f), pf1 = pf0;
  float4 hf0 = pf0, hf1 = pf0, hf2 = pf0, hf3 = pf0;               // h1 tile: thread -> (row tid>>4 + 16 i, float4 tid&15)
  const int prow0 = tid >> 3, pc4 = tid & 7;                        // P tile: thread -> (row, float4) x 2
  const int hrow0 = tid >> 4, hc4 = tid & 15;
#define EB_LOAD_TILES(tile_)                                                                           \
  do {                                                                                                 \
    const int last_ = a.n_edge - 1;                                                                    \
    pf0 = *reinterpret_cast<const float4*>(a.pw + (size_t)min((tile_) * EB_T + prow0, last_) * D_E + 4 * pc4);      \
    pf1 = *reinterpret_cast<const float4*>(a.pw + (size_t)min((tile_) * EB_T + 32 + prow0, last_) * D_E + 4 * pc4); \
    /* rows past the end re-read the last real row: finite data (their d h2 rows are zero) */          \
    hf0 = *reinterpret_cast<const float4*>(a.h1 + (size_t)min((tile_) * EB_T + hrow0, last_) * D_P + 4 * hc4);       \
    hf1 = *reinterpret_cast<const float4*>(a.h1 + (size_t)min((tile_) * EB_T + hrow0 + 16, last_) * D_P + 4 * hc4);  \
    hf2 = *reinterpret_cast<const float4*>(a.h1 + (size_t)min((tile_) * EB_T + hrow0 + 32, last_) * D_P + 4 * hc4);  \
    hf3 = *reinterpret_cast<const float4*>(a.h1 + (size_t)min((tile_) * EB_T + hrow0 + 48, last_) * D_P + 4 * hc4);  \
  } while (0)
  if (t0 < t1) {
    const int e = t0 * EB_T + 32 * mt + col;
    if (e < a.n_edge) nx_c = a.edge_c[e];
    EB_LOAD_TILES(t0);
  }
  // centre-side values of the first two segments (A = first centre of this wave's 32 rows, B = the next
  // distinct centre or the same): fetched unconditionally, wave-uniformly, one tile ahead
  int cA = -1, cB = -1, hiA = 32;
  float pmA = 0.f, dpA = 0.f, pmB = 0.f, dpB = 0.f;
  const unsigned lane_b = (unsigned)(32 * nt + col) * 4u;           // byte offset of this lane's column in a 64-float row
#define EB_PREFETCH_NEXT()                                                                             \
  do {                                                                                                 \
    cA = __builtin_amdgcn_readfirstlane(nx_c); cB = cA; hiA = 32;                                      \
    const int prev_ = __shfl_up(nx_c, 1);                                                              \
    const unsigned hm_ = (unsigned)__ballot(half == 0 && col > 0 && nx_c != prev_ && nx_c >= 0);       \
    if (hm_) { hiA = __builtin_ctz(hm_); cB = __builtin_amdgcn_readlane(nx_c, hiA); }                  \
    const unsigned oa_ = (unsigned)max(cA, 0) * (D_P * 4u) + lane_b, ob_ = (unsigned)max(cB, 0) * (D_P * 4u) + lane_b; \
    pmA = __uint_as_float(ldg_b(pmw, 2 * oa_ + 4)); dpA = ldg_b(a.d_pc, oa_);                          \
    pmB = __uint_as_float(ldg_b(pmw, 2 * ob_ + 4)); dpB = ldg_b(a.d_pc, ob_);                          \
  } while (0)
  EB_PREFETCH_NEXT();
  drain_vmem_before_loop();
  __syncthreads();
  // The tile body exists twice: FULL tiles (all 64 edges exist) run in the loop with unconditional loads and
  // stores, the one possibly partial tile of the launch runs after it.  Conditional (exec-masked or
  // branched-around) memory operations inside the loop would shrink the guaranteed number of operations
  // behind the tile prefetch to almost zero, and the compiler's wait for the prefetch would become a wait
  // for the previous tile's stores (vmcnt is one in-order counter; see drain_vmem_before_loop).
  auto tile_body = [&](const int t, auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const int e0 = t * EB_T + 32 * mt;                              // first edge of this wave's rows
    const int my_c = nx_c;
    const int nrows = FULL ? 32 : min(32, a.n_edge - e0);           // may be <= 0 for the last tile
    const int thiA = hiA;
    const float tpmA = pmA, tdpA = dpA, tpmB = pmB, tdpB = dpB;
    int nseg;
    {
      const int prev = __shfl_up(my_c, 1);
      nseg = __popcll(__ballot(half == 0 && col < nrows && (col == 0 || my_c != prev)));
    }
    const bool simple = nseg <= 2;                                  // wave-uniform
    float* sAp = sA + (32 * mt + 4 * half) * LD64 + 32 * nt + col;   // + crow(r, 0) * LD64 = row crow(r, half)
    float* sBp = sB + (32 * mt + 4 * half) * LD64 + 32 * nt + col;
    const int tvalid = FULL ? EB_T * D_E : min(EB_T, a.n_edge - t * EB_T) * D_E;   // valid floats of the d_pw tile
    float* dpw_tile = a.d_pw + (size_t)(t * EB_T) * D_E;
    *reinterpret_cast<float4*>(sP + prow0 * LD32 + 4 * pc4) = pf0;
    *reinterpret_cast<float4*>(sP + (32 + prow0) * LD32 + 4 * pc4) = pf1;
    *reinterpret_cast<float4*>(sA + hrow0 * LD64 + 4 * hc4) = hf0;
    *reinterpret_cast<float4*>(sA + (hrow0 + 16) * LD64 + 4 * hc4) = hf1;
    *reinterpret_cast<float4*>(sA + (hrow0 + 32) * LD64 + 4 * hc4) = hf2;
    *reinterpret_cast<float4*>(sA + (hrow0 + 48) * LD64 + 4 * hc4) = hf3;
    nx_c = -1;
    if (t + 1 < t1) {
      const int e = e0 + EB_T + col;
      if (e < a.n_edge) nx_c = a.edge_c[e];
      EB_LOAD_TILES(t + 1);
    }
    __syncthreads();                                                // B1: P and h1 tiles in LDS
    // ---- S2: h2 = relu(h1 . W2 + b2); d h2 = SegmentMax tie split + ReLU mask
    f32x16 d2 = zero16();
    mma_abt<D_P>(d2, sA + 32 * mt * LD64, LD64, sW2T + 32 * nt * LD64, LD64, lane);
    if (FULL && nseg == 1) {
      // relu(v) == max > 0  <=>  v == max: one compare per element against the maximum (NaN when the
      // maximum is 0: nothing passes the ReLU then)
      const float pmq = tpmA > 0.f ? tpmA : __builtin_nanf("");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = (d2[r] + bias == pmq) ? tdpA : 0.f;
        gb2 += x;
        sBp[crow(r, 0) * LD64] = x;
      }
    } else if (simple) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(d2[r] + bias, 0.f);
        const bool inA = crow(r, half) < thiA;
        const float pmr = inA ? tpmA : tpmB, dpr = inA ? tdpA : tdpB;
        const float x = (crow(r, half) < nrows && v > 0.f && v == pmr) ? dpr : 0.f;
        gb2 += x;
        sBp[crow(r, 0) * LD64] = x;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(d2[r] + bias, 0.f);
        const unsigned oc = (unsigned)max(row_bcast(my_c, r, half), 0) * D_P + 32 * nt + col;
        const float pmr = __uint_as_float(pmw[2 * oc + 1]), dpr = a.d_pc[oc];
        const float x = (crow(r, half) < nrows && v > 0.f && v == pmr) ? dpr : 0.f;
        gb2 += x;
        sBp[crow(r, 0) * LD64] = x;
      }
    }
    __syncthreads();                                                // B2: d h2 tile complete
    // gathers of the NEXT tile (its indices arrived long ago): consumed at the top of the next iteration
    EB_PREFETCH_NEXT();
    float h1[16];                                                   // this lane's h1 values: the ReLU mask of g1
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = sAp[crow(r, 0) * LD64];
    // ---- S3: d W2[mt-th row tile][nt-th column tile] += h1^T . d h2 over the 64 edges
    {
      const float* X = sA + 32 * mt + col;
      const float* Y = sB + 32 * nt + col;
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const int row = 2 * kk + half;
        aW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[row * LD64], Y[row * LD64], aW2, 0, 0, 0);
      }
    }
    // ---- S4: g1 = (d h2 . W2^T) * (h1 > 0);  B[k = h2 f][n = h1 f] = W2[h1 f][h2 f] = sW2T[h2 f][h1 f]
    f32x16 g1 = zero16();
    {
      const float* ap = sB + (32 * mt + col) * LD64 + 4 * half;
      const float* bp = sW2T + (4 * half) * LD64 + 32 * nt + col;
#pragma unroll
      for (int k = 0; k < D_P; k += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[(k + 0) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[(k + 1) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[(k + 2) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[(k + 3) * LD64], g1, 0, 0, 0);
      }
    }
    __syncthreads();                                                // B3: every read of h1 (sA) is done
    float dold[8];                                                  // old d_pw values of the final RMW
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      dold[i] = (FULL || idx < tvalid) ? dpw_tile[idx] : 0.f;     // d_pw is zeroed once per step: always +=
    }
    {
      // g1 goes to HBM: gather_sums turns it into the centre / neighbour sums in a fixed order.
      // row crow(r, half) = crow(r, 0) + 4 half: per-lane base + compile-time row offsets
      float* g1t = a.d_g1 + (size_t)e0 * D_P;                       // uniform
      const unsigned g1o = (unsigned)(4 * half) * (D_P * 4u) + lane_b;
      if (FULL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = h1[r] > 0.f ? g1[r] : 0.f;
          sAp[crow(r, 0) * LD64] = v;
          stg_b(g1t, g1o + crow(r, 0) * (D_P * 4u), v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = h1[r] > 0.f ? g1[r] : 0.f;
          sAp[crow(r, 0) * LD64] = v;
          if (crow(r, half) < nrows) stg_b(g1t, g1o + crow(r, 0) * (D_P * 4u), v);
        }
      }
    }
    __syncthreads();                                                // B4: g1 tile complete
    // ---- S6: d Wp[:, column tile nt] += P^T . g1 over edge half mt
    {
      const float* X = sP + (32 * mt) * LD32 + col;
      const float* Y = sA + (32 * mt) * LD64 + 32 * nt + col;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        aWp = __builtin_amdgcn_mfma_f32_32x32x2f32(X[(2 * kk + half) * LD32], Y[(2 * kk + half) * LD64], aWp, 0, 0, 0);
    }
    // ---- S7: d P[rows mt] = g1 . Wp^T, K split over nt; partials through sB (d h2 is consumed)
    {
      f32x16 acc = zero16();
      const float* ap = sA + (32 * mt + col) * LD64 + 32 * nt + 4 * half;
      const float* bp = sWpT + (32 * nt + 4 * half) * LD32 + col;   // Wp[pf = col][f] = sWpT[f][pf]
#pragma unroll
      for (int k = 0; k < 32; k += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[(k + 0) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[(k + 1) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[(k + 2) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[(k + 3) * LD32], acc, 0, 0, 0);
      }
      float* part = sB + nt * (EB_T * D_E);                          // [2][64][32]
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(32 * mt + crow(r, half)) * D_E + col] = acc[r];
    }
    __syncthreads();                                                // B5: partials complete; sA, sP free
    if (FULL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        dpw_tile[idx] = dold[i] + (sB[idx] + sB[EB_T * D_E + idx]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        if (idx < tvalid) dpw_tile[idx] = dold[i] + (sB[idx] + sB[EB_T * D_E + idx]);
      }
    }
    // (the next tile's S2 writes sB only behind its B1, i.e. after every wave finished this tile)
  };
  const int t_full = max(t0, min(t1, a.n_edge / EB_T));             // tiles [t0, t_full) are complete
  for (int t = t0; t < t_full; ++t) tile_body(t, std::true_type{});
  if (t_full < t1) tile_body(t_full, std::false_type{});            // at most one partial tile per launch
  // ---- partial weight gradients of this workgroup
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  store_acc(ar + a.o_w2 + (size_t)(32 * mt) * D_P + 32 * nt, D_P, aW2, lane);
  __syncthreads();
  float* red = sA;
  if (mt == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[nt * 1024 + r * 64 + lane] = aWp[r];
  }
  red[2048 + wave * 64 + lane] = gb2;
  __syncthreads();
  if (mt == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) aWp[r] += red[nt * 1024 + r * 64 + lane];
    store_acc(ar + a.o_w1 + 32 * nt, D_P, aWp, lane);               // rows 0-31 of pw_fc1
  }
  if (tid < D_P) {
    // column tid: waves (mt = 0,1; nt = tid >> 5), both half-waves
    const int n2 = tid >> 5, c2 = tid & 31;
    float v = 0.f;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) v += red[2048 + (2 * m2 + n2) * 64 + c2] + red[2048 + (2 * m2 + n2) * 64 + 32 + c2];
    ar[a.o_b2 + tid] = v;
  }
}

constexpr size_t kEdgeBwdSmem = (size_t)(D_P * LD32 + D_P * LD64 + 2 * EB_T * LD64 + EB_T * LD32) * sizeof(float);

// ------------------------------------------------------------------------------------------
// Sparse SegmentMax backward.  The gradient of a segment max reaches, per (detection, column), only the edge
// that attained the maximum: d h2 has at most 64 non-zeros per DETECTION, spread over W <= 64 "winner" edges
// (measured: 23-28 % of the edges at E/N = 86, 42-51 % at E/N = 36).  Rows of d h2 that are zero give zero rows
// of g1 and d P and contribute nothing to d W2 / d Wp, so the whole edge stage of the backward pass runs on
// the winner rows only -- same sums, fewer zero terms.  The forward pass recorded the first winner of every
// (detection, column) (blk_parg); a positive maximum attained by several edges (about one per block: fp32
// coincidences) is resolved in winners_mark by recomputing that detection's pw_fc2 bit-exactly.  The dense
// kernels above remain selectable (GNET_DENSE_BWD=1) as the reference implementation of the same stage.
//
//   winners_mark      emask[e] = columns for which edge e is the arg-max (0: no gradient through e)
//   edge_bwd_sparse   compacts its edge range's winners into 64-row tiles: d h2 from (emask, d_pc), h1 / P
//                     rows gathered, dW2 / g1 / dWp / dP as in the dense kernel (96 MFMAs per wave and tile),
//                     d_pw[e] += dP and d_g1[e] = g1 on winner rows only
//   gather_sparse     d_rc / d_rn from the winner rows of d_g1
// One launch for all blocks (blockIdx.y = block - 1): the masks depend on the forward pass only, so they are
// off the backward pass's critical chain (16 dependent ~25 us launches otherwise).
struct WinArgs {
  int n_det;
  long long emask_stride;           // words between the blocks' mask arrays
  unsigned long long* emask;        // [B][emask_stride], zeroed
  const int* row_ptr;
  const unsigned long long* pm[GNET_MAX_BLOCKS];     // [N,64] (max bits << 32) | tie count
  const unsigned long long* parg[GNET_MAX_BLOCKS];   // [N,64] (max bits << 32) | first edge attaining it
  // tie resolution only (a positive maximum attained by 2+ edges; about one (detection, column) per block):
  const float* h1[GNET_MAX_BLOCKS]; const float* w2t[GNET_MAX_BLOCKS]; const float* b2[GNET_MAX_BLOCKS];
};

__global__ void __launch_bounds__(256) winners_mark(const WinArgs a) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const int nwaves = gridDim.x * 4;
  const int blk = blockIdx.y;
  unsigned long long* emask = a.emask + (size_t)blk * a.emask_stride;
  const float* h1p = a.h1[blk]; const float* w2tp = a.w2t[blk]; const float* b2p = a.b2[blk];
  for (int node = blockIdx.x * 4 + (threadIdx.x >> 6); node < a.n_det; node += nwaves) {
    const unsigned long long pv = a.parg[blk][(size_t)node * D_P + lane];
    const unsigned long long pc = a.pm[blk][(size_t)node * D_P + lane];
    const bool valid = (pv >> 32) != 0ull;                  // maximum > 0: the ReLU passes the gradient
    const int arg = (int)(unsigned)pv;
    const unsigned long long ties = __ballot(valid && (unsigned)pc > 1u);
    unsigned long long todo = __ballot(valid);
    unsigned long long mine = 0ull;                         // set on the first lane of every distinct winner edge
    while (todo) {                                          // one iteration per distinct winner edge
      const int k = __builtin_ctzll(todo);
      const int ak = __builtin_amdgcn_readlane(arg, k);
      const unsigned long long m = __ballot(valid && arg == ak);
      mine = lane == k ? m : mine;
      todo &= ~m;
    }
    if (mine != 0ull) {                                     // one store instruction per detection
      if (ties) atomicOr(emask + arg, mine);                // (the tie pass below ORs into the same words)
      else emask[arg] = mine;                               // every edge belongs to exactly one detection
    }
    if (ties) {
      // Rare: some column's maximum is attained by several edges, and the forward pass kept only the first.
      // Recompute pw_fc2 for this detection's edges with the forward kernel's exact MFMA sequence (operand
      // fragments and k order of edge_fwd_w, so the bits match) and mark every edge that attains a tied
      // maximum; d_pc already carries the 1 / count split (network.py:383-386, TF SegmentMax gradient).
      const int eb = a.row_ptr[node], ee = a.row_ptr[node + 1];
      const float bias0 = b2p[col], bias1 = b2p[32 + col];
      const float mx = __uint_as_float((unsigned)(pv >> 32));
      for (int e0 = eb; e0 < ee; e0 += 32) {
        const int nrows = min(32, ee - e0);
        const float* ap = h1p + (size_t)min(e0 + col, ee - 1) * D_P + 4 * half;
        const float* b0 = w2tp + (size_t)col * D_P + 4 * half;
        const float* b1 = b0 + 32 * D_P;
        f32x16 h2a = zero16(), h2b = zero16();
#pragma unroll 4
        for (int k = 0; k < D_P; k += 8) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k);
          const f32x4 bv0 = *reinterpret_cast<const f32x4*>(b0 + k);
          const f32x4 bv1 = *reinterpret_cast<const f32x4*>(b1 + k);
          h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv0.x, h2a, 0, 0, 0);
          h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv1.x, h2b, 0, 0, 0);
          h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv0.y, h2a, 0, 0, 0);
          h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv1.y, h2b, 0, 0, 0);
          h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv0.z, h2a, 0, 0, 0);
          h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv1.z, h2b, 0, 0, 0);
          h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv0.w, h2a, 0, 0, 0);
          h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv1.w, h2b, 0, 0, 0);
        }
        unsigned long long tleft = ties;
        while (tleft) {
          const int j = __builtin_ctzll(tleft);
          tleft &= tleft - 1;
          const float mj = __shfl(mx, j);                   // lane j holds column j's maximum
          if (col == (j & 31)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = j < 32 ? h2a[r] + bias0 : h2b[r] + bias1;
              if (crow(r, half) < nrows && v == mj) atomicOr(emask + e0 + crow(r, half), 1ull << j);
            }
          }
        }
      }
    }
  }
}

struct EdgeBwdSparseArgs {
  int n_edge; int n_det;
  const int* edge_c;
  const unsigned long long* emask;
  unsigned long long* ewin;         // [E/64] out: bit e = edge e is a winner (for gather_sparse; by-product of the scan)
  unsigned long long* eany;         // [E/64] |= ewin over the blocks: edges that carry any gradient into the pw-MLP
  const float* pw; const float* h1; const float* d_pc;
  const float* w1t; const float* w2t;
  float* d_pw; float* d_g1;
  float* arena; long long stride;
  long long o_w1, o_w2, o_b2;
};

constexpr int EBS_RING = 512;
constexpr size_t kEdgeBwdSparseSmem = kEdgeBwdSmem + (size_t)EBS_RING * (2 * sizeof(int) + sizeof(unsigned long long)) + 64;

__global__ void __launch_bounds__(256, 2) edge_bwd_sparse(const EdgeBwdSparseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sWpT = smem;                     // [64][36]  Wp^T (B operand of d P)
  float* sW2T = sWpT + D_P * LD32;        // [64][68]  W2^T (B operand of g1, strided)
  float* sA = sW2T + D_P * LD64;          // [64][68]  h1 rows of the winners, later g1
  float* sB = sA + EB_T * LD64;           // [64][68]  d h2, later the K-split partials of d P
  float* sP = sB + EB_T * LD64;           // [64][36]  P rows of the winners
  unsigned long long* sLm = reinterpret_cast<unsigned long long*>(sP + EB_T * LD32);   // ring: winner masks
  int* sLe = reinterpret_cast<int*>(sLm + EBS_RING);                                    // ring: winner edges
  int* sLc = sLe + EBS_RING;                                                            // ring: their centres
  int* sWc = sLc + EBS_RING;                                                            // [4] per-wave counts
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  f32x16 aW2 = zero16(), aWp = zero16();
  float gb2 = 0.f;                         // column (tid & 63), rows [16 wave, 16 wave + 16) of every d h2 tile
  for (int i = tid; i < D_P * D_E; i += 256) sWpT[(i >> 5) * LD32 + (i & 31)] = a.w1t[(i >> 5) * (D_E + 2 * D_R) + (i & 31)];
  for (int i = tid; i < D_P * D_P; i += 256) sW2T[(i >> 6) * LD64 + (i & 63)] = a.w2t[i];
  // this workgroup scans the 256-edge chunks blockIdx.x, blockIdx.x + gridDim.x, ... (one mask per thread and
  // scan step).  Round-robin chunks: the winner density varies from image to image (measured 462..1080
  // winners per contiguous range), the interleave gives every workgroup the same mix -- statically, so the
  // summation order of the weight gradients stays reproducible.
  const int r1 = a.n_edge;
  const int pstep = 256 * (int)gridDim.x;
  const unsigned lane_b = (unsigned)(32 * nt + col) * 4u;
  constexpr int RM = EBS_RING - 1;
  // thread roles of the staging loads
  const int hrow0 = tid >> 4, hc4 = tid & 15;      // h1 rows hrow0 + 16 i, one float4 each
  const int prow0 = tid >> 3, pc4 = tid & 7;       // P rows prow0 + 32 i (also the d_pw rows of the final update)
  const int drow = tid >> 2, dq = tid & 3;         // d h2 row drow, columns [16 dq, 16 dq + 16)
  // scan state: masks / centres of the next 256 edges are requested one step ahead
  int pos = 256 * (int)blockIdx.x;
  unsigned long long m_pf = 0ull; int c_pf = 0;
#define EBS_PREFETCH_SCAN()                                                                             \
  do {                                                                                                  \
    const int e_ = pos + tid;                                                                           \
    m_pf = e_ < r1 ? a.emask[e_] : 0ull;                                                                \
    c_pf = e_ < r1 ? a.edge_c[e_] : 0;                                                                  \
  } while (0)
  // append the winners of the prefetched 256-edge chunk to the ring at [wbase + wcnt, ...)
#define EBS_SCAN_STEP(wbase, wcnt)                                                                      \
  do {                                                                                                  \
    const unsigned long long m_ = m_pf; const int c_ = c_pf; const int e_ = pos + tid;                  \
    pos += pstep;                                                                                       \
    if (pos < r1) EBS_PREFETCH_SCAN();                                                                  \
    const unsigned long long bm_ = __ballot(m_ != 0ull);                                                \
    if (lane == 0) {                                                                                    \
      sWc[wave] = __popcll(bm_); a.ewin[(e_ >> 6)] = bm_;                                               \
      if (bm_) __hip_atomic_fetch_or(a.eany + (e_ >> 6), bm_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }                                                                                                   \
    __syncthreads();                                                                                    \
    int base_ = (wcnt), total_ = 0;                                                                     \
    _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) { const int n_ = sWc[w_]; base_ += w_ < wave ? n_ : 0; total_ += n_; } \
    if (m_ != 0ull) {                                                                                   \
      const int idx_ = ((wbase) + base_ + __popcll(bm_ & ((1ull << lane) - 1ull))) & RM;                \
      sLe[idx_] = e_; sLm[idx_] = m_; sLc[idx_] = c_;                                                   \
    }                                                                                                   \
    (wcnt) += total_;                                                                                   \
    __syncthreads();                                                                                    \
  } while (0)
  // staging registers of one tile (h1 rows, P rows, the d_pc rows its d h2 is cut from, old d_pw values)
  float4 rh0, rh1, rh2, rh3, rp0, rp1, rd0, rd1, rd2, rd3, ro0, ro1;
  unsigned rbits = 0;
#define EBS_ISSUE_LOADS(tb, nt_)                                                                        \
  do {                                                                                                  \
    const int e0_ = sLe[((tb) + (hrow0 < (nt_) ? hrow0 : 0)) & RM];                                     \
    const int e1_ = sLe[((tb) + (hrow0 + 16 < (nt_) ? hrow0 + 16 : 0)) & RM];                           \
    const int e2_ = sLe[((tb) + (hrow0 + 32 < (nt_) ? hrow0 + 32 : 0)) & RM];                           \
    const int e3_ = sLe[((tb) + (hrow0 + 48 < (nt_) ? hrow0 + 48 : 0)) & RM];                           \
    rh0 = ldg4_b(a.h1, (unsigned)e0_ * (D_P * 4u) + 16u * hc4);                                         \
    rh1 = ldg4_b(a.h1, (unsigned)e1_ * (D_P * 4u) + 16u * hc4);                                         \
    rh2 = ldg4_b(a.h1, (unsigned)e2_ * (D_P * 4u) + 16u * hc4);                                         \
    rh3 = ldg4_b(a.h1, (unsigned)e3_ * (D_P * 4u) + 16u * hc4);                                         \
    const int q0_ = sLe[((tb) + (prow0 < (nt_) ? prow0 : 0)) & RM];                                     \
    const int q1_ = sLe[((tb) + (prow0 + 32 < (nt_) ? prow0 + 32 : 0)) & RM];                           \
    rp0 = ldg4_b(a.pw, (unsigned)q0_ * (D_E * 4u) + 16u * pc4);                                         \
    rp1 = ldg4_b(a.pw, (unsigned)q1_ * (D_E * 4u) + 16u * pc4);                                         \
    ro0 = ldg4_b(a.d_pw, (unsigned)q0_ * (D_E * 4u) + 16u * pc4);                                       \
    ro1 = ldg4_b(a.d_pw, (unsigned)q1_ * (D_E * 4u) + 16u * pc4);                                       \
    const int li_ = ((tb) + (drow < (nt_) ? drow : 0)) & RM;                                            \
    rbits = drow < (nt_) ? (unsigned)(sLm[li_] >> (16 * dq)) & 0xffffu : 0u;                            \
    const unsigned do_ = (unsigned)sLc[li_] * (D_P * 4u) + 64u * dq;                                    \
    rd0 = ldg4_b(a.d_pc, do_); rd1 = ldg4_b(a.d_pc, do_ + 16u);                                         \
    rd2 = ldg4_b(a.d_pc, do_ + 32u); rd3 = ldg4_b(a.d_pc, do_ + 48u);                                   \
  } while (0)
#define EBS_SEL4(v_, k_)                                                                                \
  make_float4((rbits >> (4 * (k_) + 0)) & 1u ? (v_).x : 0.f, (rbits >> (4 * (k_) + 1)) & 1u ? (v_).y : 0.f, \
              (rbits >> (4 * (k_) + 2)) & 1u ? (v_).z : 0.f, (rbits >> (4 * (k_) + 3)) & 1u ? (v_).w : 0.f)

  int head = 0, ntile = 0;                 // current tile: ring [head, head + ntile)
  int wcnt = 0;                            // winners collected beyond the current tile
  if (pos < r1) EBS_PREFETCH_SCAN();
  __syncthreads();
  while (wcnt < EB_T && pos < r1) EBS_SCAN_STEP(head, wcnt);
  ntile = min(wcnt, EB_T); wcnt -= ntile;
  if (ntile > 0) EBS_ISSUE_LOADS(head, ntile);
  while (ntile > 0) {
    // ---- stage the tile from the registers requested during the previous tile
    *reinterpret_cast<float4*>(sA + hrow0 * LD64 + 4 * hc4) = rh0;
    *reinterpret_cast<float4*>(sA + (hrow0 + 16) * LD64 + 4 * hc4) = rh1;
    *reinterpret_cast<float4*>(sA + (hrow0 + 32) * LD64 + 4 * hc4) = rh2;
    *reinterpret_cast<float4*>(sA + (hrow0 + 48) * LD64 + 4 * hc4) = rh3;
    *reinterpret_cast<float4*>(sP + prow0 * LD32 + 4 * pc4) = rp0;
    *reinterpret_cast<float4*>(sP + (prow0 + 32) * LD32 + 4 * pc4) = rp1;
    *reinterpret_cast<float4*>(sB + drow * LD64 + 16 * dq) = EBS_SEL4(rd0, 0);
    *reinterpret_cast<float4*>(sB + drow * LD64 + 16 * dq + 4) = EBS_SEL4(rd1, 1);
    *reinterpret_cast<float4*>(sB + drow * LD64 + 16 * dq + 8) = EBS_SEL4(rd2, 2);
    *reinterpret_cast<float4*>(sB + drow * LD64 + 16 * dq + 12) = EBS_SEL4(rd3, 3);
    const float4 old0 = ro0, old1 = ro1;   // d_pw rows of THIS tile (the next tile's overwrite ro0 / ro1 below)
    __syncthreads();                                                // tiles complete
    // ---- next tile: collect its winners and request its rows (hidden under this tile's MFMAs)
    const int nhead = (head + ntile) & RM;
    while (wcnt < EB_T && pos < r1) EBS_SCAN_STEP(nhead, wcnt);
    const int nnext = min(wcnt, EB_T);
    if (nnext > 0) EBS_ISSUE_LOADS(nhead, nnext);
    float* sAp = sA + (32 * mt + 4 * half) * LD64 + 32 * nt + col;   // + crow(r, 0) * LD64 = row crow(r, half)
    float h1[16];                                                   // this lane's h1 values: the ReLU mask of g1
    unsigned go[16];                                                // byte offsets of this lane's g1 rows in d_g1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h1[r] = sAp[crow(r, 0) * LD64];
      const int row = 32 * mt + crow(r, half);                      // rows past the tile go to the slack row E
      go[r] = (unsigned)(row < ntile ? sLe[(head + row) & RM] : a.n_edge) * (D_P * 4u) + lane_b;
    }
    {                                                               // bias gradient: column sums of d h2
      const float* cp = sB + (16 * wave) * LD64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) gb2 += cp[r * LD64];
    }
    // ---- d W2[mt-th row tile][nt-th column tile] += h1^T . d h2 over the 64 rows
    {
      const float* X = sA + 32 * mt + col;
      const float* Y = sB + 32 * nt + col;
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const int row = 2 * kk + half;
        aW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(X[row * LD64], Y[row * LD64], aW2, 0, 0, 0);
      }
    }
    // ---- g1 = (d h2 . W2^T) * (h1 > 0)
    f32x16 g1 = zero16();
    {
      const float* ap = sB + (32 * mt + col) * LD64 + 4 * half;
      const float* bp = sW2T + (4 * half) * LD64 + 32 * nt + col;
#pragma unroll
      for (int k = 0; k < D_P; k += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[(k + 0) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[(k + 1) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[(k + 2) * LD64], g1, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[(k + 3) * LD64], g1, 0, 0, 0);
      }
    }
    __syncthreads();                                                // every read of h1 (sA) is done
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = h1[r] > 0.f ? g1[r] : 0.f;
      sAp[crow(r, 0) * LD64] = v;
      stg_b(a.d_g1, go[r], v);
    }
    __syncthreads();                                                // g1 tile complete
    // ---- d Wp[:, column tile nt] += P^T . g1 over row half mt
    {
      const float* X = sP + (32 * mt) * LD32 + col;
      const float* Y = sA + (32 * mt) * LD64 + 32 * nt + col;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        aWp = __builtin_amdgcn_mfma_f32_32x32x2f32(X[(2 * kk + half) * LD32], Y[(2 * kk + half) * LD64], aWp, 0, 0, 0);
    }
    // ---- d P[rows mt] = g1 . Wp^T, K split over nt; partials through sB (d h2 is consumed)
    {
      f32x16 acc = zero16();
      const float* ap = sA + (32 * mt + col) * LD64 + 32 * nt + 4 * half;
      const float* bp = sWpT + (32 * nt + 4 * half) * LD32 + col;
#pragma unroll
      for (int k = 0; k < 32; k += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bp[(k + 0) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bp[(k + 1) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bp[(k + 2) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bp[(k + 3) * LD32], acc, 0, 0, 0);
      }
      float* part = sB + nt * (EB_T * D_E);                          // [2][64][32]
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(32 * mt + crow(r, half)) * D_E + col] = acc[r];
    }
    __syncthreads();                                                // partials complete; sA, sP free
    // d_pw[e] += d P on the winner rows (an edge is a winner row of exactly one tile per block); this
    // thread owns the float4 (row prow0 [+ 32], pc4) it fetched with the tile
    {
      const float4 x0 = *reinterpret_cast<const float4*>(sB + prow0 * D_E + 4 * pc4);
      const float4 y0 = *reinterpret_cast<const float4*>(sB + EB_T * D_E + prow0 * D_E + 4 * pc4);
      const float4 x1 = *reinterpret_cast<const float4*>(sB + (prow0 + 32) * D_E + 4 * pc4);
      const float4 y1 = *reinterpret_cast<const float4*>(sB + EB_T * D_E + (prow0 + 32) * D_E + 4 * pc4);
      if (prow0 < ntile)
        *reinterpret_cast<float4*>(a.d_pw + (size_t)sLe[(head + prow0) & RM] * D_E + 4 * pc4) =
            make_float4(old0.x + (x0.x + y0.x), old0.y + (x0.y + y0.y), old0.z + (x0.z + y0.z), old0.w + (x0.w + y0.w));
      if (prow0 + 32 < ntile)
        *reinterpret_cast<float4*>(a.d_pw + (size_t)sLe[(head + prow0 + 32) & RM] * D_E + 4 * pc4) =
            make_float4(old1.x + (x1.x + y1.x), old1.y + (x1.y + y1.y), old1.z + (x1.z + y1.z), old1.w + (x1.w + y1.w));
    }
    head = nhead; ntile = nnext; wcnt -= nnext;
    __syncthreads();                                                // sA / sB / sP reusable
  }
#undef EBS_PREFETCH_SCAN
#undef EBS_SCAN_STEP
#undef EBS_ISSUE_LOADS
#undef EBS_SEL4
  // ---- partial weight gradients of this workgroup
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  store_acc(ar + a.o_w2 + (size_t)(32 * mt) * D_P + 32 * nt, D_P, aW2, lane);
  __syncthreads();
  float* red = sA;
  if (mt == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[nt * 1024 + r * 64 + lane] = aWp[r];
  }
  red[2048 + tid] = gb2;
  __syncthreads();
  if (mt == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) aWp[r] += red[nt * 1024 + r * 64 + lane];
    store_acc(ar + a.o_w1 + 32 * nt, D_P, aWp, lane);               // rows 0-31 of pw_fc1
  }
  if (tid < D_P) ar[a.o_b2 + tid] = (red[2048 + tid] + red[2048 + 64 + tid]) + (red[2048 + 128 + tid] + red[2048 + 192 + tid]);
}

// d_rc[i] = sum over i's winner edges of g1[e];  d_rn[i] = sum over i's edges e = (i, n), n != i, of
// g1[reverse(e)] when the reversed pair is a winner of n.  Only winner rows of d_g1 are valid.
__global__ void __launch_bounds__(256) gather_sparse(const float* __restrict__ g1, const int* __restrict__ row_ptr,
                                                     const int* __restrict__ edge_n, const int* __restrict__ edge_t,
                                                     const unsigned long long* __restrict__ ewin,
                                                     int n_det, float* __restrict__ d_rc, float* __restrict__ d_rn) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_det) return;
  const int lane = threadIdx.x & 63, sub = lane >> 4, f4 = lane & 15;
  const int eb = row_ptr[node], ee = row_ptr[node + 1];
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sn = sc;
  for (int base = eb; base < ee; base += 64) {
    const int el = base + lane;
    int tt = -1; bool own = false;
    if (el < ee) {
      own = (ewin[el >> 6] >> (el & 63)) & 1ull;            // the 1-bit-per-edge winner map stays in L2 (E / 8 bytes)
      if (edge_n[el] != node) {                               // self pair: n_feats zeroed (network.py:371-374)
        const int t = edge_t[el];
        tt = (ewin[t >> 6] >> (t & 63)) & 1ull ? t : -1;
      }
    }
    unsigned long long mo = __ballot(own), mr = __ballot(tt >= 0);
    // winner rows only, four at a time (one per quarter-wave), ascending edge order
    while (mo) {
      int j = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) { if (mo) { const int b = __builtin_ctzll(mo); mo &= mo - 1; if (q == sub) j = b; } }
      if (j >= 0) {
        const float4 c = *reinterpret_cast<const float4*>(g1 + (size_t)(base + j) * D_P + 4 * f4);
        sc.x += c.x; sc.y += c.y; sc.z += c.z; sc.w += c.w;
      }
    }
    while (mr) {
      int j = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) { if (mr) { const int b = __builtin_ctzll(mr); mr &= mr - 1; if (q == sub) j = b; } }
      const int t = __shfl(tt, j < 0 ? 0 : j);
      if (j >= 0) {
        const float4 v = *reinterpret_cast<const float4*>(g1 + (size_t)t * D_P + 4 * f4);
        sn.x += v.x; sn.y += v.y; sn.z += v.z; sn.w += v.w;
      }
    }
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    sc.x += __shfl_xor(sc.x, o); sc.y += __shfl_xor(sc.y, o); sc.z += __shfl_xor(sc.z, o); sc.w += __shfl_xor(sc.w, o);
    sn.x += __shfl_xor(sn.x, o); sn.y += __shfl_xor(sn.y, o); sn.z += __shfl_xor(sn.z, o); sn.w += __shfl_xor(sn.w, o);
  }
  if (sub == 0) {
    *reinterpret_cast<float4*>(d_rc + (size_t)node * D_P + 4 * f4) = sc;
    *reinterpret_cast<float4*>(d_rn + (size_t)node * D_P + 4 * f4) = sn;
  }
}

// ------------------------------------------------------------------------------------------
// Rows of the pw-MLP backward: only edges that were a winner row in at least one block have a non-zero
// d_pw row (72 % of the edges at E/N = 86); the others contribute exact zeros to every sum.  rowlist_* turn
// the bitmap into an ascending list (count -> exclusive_scan -> fill: deterministic order).
__device__ __forceinline__ unsigned long long rowlist_word(const unsigned long long* __restrict__ bits, int w, int n_words, int n_edge) {
  if (w >= n_words) return 0ull;
  unsigned long long b = bits[w];
  if (w == n_words - 1 && (n_edge & 63)) b &= (1ull << (n_edge & 63)) - 1ull;     // bits past the last edge
  return b;
}

__global__ void __launch_bounds__(256) rowlist_count(const unsigned long long* __restrict__ bits, int n_words, int n_edge, int* __restrict__ wg_count) {
  __shared__ int red[4];
  const int w = blockIdx.x * 256 + threadIdx.x;
  int c = __popcll(rowlist_word(bits, w, n_words, n_edge));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) wg_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) rowlist_fill(const unsigned long long* __restrict__ bits, int n_words, int n_edge,
                                                    const int* __restrict__ wg_off, int* __restrict__ rows) {
  __shared__ int part[256];
  const int t = threadIdx.x, w = blockIdx.x * 256 + t;
  unsigned long long b = rowlist_word(bits, w, n_words, n_edge);
  const int c = __popcll(b);
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int pos = wg_off[blockIdx.x] + part[t] - c;
  while (b) {
    const int j = __builtin_ctzll(b);
    b &= b - 1;
    rows[pos++] = 64 * w + j;
  }
}

// exclusive scan of cnt[0..n) into out[0..n], out[n] = total.  One workgroup.
__global__ void __launch_bounds__(1024) rowlist_scan(const int* __restrict__ cnt, int n, int* __restrict__ out) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = t * per, e = min(n, b + per);
  int sum = 0;
  for (int i = b; i < e; ++i) sum += cnt[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int i = b; i < e; ++i) { const int d = cnt[i]; out[i] = run; run += d; }
  if (t == 1023) out[n] = part[1023];
}

struct PwBwdArgs {
  int n_edge;
  const int* rows;           // ascending edge indices with a non-zero d_pw row
  const int* n_rows;         // their number (device)
  const float* pw; const float* d_pw; const float* h1; const float* h2;
  const float* w2; const float* w3;          // natural [256,256], [256,32]
  float* d_h1;
  float* arena; long long stride;
  long long o_w2, o_b2, o_w3, o_b3;
};

// Asynchronous global -> LDS copy of a [32][256] fp32 tile (one 1 KB row per wave-instruction: the LDS
// destination of global_load_lds is wave-uniform base + lane * 16, exactly one padded row).  No staging
// registers; completion is tracked by vmcnt.  Rows past the end re-read the last real row (finite data;
// their d3 rows are zero, so they contribute nothing).
__device__ __forceinline__ void dma_tile32(float* sdst, const float* __restrict__ g, long long e0, long long n_rows,
                                           int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 4 + q;
    const long long er = min(e0 + row, n_rows - 1);
    const float* src = g + (size_t)er * D_H + 4 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sdst + row * LD256), 16, 0, 0);
  }
}

// the same copy for rows given by an index list (rows past the list re-read its last row)
__device__ __forceinline__ void dma_rows32(float* sdst, const float* __restrict__ g, const int* __restrict__ rows, int p0,
                                           int n_rows, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 4 + q;
    const int er = rows[min(p0 + row, n_rows - 1)];
    const float* src = g + (size_t)er * D_H + 4 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sdst + row * LD256), 16, 0, 0);
  }
}

__global__ void __launch_bounds__(512) pw_bwd_main(const PwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // two tile buffers {h1 [32][260], h2 [32][260]} filled by DMA one tile ahead, + d3 [32][36]
  float* sD3 = smem + 4 * 32 * LD256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  f32x16 aW2[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) aW2[0][j] = zero16();
  f32x16 aW3 = zero16();
  float gb2 = 0.f, gb3 = 0.f;
  __shared__ int sRows[2][32];              // edge index of the tile's rows (this tile / the next one)
  const int n_rows = *a.n_rows;
  const int ntiles = (n_rows + 31) / 32;
  // Everything a tile reads from HBM is requested one tile ahead and BEFORE the tile's 16 d_h1 stores: the
  // h1/h2 tiles by DMA into the other LDS buffer, the d_pw / pw values of the d3 tile into registers.  The
  // wait at the top of a tile is then vmcnt(16): "everything but the 16 youngest operations", i.e. it never
  // waits for the previous tile's stores to be acknowledged (vmcnt is one in-order counter for loads and
  // stores).  W3 (the same for every tile) stays in registers for the same reason.
  f32x4 w3f[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w3f[k] = *reinterpret_cast<const f32x4*>(a.w3 + (size_t)(32 * wave + col) * D_E + 4 * half + 8 * k);
  float pq0 = 0.f, pq1 = 0.f, dq0 = 0.f, dq1 = 0.f;     // d3 tile sources: elements tid and tid + 512 of [32][32]
  // rows past the list re-read its last row (finite data); their d3 rows are zero
#define PB_ROW(tile_, r_) a.rows[min((tile_) * 32 + (r_), n_rows - 1)]
  // the row indices themselves are fetched one tile earlier still (ra / rb / rs), so the d3 source requests
  // never wait for an index
  int ra = 0, rb = 0, rs = 0, rs_tile = 0;
#define PB_LOAD_ROWIDS(tile_)                                                                           \
  do {                                                                                                  \
    ra = PB_ROW(tile_, tid >> 5); rb = PB_ROW(tile_, (tid >> 5) + 16);                                  \
    rs = (tile_) * 32 + (tid & 31) < n_rows ? a.rows[(tile_) * 32 + (tid & 31)] : a.n_edge;   /* slack row */ \
  } while (0)
#define PB_PREFETCH_D3()                                                                                \
  do {                                                                                                  \
    pq0 = a.pw[(size_t)ra * D_E + (tid & 31)]; dq0 = a.d_pw[(size_t)ra * D_E + (tid & 31)];             \
    pq1 = a.pw[(size_t)rb * D_E + (tid & 31)]; dq1 = a.d_pw[(size_t)rb * D_E + (tid & 31)];             \
    rs_tile = rs;                                                                                       \
  } while (0)
  if ((int)blockIdx.x < ntiles) {
    PB_LOAD_ROWIDS((int)blockIdx.x);
    dma_rows32(smem, a.h1, a.rows, (int)blockIdx.x * 32, n_rows, wave, lane);
    dma_rows32(smem + 32 * LD256, a.h2, a.rows, (int)blockIdx.x * 32, n_rows, wave, lane);
    PB_PREFETCH_D3();
    if ((int)(blockIdx.x + gridDim.x) < ntiles) PB_LOAD_ROWIDS((int)(blockIdx.x + gridDim.x));
  }
  drain_vmem_before_loop();
  int it = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
    const int e0 = t * 32;                 // first list position of the tile
    float* sH1 = smem + (it & 1) * (2 * 32 * LD256);
    float* sH2 = sH1 + 32 * LD256;        // fc2 output, then d(fc2 pre-activation)
    float* nH1 = smem + ((it & 1) ^ 1) * (2 * 32 * LD256);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // this tile's DMA and d3 sources (issued one tile ago) have landed
    if (tid < 32) sRows[it & 1][tid] = rs_tile;
    {
      const int row0 = tid >> 5, j = tid & 31;            // rows past the list: zero gradient
      sD3[row0 * LD32 + j] = (e0 + row0 < n_rows && pq0 > 0.f) ? dq0 : 0.f;                // ReLU of fc3
      sD3[(row0 + 16) * LD32 + j] = (e0 + row0 + 16 < n_rows && pq1 > 0.f) ? dq1 : 0.f;
    }
    __syncthreads();
    // d(fc2 pre) tile w = (d3 . W3^T) * (h2 > 0)
    f32x16 d2 = zero16();
    {
      const float* ap = sD3 + col * LD32 + 4 * half;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 8 * k);
        d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w3f[k].x, d2, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w3f[k].y, d2, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w3f[k].z, d2, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w3f[k].w, d2, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[r] = sH2[crow(r, half) * LD256 + 32 * wave + col] > 0.f ? d2[r] : 0.f;
    // d W3 += h2^T . d3 (rows [32w, 32w+32) of W3)
    {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + h;
        aW3 = __builtin_amdgcn_mfma_f32_32x32x2f32(sH2[row * LD256 + 32 * wave + r], sD3[row * LD32 + r], aW3, 0, 0, 0);
      }
    }
    if (tid < D_E) gb3 += col_sum32(sD3, LD32, tid);
    __syncthreads();      // every wave is done with the fc2 outputs (and with d3)
#pragma unroll
    for (int r = 0; r < 16; ++r) sH2[crow(r, half) * LD256 + 32 * wave + col] = d2[r];
    __syncthreads();
    // next tile's h1/h2 -> the other buffer (last read during the previous tile).  Issued here: the
    // d W2 phase below touches only LDS, so the in-order vmcnt never waits on this copy.
    if (t + (int)gridDim.x < ntiles) {
      dma_rows32(nH1, a.h1, a.rows, e0 + (int)gridDim.x * 32, n_rows, wave, lane);
      dma_rows32(nH1 + 32 * LD256, a.h2, a.rows, e0 + (int)gridDim.x * 32, n_rows, wave, lane);
      PB_PREFETCH_D3();
      if (t + 2 * (int)gridDim.x < ntiles) PB_LOAD_ROWIDS(t + 2 * (int)gridDim.x);
    }
    if (tid < D_H) gb2 += col_sum32(sH2, LD256, tid);
    // d W2 += h1^T . d2 (rows [32w, 32w+32) of W2, all 256 columns)
    mma_xty<1, 8>(aW2, sH1 + 32 * wave, LD256, sH2, LD256, lane);
    // d(fc1 pre) tile w = (d2 . W2^T) * (h1 > 0)
    {
      f32x16 acc = zero16();
      mma_abt_gB<D_H, 6>(acc, sH2, LD256, a.w2 + (size_t)(32 * wave) * D_H, D_H, lane);
      // d_h1 rows go back to their edge positions (rows past the list: the slack row E); exactly 16 stores
      const int* rp = sRows[it & 1] + 4 * half;
      const float* hp = sH1 + (4 * half) * LD256 + 32 * wave + col;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        a.d_h1[(size_t)rp[crow(r, 0)] * D_H + 32 * wave + col] = hp[crow(r, 0) * LD256] > 0.f ? acc[r] : 0.f;
    }
  }
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
#pragma unroll
  for (int j = 0; j < 8; ++j) store_acc(ar + a.o_w2 + (size_t)(32 * wave) * D_H + 32 * j, D_H, aW2[0][j], lane);
  store_acc(ar + a.o_w3 + (size_t)(32 * wave) * D_E, D_E, aW3, lane);
  if (tid < D_H) ar[a.o_b2 + tid] = gb2;
  if (tid < D_E) ar[a.o_b3 + tid] = gb3;
}

constexpr size_t kPwBwdSmem = (size_t)(4 * 32 * LD256 + 32 * LD32) * sizeof(float);

// fc1 of the pw-MLP: d W1 = X^T . d_h1 with X = [one-hot(c) * s_c | one-hot(n) * s_n | geo(7)].
// The score columns factor through per-detection sums (deterministic, no class table in LDS):
//   d W1[class k        ] = sum_{i : class_i = k} s_i * S[i],  S[i] = sum over i's own pairs of d_h1
//   d W1[C' + class k   ] = sum_{i : class_i = k} s_i * T[i],  T[i] = sum over the reversed pairs
// pw_w1_nodesums streams d_h1 twice (own rows, reversed rows: 2 x 1 KB per edge, HBM-bound) and also
// accumulates the 7 geometry rows and the bias; pw_w1_classrows folds S/T by class.
struct PwW1Args {
  int n_det; int cprime; int multiclass;
  const int* row_ptr; const int* edge_t; const float* geo; const float* d_h1;
  const float* scores; const int* classes;
  const unsigned long long* eany;   // bit e: d_h1 row e is valid (edges without gradient were never written)
  float* w1_s; float* w1_t;
  float* arena; long long stride;
  long long o_w1, o_b1;
  int nchunks;
};

__global__ void __launch_bounds__(256) pw_w1_nodesums(const PwW1Args a) {
  __shared__ float red[4 * 8 * D_H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 g[7], gb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 7; ++k) g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nwaves = gridDim.x * 4;
  for (int node = blockIdx.x * 4 + wave; node < a.n_det; node += nwaves) {
    const int eb = a.row_ptr[node], ee = a.row_ptr[node + 1];
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f), T = S;
#define W1_BIT(e_) ((a.eany[(e_) >> 6] >> ((e_) & 63)) & 1ull)
#define ACC4(dst, s, v) dst.x = fmaf(s, v.x, dst.x); dst.y = fmaf(s, v.y, dst.y); dst.z = fmaf(s, v.z, dst.z); dst.w = fmaf(s, v.w, dst.w)
    // 64 edges per pass: reversed-pair positions and the "carries gradient" bits of both the own and the
    // reversed rows arrive in coalesced / L2-resident loads; only rows that carry gradient are read (the
    // others were never written: exact zeros), two at a time, ascending edge order
    for (int base = eb; base < ee; base += 64) {
      const int el = base + lane;
      const bool in = el < ee;
      const int tt = in ? a.edge_t[el] : 0;
      unsigned long long mo = __ballot(in && W1_BIT(el));
      unsigned long long mr = __ballot(in && W1_BIT(tt));
      while (mo) {
        const int j0 = __builtin_ctzll(mo); mo &= mo - 1;
        const bool two = mo != 0ull;
        const int j1 = two ? __builtin_ctzll(mo) : j0; mo &= mo - 1;      // (mo == 0 stays 0)
        const int e = base + j0, e1 = base + j1;
        const float4 d0 = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)e * D_H + 4 * lane);
        const float4 d1 = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)e1 * D_H + 4 * lane);
        const float4 ga0 = *reinterpret_cast<const float4*>(a.geo + (size_t)e * 8);
        const float4 gc0 = *reinterpret_cast<const float4*>(a.geo + (size_t)e * 8 + 4);
        const float4 ga1 = *reinterpret_cast<const float4*>(a.geo + (size_t)e1 * 8);
        const float4 gc1 = *reinterpret_cast<const float4*>(a.geo + (size_t)e1 * 8 + 4);
        S.x += d0.x; S.y += d0.y; S.z += d0.z; S.w += d0.w;
        ACC4(g[0], ga0.x, d0); ACC4(g[1], ga0.y, d0); ACC4(g[2], ga0.z, d0); ACC4(g[3], ga0.w, d0);
        ACC4(g[4], gc0.x, d0); ACC4(g[5], gc0.y, d0); ACC4(g[6], gc0.z, d0);
        if (two) {
          S.x += d1.x; S.y += d1.y; S.z += d1.z; S.w += d1.w;
          ACC4(g[0], ga1.x, d1); ACC4(g[1], ga1.y, d1); ACC4(g[2], ga1.z, d1); ACC4(g[3], ga1.w, d1);
          ACC4(g[4], gc1.x, d1); ACC4(g[5], gc1.y, d1); ACC4(g[6], gc1.z, d1);
        }
      }
      while (mr) {
        const int j0 = __builtin_ctzll(mr); mr &= mr - 1;
        const bool two = mr != 0ull;
        const int j1 = two ? __builtin_ctzll(mr) : j0; mr &= mr - 1;
        const int r0 = __builtin_amdgcn_readlane(tt, j0), r1 = __builtin_amdgcn_readlane(tt, j1);
        const float4 t0 = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)r0 * D_H + 4 * lane);
        const float4 t1 = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)r1 * D_H + 4 * lane);
        T.x += t0.x; T.y += t0.y; T.z += t0.z; T.w += t0.w;
        if (two) { T.x += t1.x; T.y += t1.y; T.z += t1.z; T.w += t1.w; }
      }
    }
#undef W1_BIT
#undef ACC4
    gb.x += S.x; gb.y += S.y; gb.z += S.z; gb.w += S.w;                 // bias: sum of d_h1 over all edges
    *reinterpret_cast<float4*>(a.w1_s + (size_t)node * D_H + 4 * lane) = S;
    *reinterpret_cast<float4*>(a.w1_t + (size_t)node * D_H + 4 * lane) = T;
  }
  // fold the four waves, write the 7 geometry rows + bias partial of this workgroup
#pragma unroll
  for (int k = 0; k < 7; ++k) *reinterpret_cast<float4*>(red + (wave * 8 + k) * D_H + 4 * lane) = g[k];
  *reinterpret_cast<float4*>(red + (wave * 8 + 7) * D_H + 4 * lane) = gb;
  __syncthreads();
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  const int f = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = red[(0 * 8 + k) * D_H + f] + red[(1 * 8 + k) * D_H + f] + red[(2 * 8 + k) * D_H + f] + red[(3 * 8 + k) * D_H + f];
    if (k < 7) ar[a.o_w1 + (size_t)(2 * a.cprime + k) * D_H + f] = v; else ar[a.o_b1 + f] = v;
  }
}

// grid (2 C', nchunks): row r < C' folds S by class r + 1 (centre score column), row C' + k folds T.
// Each wave scans its quarter of the chunk 64 detections at a time (one coalesced class/score load,
// a ballot picks the members of the class), then adds the selected 1 KB rows in index order.
__global__ void __launch_bounds__(256) pw_w1_classrows(const PwW1Args a) {
  __shared__ float red[4 * D_H];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = r < a.cprime ? r : r - a.cprime;
  const float* src = r < a.cprime ? a.w1_s : a.w1_t;
  const int per = (a.n_det + a.nchunks - 1) / a.nchunks;
  const int c0 = blockIdx.y * per, c1 = min(a.n_det, c0 + per);
  const int wper = ((c1 - c0 + 3) / 4 + 63) / 64 * 64;
  const int i0 = c0 + wave * wper, i1 = min(c1, i0 + wper);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int base = i0; base < i1; base += 64) {
    const int i = base + lane;
    const bool valid = i < i1;
    const float sc = valid ? a.scores[i] : 0.f;
    const bool member = valid && (!a.multiclass || a.classes[i] - 1 == k);
    unsigned long long mask = __ballot(member);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float s = __shfl(sc, j);
      const float* row = src + (size_t)(base + j) * D_H + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(s, row[64 * q], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[wave * D_H + 64 * q + lane] = acc[q];
  __syncthreads();
  const int f = threadIdx.x;
  a.arena[(size_t)blockIdx.y * a.stride + a.o_w1 + (size_t)r * D_H + f] =
      red[f] + red[D_H + f] + red[2 * D_H + f] + red[3 * D_H + f];
}

// ------------------------------------------------------------------------------------------
// grads[p] = sum over the n(p) partial copies arena[k][p], k ascending.
struct ReduceArgs {
  const float* arena; long long stride; long long total;
  long long w1c_end;      // end of the score-column rows of pw fc1 (2 C' x 256)
  long long pw1_end;      // end of pw fc1 weights+bias
  long long pw_end;       // end of the pw-MLP parameters
  long long blk_sz; int nblocks;
  int n_w1c, n_w1, n_pw, n_edge, n_node, n_head;
  float* grads;
};

__global__ void __launch_bounds__(256) reduce_partials(const ReduceArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.total) return;
  int n;
  if (p < a.w1c_end) n = a.n_w1c;
  else if (p < a.pw1_end) n = a.n_w1;
  else if (p < a.pw_end) n = a.n_pw;
  else if (p < a.pw_end + a.blk_sz * a.nblocks) {
    const long long q = (p - a.pw_end) % a.blk_sz;
    const long long w1 = D_S * D_R + D_R;                       // start of pw_fc1 weights
    const long long w2 = w1 + (D_E + 2 * D_R) * D_P + D_P;      // start of pw_fc2 weights
    const bool edge = (q >= w1 && q < w1 + D_E * D_P) || (q >= w2 && q < w2 + D_P * D_P + D_P);
    n = edge ? a.n_edge : a.n_node;
  } else n = a.n_head;
  float v = 0.f;
  const float* src = a.arena + p;
  for (int k = 0; k < n; ++k) v += src[(size_t)k * a.stride];
  a.grads[p] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" int gnet_backward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                             const float* params, gnet_buffers* buf, float* grads, gnet_stream_t stream) {
  clear_hip_error();
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !in || !params || !buf || !grads) return GNET_ERR_INVALID;
  if (!buf->arena || !buf->d_x || !buf->d_logits || !buf->pw_h1 || !buf->blk_h1[1] || !buf->blk_parg[1] || !buf->emask || !buf->pw_rows) return GNET_ERR_INVALID;   // plan(training=1)
  hipStream_t s = (hipStream_t)stream;
  const ParamLayout L = make_layout(cfg);
  const int B = cfg->num_blocks;
  const int N = shape->n_det;
  const int E = (int)shape->n_edge;
  if (N == 0) {
    HIP_CHECK_RET(hipMemsetAsync(grads, 0, (size_t)L.total * sizeof(float), s));
    return GNET_OK;
  }
  if ((size_t)GNET_ARENA_PARTIALS * (size_t)L.total > buf->arena_floats) return GNET_ERR_WORKSPACE;
  if (E > (1 << 24) - 128) return GNET_ERR_UNSUPPORTED;   // 32-bit byte offsets into the [E,64] fp32 arrays
  const float* pt = buf->packed_t;
  void* prof = buf->profiler;
  const long long stride = L.total;
  const int ntile_n = (N + 31) / 32;
  const int g_node = min(ntile_n, 256);
  const int etiles = (E + 31) / 32;
  const int g_edge = E > 0 ? max(1, min(GNET_ARENA_PARTIALS, (E + EB_T - 1) / EB_T)) : 0;
  const int g_pw = E > 0 ? min(etiles, GNET_ARENA_PARTIALS) : 0;
  const int g_w1 = E > 0 ? max(1, min(GNET_ARENA_PARTIALS, (N + 3) / 4)) : 0;          // node-sum workgroups
  const int g_w1c = E > 0 ? max(1, min(128, 256 / (2 * L.cprime))) : 0;               // node chunks per class row

  // GNET_DENSE_BWD=1 forces the dense edge stage for every block (A/B measurements, tests of the dense path)
  static const bool g_force_dense = getenv("GNET_DENSE_BWD") && atoi(getenv("GNET_DENSE_BWD")) != 0;
  // the edge stages accumulate into d_pw (the sparse one touches winner rows only)
  if (E > 0) HIP_CHECK_RET(hipMemsetAsync(buf->d_pw, 0, (size_t)E * D_E * sizeof(float), s));
  // winner bitmaps behind the per-edge masks: ewin (this block), eany (OR over the blocks)
  const size_t n_words = ((size_t)E + 63) / 64;
  const size_t bm_stride = (n_words + 256 + 63) & ~(size_t)63;            // whole 256-word scan chunks
  const size_t em_stride = ((size_t)E + 128 + 63) & ~(size_t)63;         // per-block mask arrays
  unsigned long long* ewin = (unsigned long long*)buf->emask + (size_t)B * em_stride;
  unsigned long long* eany = ewin + bm_stride;
  if (E > 0) HIP_CHECK_RET(hipMemsetAsync(eany, g_force_dense ? 0xff : 0, bm_stride * sizeof(unsigned long long), s));
  if (E > 0 && !g_force_dense) {
    // per-edge column masks of ALL blocks in one launch (they depend on the forward pass only)
    HIP_CHECK_RET(hipMemsetAsync(buf->emask, 0, (size_t)B * em_stride * sizeof(unsigned long long), s));
    WinArgs w;
    w.n_det = N; w.emask_stride = (long long)em_stride; w.emask = (unsigned long long*)buf->emask; w.row_ptr = buf->row_ptr;
    for (int b = 1; b <= B; ++b) {
      w.pm[b - 1] = (const unsigned long long*)buf->blk_pm[b]; w.parg[b - 1] = (const unsigned long long*)buf->blk_parg[b];
      w.h1[b - 1] = buf->blk_h1[b]; w.w2t[b - 1] = pt + L.blk[b].w2; w.b2[b - 1] = params + L.blk[b].b2;
    }
    GNET_LAUNCH(prof, GNET_K_BLK_POST, s, winners_mark<<<dim3(min((N + 3) / 4, 1024), B), 256, 0, s>>>(w));
  }
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeBwdSmem));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_bwd_sparse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeBwdSparseSmem));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_bwd_main, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwBwdSmem));
    attr_set = true;
  }

  {
    HeadBwdArgs h;
    h.n_det = N; h.d_logits = buf->d_logits; h.head2 = buf->head2; h.head1 = buf->head1; h.xb = buf->block_feats[B];
    h.hw1 = params + L.hw1; h.hw2 = params + L.hw2; h.hwl = params + L.hwl;
    h.d_x = buf->d_x; h.arena = buf->arena; h.stride = stride;
    h.o_hw1 = L.hw1; h.o_hb1 = L.hb1; h.o_hw2 = L.hw2; h.o_hb2 = L.hb2; h.o_hwl = L.hwl; h.o_hbl = L.hbl;
    GNET_LAUNCH(prof, GNET_K_HEAD_BWD, s, head_bwd<<<g_node, 256, 0, s>>>(h));
  }
  for (int b = B; b >= 1; --b) {
    const BlockLayout& K = L.blk[b];
    {
      BlkPostArgs p;
      p.n_det = N; p.d_x = buf->d_x; p.x_out = buf->block_feats[b]; p.q = buf->blk_q[b];
      p.pm = (const unsigned long long*)buf->blk_pm[b];
      p.w4 = params + K.w4; p.w3 = params + K.w3;
      p.d_pc = buf->d_pc;
      p.arena = buf->arena; p.stride = stride; p.o_w4 = K.w4; p.o_b4 = K.b4; p.o_w3 = K.w3; p.o_b3 = K.b3;
      GNET_LAUNCH(prof, GNET_K_BLK_POST, s, blk_bwd_post<<<g_node, 256, 0, s>>>(p));
    }
    if (E > 0 && !g_force_dense) {
      // ---- sparse edge stage: winner rows only (masks: winners_mark above)
      EdgeBwdSparseArgs e;
      e.n_edge = E; e.n_det = N; e.edge_c = buf->edge_c; e.emask = (const unsigned long long*)buf->emask + (size_t)(b - 1) * em_stride;
      e.pw = buf->pw_feats; e.h1 = buf->blk_h1[b]; e.d_pc = buf->d_pc;
      e.ewin = ewin; e.eany = eany;
      e.w1t = pt + K.w1; e.w2t = pt + K.w2;
      e.d_pw = buf->d_pw; e.d_g1 = buf->d_g1;
      e.arena = buf->arena; e.stride = stride; e.o_w1 = K.w1; e.o_w2 = K.w2; e.o_b2 = K.b2;
      GNET_LAUNCH(prof, GNET_K_EDGE_BWD, s, edge_bwd_sparse<<<g_edge, 256, kEdgeBwdSparseSmem, s>>>(e));
      GNET_LAUNCH(prof, GNET_K_BLK_PRE, s, gather_sparse<<<(N + 3) / 4, 256, 0, s>>>(buf->d_g1, buf->row_ptr, buf->edge_n, buf->edge_t,
                                                                                 (const unsigned long long*)ewin, N, buf->d_rc, buf->d_rn));
    } else if (E > 0) {
      // ---- dense edge stage (every edge row; GNET_DENSE_BWD=1)
      EdgeBwdArgs e;
      e.n_edge = E; e.n_det = N;
      e.edge_c = buf->edge_c; e.pw = buf->pw_feats; e.h1 = buf->blk_h1[b];
      e.pm = (const unsigned long long*)buf->blk_pm[b]; e.d_pc = buf->d_pc;
      e.w1t = pt + K.w1; e.w2t = pt + K.w2; e.b2 = params + K.b2; e.w1 = params + K.w1; e.w2 = params + K.w2;
      e.d_pw = buf->d_pw; e.d_g1 = buf->d_g1;
      e.arena = buf->arena; e.stride = stride; e.o_w1 = K.w1; e.o_w2 = K.w2; e.o_b2 = K.b2;
      GNET_LAUNCH(prof, GNET_K_EDGE_BWD, s, edge_bwd<<<g_edge, 256, kEdgeBwdSmem, s>>>(e));
      GNET_LAUNCH(prof, GNET_K_BLK_PRE, s, gather_sums<<<(N + 3) / 4, 256, 0, s>>>(buf->d_g1, buf->row_ptr, buf->edge_n, buf->edge_t,
                                                                                N, buf->d_rc, buf->d_rn));
    } else {
      HIP_CHECK_RET(hipMemsetAsync(buf->d_rc, 0, (size_t)N * D_P * sizeof(float), s));
      HIP_CHECK_RET(hipMemsetAsync(buf->d_rn, 0, (size_t)N * D_P * sizeof(float), s));
    }
    {
      BlkPreArgs p;
      p.n_det = N; p.write_dx = b > 1;
      p.d_rc = buf->d_rc; p.d_rn = buf->d_rn; p.r = buf->blk_r[b];
      p.x_prev = b > 1 ? buf->block_feats[b - 1] : nullptr;
      p.w1 = params + K.w1; p.wr = params + K.wr; p.d_x = buf->d_x;
      p.arena = buf->arena; p.stride = stride; p.o_w1 = K.w1; p.o_b1 = K.b1; p.o_wr = K.wr; p.o_br = K.br;
      GNET_LAUNCH(prof, GNET_K_BLK_PRE, s, blk_bwd_pre<<<g_node, 256, 0, s>>>(p));
    }
  }
  if (E > 0) {
    // rows of the pw-MLP backward = edges with a non-zero d_pw row
    int* rl_count = buf->scratch_i;                     // [n_wg], then offsets [n_wg + 1] (total last)
    const int n_wg = (int)((n_words + 255) / 256);
    int* rl_off = rl_count + n_wg;
    if (2 * n_wg + 1 > N + 1024) return GNET_ERR_WORKSPACE;
    rowlist_count<<<n_wg, 256, 0, s>>>(eany, (int)n_words, E, rl_count);
    rowlist_scan<<<1, 1024, 0, s>>>(rl_count, n_wg, rl_off);
    rowlist_fill<<<n_wg, 256, 0, s>>>(eany, (int)n_words, E, rl_off, buf->pw_rows);
    PwBwdArgs p;
    p.rows = buf->pw_rows; p.n_rows = rl_off + n_wg;
    p.n_edge = E; p.pw = buf->pw_feats; p.d_pw = buf->d_pw; p.h1 = buf->pw_h1; p.h2 = buf->pw_h2;
    p.w2 = params + L.pw2; p.w3 = params + L.pw3; p.d_h1 = buf->d_h1;
    p.arena = buf->arena; p.stride = stride; p.o_w2 = L.pw2; p.o_b2 = L.pb2; p.o_w3 = L.pw3; p.o_b3 = L.pb3;
    GNET_LAUNCH(prof, GNET_K_PW_BWD, s, pw_bwd_main<<<g_pw, 512, kPwBwdSmem, s>>>(p));
    PwW1Args w;
    w.n_det = N; w.cprime = L.cprime; w.multiclass = cfg->num_classes > 1;
    w.row_ptr = buf->row_ptr; w.edge_t = buf->edge_t; w.geo = buf->geo; w.d_h1 = buf->d_h1;
    w.scores = in->det_scores; w.classes = in->det_classes; w.eany = eany;
    w.w1_s = buf->w1_s; w.w1_t = buf->w1_t;
    w.arena = buf->arena; w.stride = stride; w.o_w1 = L.pw1; w.o_b1 = L.pb1; w.nchunks = g_w1c;
    GNET_LAUNCH(prof, GNET_K_PW_W1, s, pw_w1_nodesums<<<g_w1, 256, 0, s>>>(w));
    GNET_LAUNCH(prof, GNET_K_PW_W1, s, pw_w1_classrows<<<dim3(2 * L.cprime, g_w1c), 256, 0, s>>>(w));
  }
  {
    ReduceArgs r;
    r.arena = buf->arena; r.stride = stride; r.total = L.total;
    r.w1c_end = (long long)2 * L.cprime * D_H; r.pw1_end = L.pw2; r.pw_end = L.blk[1].wr;
    r.blk_sz = (B > 1) ? (L.blk[2].wr - L.blk[1].wr) : (L.hw1 - L.blk[1].wr);
    r.nblocks = B;
    r.n_w1c = g_w1c; r.n_w1 = g_w1; r.n_pw = g_pw; r.n_edge = g_edge; r.n_node = g_node; r.n_head = g_node;
    r.grads = grads;
    GNET_LAUNCH(prof, GNET_K_REDUCE, s, reduce_partials<<<(int)((L.total + 255) / 256), 256, 0, s>>>(r));
  }
  return launch_status();
}
