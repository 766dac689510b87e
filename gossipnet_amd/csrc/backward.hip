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
//   winners_mark + winner lists of every block (backward_edge.hip; depend on the forward pass only)
//   per block b = B..1, three launches:
//     gather_winners  centre / reversed-edge sums of block b+1's compact g1 rows -> d_rc, d_rn
//     blk_bwd_node  [per-node halves of pw_fc1, reduce_dim of block b+1 -> d_x += drpre . Wr^T] +
//                   [shortcut ReLU, fc2, fc1 of block b -> d_x := dz, d_pc]
//     edge_bwd_w    (backward_edge.hip) pw_fc2, pw_fc1 on 32-row tiles of the winner edges -> d_pw (+=), g1 rows
//   gather_winners + blk_bwd_node once more for the first block's pre stage
//   pw_bwd_main    pw_feats fc3, fc2 (+ d_h1 = grad wrt fc1 pre-activation), on the listed rows
//   pw_w1_nodesums + pw_w1_classrows   pw_feats fc1 (score columns via per-detection sums, 7 geometry rows)
//   reduce_partials  sums the per-workgroup partial weight gradients in a fixed order
// Weight gradients are accumulated in MFMA accumulators across a workgroup's tiles and written once
// per workgroup to an arena; reduce_partials adds them in index order.  Every sum has a fixed order (the one
// float atomic, d_pw += in edge_bwd_w, receives exactly one addition per element and launch: ordered by the
// launch order), so gradients are reproducible run to run.
#include <type_traits>
#include "common.hpp"
#include <stdlib.h>
#include "backward_edge.hpp"

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

__device__ __forceinline__ float col_sum8(const float* s, int ld, int col) {
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) v += s[r * ld + col];
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
    {
      // the three [32][128] tiles: all twelve 16-byte requests of a thread before the first LDS store
      static_assert(D_S == D_HEAD, "one offset for the three tiles");
      float4 v2[4], v1[4], vx[4];
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
        const bool ok = row0 + row < a.n_det;
        const size_t o = (size_t)(row0 + row) * D_HEAD + 4 * c4;
        v2[j] = ok ? *reinterpret_cast<const float4*>(a.head2 + o) : z4;
        v1[j] = ok ? *reinterpret_cast<const float4*>(a.head1 + o) : z4;
        vx[j] = ok ? *reinterpret_cast<const float4*>(a.xb + o) : z4;
      }
      const float dl = (tid < 32 && row0 + tid < a.n_det) ? a.d_logits[row0 + tid] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
        *reinterpret_cast<float4*>(sH2 + row * LD128 + 4 * c4) = v2[j];
        *reinterpret_cast<float4*>(sH1 + row * LD128 + 4 * c4) = v1[j];
        *reinterpret_cast<float4*>(sX + row * LD128 + 4 * c4) = vx[j];
      }
      if (tid < 32) sDl[tid] = dl;
    }
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
      mma_abt_gB<D_HEAD, 8>(acc, sD2, LD128, a.hw2 + (size_t)(32 * wave) * D_HEAD, D_HEAD, lane);   // weight loads 8 k-steps ahead
#pragma unroll
      for (int r = 0; r < 16; ++r) sD1[crow(r, half) * LD128 + 32 * wave + col] = acc[r];
    }
    __syncthreads();
    mma_xty<1, 4>(aW1, sX + 32 * wave, LD128, sD1, LD128, lane);            // d W(fc1) += x^T . d head1
    if (tid < D_HEAD) gb1 += col_sum32(sD1, LD128, tid);
    {
      f32x16 acc = zero16();                                                // d x = d head1 . W1^T
      mma_abt_gB<D_HEAD, 8>(acc, sD1, LD128, a.hw1 + (size_t)(32 * wave) * D_HEAD, D_HEAD, lane);
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
// gather_winners: d_rc[i] = sum of g1 over i's winner edges (a contiguous range of the block's compact g1 rows),
// d_rn[i] = sum over i's edges e = (i, n), n != i (self pair: n_feats zeroed, network.py:371-374), of g1[reverse(e)]
// when the reversed pair is a winner of n.  One wave per detection (the sums are latency / L2-bandwidth bound: 16 000
// waves in flight; folded into the node kernel's 2 000 waves they took twice as long), four rows per
// wave-instruction, ascending order, no atomics.
// blk_bwd_node: the N-sized GEMMs between two edge stages of the backward chain in ONE launch:
//   pre(b)   per-node halves of pw_fc1 and reduce_dim of block b (network.py:348-376 backward):
//            d W1[32:96], d b1, d Wr, d br; d_x += drpre . Wr^T
//   post(b-1) shortcut ReLU, fc2, fc1 of block b-1 (network.py:390-408 backward): d_x := dz, d W4, d b4, d W3, d b3,
//            d_pc = dp / tie count (the SegmentMax gradient's even split, TF _SegmentMinOrMaxGrad)
// The forward pass fuses post(b) + pre(b+1) the same way (node_fwd).  One workgroup = 64 detections, 8 waves:
// wave = (row tile rt, role cw); the two row tiles run the same phase sequence in lock-step.
struct BlkNodeArgs {
  int n_det;
  int do_pre, do_post;
  int want_dx0;                               // pre stage of block 1 with start features: d_x := dz + drpre . Wr^T (unmasked)
  const float* d_rc; const float* d_rn;       // [N,64] gather_winners of block b (NULL: no edges)
  // pre (block b)
  const float* r; const float* x_prev;        // x_prev = block_feats[b-1] (NULL = zeros); also x_out of the post stage
  const float* w1; const float* wr;           // natural [96,64] (rows 32-63 centre, 64-95 neighbour), [128,32]
  long long o_w1, o_b1, o_wr, o_br;
  // neighbor_feats (network.py:356-365): the neighbour half of build_context comes from r_n = relu(x . Wrn + brn)
  const float* r_nb; const float* wrn;        // [N,32] relu(reduce_dim_neighbor), natural [128,32]; NULL without neighbor_feats
  long long o_wrn, o_brn;
  // post (block b-1)
  const float* q; const unsigned long long* pm;
  const float* w4; const float* w3;           // natural [64,128], [64,64]
  long long o_w4, o_b4, o_w3, o_b3;
  float* d_x; float* d_pc;
  float* arena; long long stride;
  GNET_TRACE_FIELD
};

constexpr int BN_FLOATS = 2 * 32 * LD128 + 2 * 32 * LD64 + 3 * 32 * LD32 + 4 * 32 * 32;   // X, DZ, Rc|Rn, Rr|Dr, Rrn, Part
constexpr size_t kBlkNodeSmem = (size_t)BN_FLOATS * sizeof(float);              // 80 KB

__device__ __forceinline__ unsigned long long bn_low_mask(int bit) { return bit ? (~0ull >> (64 - bit)) : 0ull; }
__device__ __forceinline__ int bn_winner_pos(const unsigned long long* __restrict__ ewin, const int* __restrict__ wprefix, int e) {
  const int w = e >> 6;
  return wprefix[w] + __popcll(ewin[w] & bn_low_mask(e & 63));
}

__global__ void __launch_bounds__(256) gather_winners(const float* __restrict__ g1c, const int* __restrict__ row_ptr,
                                                      const int* __restrict__ wrow, const int* __restrict__ tpos,
                                                      int n_det, float* __restrict__ d_rc, float* __restrict__ d_rn
#ifdef GNET_TRACE
                                                      , unsigned long long* trace_ptr
#endif
) {
#ifdef GNET_TRACE
  const struct { unsigned long long* trace; } tr = {trace_ptr};
#endif
  GSTAMP(tr, 0);
  // XCD-aware: workgroups go round-robin to the 8 XCDs; XCD x takes the x-th contiguous eighth of the detections (one image's
  // worth when the batch has 8), whose own rows AND reversed pairs' rows -- neighbours are detections of the same image -- lie in
  // the eighth of the list that edge_bwd_w's XCD x has just written: one L2 sees both reads of a row
  // (-1.6 %: probe builds put 0.30 of the kernel's 0.58 ms on the reversed rows -- random 256-byte gathers at 4.6 TB/s -- 0.13 on
  // the own rows, which stream at 10 TB/s out of the caches edge_bwd_w left them in, and 0.15 on the launches' fixed part)
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int node = lb * 4 + (threadIdx.x >> 6);
  if (node >= n_det) return;
  const int lane = threadIdx.x & 63, sub = lane >> 4, f4 = lane & 15;
  const int eb = row_ptr[node], ee = row_ptr[node + 1];
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sn = sc;
  // list positions come precomputed (winner_tpos): the own rows are [wrow[node], wrow[node + 1]), the reversed pairs'
  // positions tpos[e]; the first 64 of those are requested before the own rows are summed
  const int tp_first = eb + lane < ee ? tpos[eb + lane] : -1;
  const int p0 = wrow[node], p1 = wrow[node + 1];
  // the sums are latency-bound: eight rows per quarter-wave in flight (a missing row re-reads row p0 / position 0 with
  // weight 0); ascending order per quarter-wave, the quarter-waves are folded at the end (fixed order)
  for (int p = p0 + sub; p < p1; p += 32) {
    float4 c[8]; float w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool have = p + 4 * q < p1;
      w[q] = have ? 1.f : 0.f;
      c[q] = *reinterpret_cast<const float4*>(g1c + (size_t)(have ? p + 4 * q : p0) * D_P + 4 * f4);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { sc.x = fmaf(w[q], c[q].x, sc.x); sc.y = fmaf(w[q], c[q].y, sc.y); sc.z = fmaf(w[q], c[q].z, sc.z); sc.w = fmaf(w[q], c[q].w, sc.w); }
  }
  GSTAMP(tr, 1);
  for (int base = eb; base < ee; base += 64) {
    const int el = base + lane;
    int tp = -1;                                            // list position of the reversed pair, if it is a winner
    if (base == eb) tp = tp_first;
    else if (el < ee) tp = tpos[el];
    unsigned long long mr = __ballot(tp >= 0);
    while (mr) {
      // next 32 reversed rows: eight per quarter-wave, all requested before the first is added (a detection with many
      // winning neighbours used to need one memory round trip per 16 rows: the slowest wave set the kernel's duration)
      int j[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
#pragma unroll
      for (int q = 0; q < 32; ++q) { if (mr) { const int b = __builtin_ctzll(mr); mr &= mr - 1; if ((q & 3) == sub) j[q >> 2] = b; } }
      float4 v[8]; float w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pq = __shfl(tp, j[q] < 0 ? 0 : j[q]);
        w[q] = j[q] >= 0 ? 1.f : 0.f;
        v[q] = *reinterpret_cast<const float4*>(g1c + (size_t)max(pq, 0) * D_P + 4 * f4);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { sn.x = fmaf(w[q], v[q].x, sn.x); sn.y = fmaf(w[q], v[q].y, sn.y); sn.z = fmaf(w[q], v[q].z, sn.z); sn.w = fmaf(w[q], v[q].w, sn.w); }
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
  GSTAMP(tr, 15);
}

// acc += X^T . Y over the 32 rows of a tile (weight-gradient shape), X / Y already offset to the lane's column: all 32
// LDS operands are requested before the 16 MFMAs (one wave per SIMD: nobody else hides a read placed in front of each MFMA)
__device__ __forceinline__ void dw_tile(f32x16& acc, const float* X, int ldx, const float* Y, int ldy, int half) {
  float xa[16], ya[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { xa[kk] = X[(2 * kk + half) * ldx]; ya[kk] = Y[(2 * kk + half) * ldy]; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk], ya[kk], acc, 0, 0, 0);
}

// Inputs of one 32-detection tile of blk_bwd_node, in registers (256 threads): x, d_x [32][128]; d_rc, d_rn, q [32][64];
// r, r_n [32][32]; segment-max records [32][64].
struct NodeTileIn { float4 vx[4], vz[4], vr, vrn, vc[2], vn[2]; };            // needed at the top of a tile: requested one tile ahead
struct NodeTileMid { float4 vq[2]; unsigned long long vpm[8]; };              // needed from the post stage on: requested at the tile's top
template <bool NF>
__device__ __forceinline__ void load_node_tile(NodeTileIn& in, const BlkNodeArgs& a, int row0, int tid) {
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  in.vr = in.vrn = z4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
    const bool ok = row0 + row < a.n_det;
    in.vx[j] = (a.x_prev && ok) ? ldg4_b(a.x_prev, (unsigned)(row0 + row) * (D_S * 4u) + 16u * c4) : z4;
    in.vz[j] = ok ? ldg4_b(a.d_x, (unsigned)(row0 + row) * (D_S * 4u) + 16u * c4) : z4;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 256 * j, row = i >> 4, c4 = i & 15;
    const bool ok = row0 + row < a.n_det;
    in.vc[j] = (a.do_pre && a.d_rc && ok) ? ldg4_b(a.d_rc, (unsigned)(row0 + row) * (D_P * 4u) + 16u * c4) : z4;
    in.vn[j] = (a.do_pre && a.d_rc && ok) ? ldg4_b(a.d_rn, (unsigned)(row0 + row) * (D_P * 4u) + 16u * c4) : z4;
  }
  if (a.do_pre) {
    const int row = tid >> 3, c4 = tid & 7;
    const bool ok = row0 + row < a.n_det;
    if (ok) in.vr = ldg4_b(a.r, (unsigned)(row0 + row) * (D_R * 4u) + 16u * c4);
    if (NF && ok) in.vrn = ldg4_b(a.r_nb, (unsigned)(row0 + row) * (D_R * 4u) + 16u * c4);
  }
}
__device__ __forceinline__ void load_node_mid(NodeTileMid& m, const BlkNodeArgs& a, int row0, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 256 * j, row = i >> 4, c4 = i & 15;
    m.vq[j] = (a.do_post && row0 + row < a.n_det) ? ldg4_b(a.q, (unsigned)(row0 + row) * (D_P * 4u) + 16u * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + 256 * j, row = i >> 6, ff = i & 63;
    m.vpm[j] = (a.do_post && row0 + row < a.n_det) ? a.pm[(size_t)(row0 + row) * D_P + ff] : 0ull;
  }
}

// One workgroup (4 waves, one per SIMD: the whole register file is theirs) walks 32-detection tiles.  The stages of
// a tile are a chain of small products separated by barriers, so nothing inside the chain may wait for memory:
// the weight slices each wave multiplies by are loaded into registers once, before the first tile, and all seven
// input tiles of a detection tile (x, d_x, r, [r_n,] d_rc, d_rn, q, segment-max records) are requested together
// at its top.  The weight-gradient accumulators stay in registers across the tiles of the workgroup.
// ONE: one tile per workgroup at 256 registers, so that TWO workgroups share a CU (2 x 80 KB of LDS) and their barrier-separated
// chains interleave on its SIMDs -- the node_fwd arrangement.  The weight slices are then requested a stage ahead of their use
// instead of once for all tiles (they cost 96 of the 492 registers), and there is no next tile to prefetch.  Used while there is
// at most one tile per CU (N <= 8192 detections per step: see gnet_backward); the looping form covers larger steps.
template <bool NF, bool ONE>
__global__ void __launch_bounds__(256, ONE ? 2 : 1) blk_bwd_node(const BlkNodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GSTAMP(a, 0);
  const int tid = threadIdx.x, lane = tid & 63, cw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  float* sX = smem;                          // [32][132] x_prev (= block_feats[b-1])
  float* sDZ = sX + 32 * LD128;              // [32][132] d_x tile: dz of block b, then dz of block b-1
  float* sRc = sDZ + 32 * LD128;             // [32][68]  d_rc          | post: q
  float* sRn = sRc + 32 * LD64;              // [32][68]  d_rn          | post: p (segment max)
  float* sRr = sRn + 32 * LD64;              // [32][36]  r             | post: dq [32][68] over Rr + Dr
  float* sDr = sRr + 32 * LD32;              // [32][36]  drpre
  float* sRrn = sDr + 32 * LD32;             // [32][36]  r_n (neighbor_feats)
  float* sPart = sRrn + 32 * LD32;           // [4][32][32] K-split partials | post: [2][32][64]
  float* sDrn = sRn;                         // [32][36]  drpre of the neighbour reduce FC, over d_rn once that is consumed
  float* sQ = sRc; float* sP = sRn; float* sDq = sRr; float* sR = sPart;
  f32x16 aWcn = zero16(), aWr = zero16(), aWrn = zero16(), aW4a = zero16(), aW4b = zero16(), aW3 = zero16();
  // bias gradients = column sums of the tiles: wave cw sums rows [8 cw, 8 cw + 8) of every tile, lane = column; the four
  // waves are folded once, after the last tile
  float gb1 = 0.f, gbr = 0.f, gbrn = 0.f, gb4 = 0.f, gb4b = 0.f, gb3 = 0.f;
  BtRegs<32> gW1, gWr, gWrn, gW3; BtRegs<64> gW4;
#define BN_LOAD_W1() load_bt<32>(gW1, a.w1 + (size_t)(32 + 32 * (cw & 1)) * D_P + 32 * (cw >> 1), D_P, lane)   /* role = (term, K half) */
#define BN_LOAD_WR() do { load_bt<D_R>(gWr, a.wr + (size_t)(32 * cw) * D_R, D_R, lane); if (NF) load_bt<D_R>(gWrn, a.wrn + (size_t)(32 * cw) * D_R, D_R, lane); } while (0)
#define BN_LOAD_W4() load_bt<64>(gW4, a.w4 + (size_t)(32 * (cw & 1)) * D_S + 64 * (cw >> 1), D_S, lane)        /* role = (column tile, K half) */
#define BN_LOAD_W3() load_bt<32>(gW3, a.w3 + (size_t)(32 * (cw & 1)) * D_P + 32 * (cw >> 1), D_P, lane)
  if (a.do_pre) { BN_LOAD_W1(); if (!ONE) BN_LOAD_WR(); }
  if (a.do_post && !ONE) { BN_LOAD_W4(); BN_LOAD_W3(); }
  if (ONE && !a.do_pre && a.do_post) BN_LOAD_W4();
  const int ntiles = (a.n_det + 31) / 32;
  NodeTileIn in;
  if ((int)blockIdx.x < ntiles) load_node_tile<NF>(in, a, blockIdx.x * 32, tid);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row0 = t * 32;                    // first detection of this tile
    __syncthreads();                              // the previous tile's readers are done
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
      *reinterpret_cast<float4*>(sX + row * LD128 + 4 * c4) = in.vx[j];
      *reinterpret_cast<float4*>(sDZ + row * LD128 + 4 * c4) = in.vz[j];
    }
    if (a.do_pre) {
      { const int row = tid >> 3, c4 = tid & 7;
        *reinterpret_cast<float4*>(sRr + row * LD32 + 4 * c4) = in.vr;
        if (NF) *reinterpret_cast<float4*>(sRrn + row * LD32 + 4 * c4) = in.vrn; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = tid + 256 * j, row = i >> 4, c4 = i & 15;
        *reinterpret_cast<float4*>(sRc + row * LD64 + 4 * c4) = in.vc[j];
        *reinterpret_cast<float4*>(sRn + row * LD64 + 4 * c4) = in.vn[j];
      }
    }
    // this tile's post-stage inputs (q, segment-max records) are requested HERE, behind the staging of `in`: requested
    // in front of it they sat between the tile's top loads and the (conservative: loop-carried, conditional loads) full
    // memory wait of the staging, which then exposed their round trip at the top of every tile
    NodeTileMid mid;
    load_node_mid(mid, a, row0, tid);
    // the next tile of this workgroup is requested now and lands while this one is computed
    if (!ONE && t + (int)gridDim.x < ntiles) load_node_tile<NF>(in, a, (t + gridDim.x) * 32, tid);
    __syncthreads();
    GSTAMP(a, 1);
    if (a.do_pre) {
      // dr = drc . Wc^T + drn . Wn^T : role = (term, K half)
      {
        const int term = cw & 1, kh = cw >> 1;
        f32x16 acc = zero16();
        mma_abt_r<32>(acc, (term ? sRn : sRc) + 32 * kh, LD64, gW1, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) sPart[(cw * 32 + crow(r, half)) * 32 + col] = acc[r];
      }
      if (ONE && (a.do_post || a.want_dx0)) BN_LOAD_WR();      // (used two barriers further down)
      // d Wc += r^T . drc ; d Wn += r_n^T . drn (r_n = r without neighbor_feats) : role = (term, column tile)
      {
        const int term = cw >> 1, nj = cw & 1;
        const float* Y = term ? sRn : sRc;
        const float* X = (term && NF) ? sRrn : sRr;
        dw_tile(aWcn, X + col, LD32, Y + 32 * nj + col, LD64, half);
      }
      gb1 += col_sum8(sRc + 8 * cw * LD64, LD64, lane);
      __syncthreads();
      for (int i = tid; i < 32 * D_R; i += 256) {
        const int row = i >> 5, ff = i & 31;
        // partials: roles 0 / 2 = drc . Wc^T (K halves), roles 1 / 3 = drn . Wn^T
        if (NF) {
          const float vcs = sPart[(0 * 32 + row) * 32 + ff] + sPart[(2 * 32 + row) * 32 + ff];
          const float vns = sPart[(1 * 32 + row) * 32 + ff] + sPart[(3 * 32 + row) * 32 + ff];
          sDr[row * LD32 + ff] = sRr[row * LD32 + ff] > 0.f ? vcs : 0.f;     // ReLU of reduce_dim
          sDrn[row * LD32 + ff] = sRrn[row * LD32 + ff] > 0.f ? vns : 0.f;   // ReLU of reduce_dim_neighbor
        } else {
          float v = sPart[(0 * 32 + row) * 32 + ff] + sPart[(1 * 32 + row) * 32 + ff];
          v += sPart[(2 * 32 + row) * 32 + ff];
          v += sPart[(3 * 32 + row) * 32 + ff];
          sDr[row * LD32 + ff] = sRr[row * LD32 + ff] > 0.f ? v : 0.f;    // ReLU of reduce_dim
        }
      }
      __syncthreads();
      // d Wr += x_prev^T . drpre : role cw owns rows [32 cw, 32 cw + 32) of Wr
      dw_tile(aWr, sX + 32 * cw + col, LD128, sDr + col, LD32, half);
      gbr += col_sum8(sDr + 8 * cw * LD32, LD32, col);
      if (NF) {
        dw_tile(aWrn, sX + 32 * cw + col, LD128, sDrn + col, LD32, half);
        gbrn += col_sum8(sDrn + 8 * cw * LD32, LD32, col);
      }
      if (a.do_post || a.want_dx0) {
        // d_x += drpre . Wr^T [+ drpre_n . Wrn^T] (columns [32 cw, 32 cw + 32))
        f32x16 acc = zero16();
        mma_abt_r<D_R>(acc, sDr, LD32, gWr, lane);
        if (NF) mma_abt_r<D_R>(acc, sDrn, LD32, gWrn, lane);
        if (ONE && a.do_post) BN_LOAD_W4();                    // (used behind the next two barriers)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = crow(r, half);
          const float v = sDZ[row * LD128 + 32 * cw + col] + acc[r];
          sDZ[row * LD128 + 32 * cw + col] = v;
          if (!a.do_post && row0 + row < a.n_det) a.d_x[(size_t)(row0 + row) * D_S + 32 * cw + col] = v;   // gradient wrt the start features
        }
      }
      __syncthreads();
    }
    GSTAMP(a, 2);
    if (a.do_post) {
      // q and the segment maxima -> LDS first: their registers are waited for BEFORE this stage's d_x stores are issued
      // (behind the stores the in-order memory counter would wait for the stores' acknowledgement as well)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = tid + 256 * j, row = i >> 4, c4 = i & 15;
        *reinterpret_cast<float4*>(sQ + row * LD64 + 4 * c4) = mid.vq[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j, row = i >> 6, ff = i & 63;
        sP[row * LD64 + ff] = __uint_as_float((unsigned)(mid.vpm[j] >> 32));
      }
      // dz = d_x * (x_out > 0): also the shortcut gradient of block b-1 (network.py:407-408)
      {
        float4 g[4], x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
          g[j] = *reinterpret_cast<const float4*>(sDZ + row * LD128 + 4 * c4);
          x[j] = *reinterpret_cast<const float4*>(sX + row * LD128 + 4 * c4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
          const float4 v = make_float4(x[j].x > 0.f ? g[j].x : 0.f, x[j].y > 0.f ? g[j].y : 0.f, x[j].z > 0.f ? g[j].z : 0.f, x[j].w > 0.f ? g[j].w : 0.f);
          *reinterpret_cast<float4*>(sDZ + row * LD128 + 4 * c4) = v;
          if (row0 + row < a.n_det) *reinterpret_cast<float4*>(a.d_x + (size_t)(row0 + row) * D_S + 4 * c4) = v;
        }
      }
      __syncthreads();
      // d W4 += q^T . dz : role cw owns output column tile cw
      dw_tile(aW4a, sQ + col, LD64, sDZ + 32 * cw + col, LD128, half);
      dw_tile(aW4b, sQ + 32 + col, LD64, sDZ + 32 * cw + col, LD128, half);
      gb4 += col_sum8(sDZ + 8 * cw * LD128, LD128, lane); gb4b += col_sum8(sDZ + 8 * cw * LD128, LD128, 64 + lane);
      // dq = (dz . W4^T) * (q > 0): role = (column tile, K half)
      {
        const int nt = cw & 1, kh = cw >> 1;
        f32x16 acc = zero16();
        mma_abt_r<64>(acc, sDZ + 64 * kh, LD128, gW4, lane);
        if (ONE) BN_LOAD_W3();                                 // (used behind the next two barriers)
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
      // d W3 += p^T . dq : role = (mi, nj)
      {
        const int mi = cw >> 1, nj = cw & 1;
        dw_tile(aW3, sP + 32 * mi + col, LD64, sDq + 32 * nj + col, LD64, half);
      }
      gb3 += col_sum8(sDq + 8 * cw * LD64, LD64, lane);
      // dp = dq . W3^T : role = (column tile, K half); the partials of the dq step were consumed before the last barrier
      {
        const int nt = cw & 1, kh = cw >> 1;
        f32x16 acc = zero16();
        mma_abt_r<32>(acc, sDq + 32 * kh, LD64, gW3, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) sR[(kh * 32 + crow(r, half)) * D_P + 32 * nt + col] = acc[r];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j, row = i >> 6, ff = i & 63;
        if (row0 + row < a.n_det) {
          const float dp = sR[row * D_P + ff] + sR[(32 + row) * D_P + ff];
          const unsigned cnt = (unsigned)(mid.vpm[j] & 0xffffffffull);
          a.d_pc[(size_t)(row0 + row) * D_P + ff] = dp / (float)cnt;                     // weighted_grads = grad / num_selected
        }
      }
    }
    GSTAMP(a, 3);
    if (ONE) break;
  }
#undef BN_LOAD_W1
#undef BN_LOAD_WR
#undef BN_LOAD_W4
#undef BN_LOAD_W3
  // ---- partial weight gradients of this workgroup
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  __syncthreads();
  float* sB = smem;                            // [4 waves][352]: b1 64 | br 32 | brn 32 | b4 128 | b3 64 | pad
  sB[cw * 352 + lane] = gb1;
  if (lane < 32) { sB[cw * 352 + 64 + lane] = gbr; sB[cw * 352 + 96 + lane] = gbrn; }
  sB[cw * 352 + 128 + lane] = gb4; sB[cw * 352 + 192 + lane] = gb4b;
  sB[cw * 352 + 256 + lane] = gb3;
  __syncthreads();
  for (int i = tid; i < 320; i += 256) {
    const float v = (sB[i] + sB[352 + i]) + (sB[2 * 352 + i] + sB[3 * 352 + i]);
    if (i < 64) { if (a.do_pre) ar[a.o_b1 + i] = v; }
    else if (i < 96) { if (a.do_pre) ar[a.o_br + i - 64] = v; }
    else if (i < 128) { if (a.do_pre && NF) ar[a.o_brn + i - 96] = v; }
    else if (i < 256) { if (a.do_post) ar[a.o_b4 + i - 128] = v; }
    else { if (a.do_post) ar[a.o_b3 + i - 256] = v; }
  }
  if (a.do_pre) {
    store_acc(ar + a.o_w1 + (size_t)(32 + 32 * (cw >> 1)) * D_P + 32 * (cw & 1), D_P, aWcn, lane);
    store_acc(ar + a.o_wr + (size_t)(32 * cw) * D_R, D_R, aWr, lane);
    if (NF) {
      store_acc(ar + a.o_wrn + (size_t)(32 * cw) * D_R, D_R, aWrn, lane);
    }
  }
  if (a.do_post) {
    store_acc(ar + a.o_w4 + 32 * cw, D_S, aW4a, lane);
    store_acc(ar + a.o_w4 + (size_t)32 * D_S + 32 * cw, D_S, aW4b, lane);
    store_acc(ar + a.o_w3 + (size_t)(32 * (cw >> 1)) * D_P + 32 * (cw & 1), D_P, aW3, lane);
  }
  GSTAMP(a, 15);
}

// ------------------------------------------------------------------------------------------
struct PwBwdArgs {
  int n_edge;
  const int* rows;           // ascending edge indices with a non-zero d_pw row
  const int* n_rows;         // their number (device)
  const float* pw; const float* d_pw; const float* h1; const float* h2;
  const float* w2; const float* w3;          // natural [256,256], [256,32]
  float* d_h1;
  float* arena; long long stride;
  long long o_w2, o_b2, o_w3, o_b3;
  GNET_TRACE_FIELD
};

// Asynchronous global -> LDS copy of a [32][256] fp32 tile whose rows are given by an index list (one 1 KB row per
// wave-instruction: the LDS destination of global_load_lds is wave-uniform base + lane * 16, exactly one padded row).  No staging
// registers; completion is tracked by vmcnt.  Rows past the list re-read its last row (finite data; their d3 rows are zero).
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
// entry p of the row list as (uniform base) + (32-bit offset); positions past the list re-read its last entry
#define PB_ROW(tile_, r_) (int)ldg_b(reinterpret_cast<const unsigned*>(a.rows), 4u * (unsigned)min((tile_) * 32 + (r_), n_rows - 1))

// ------------------------------------------------------------------------------------------
// pw_bwd_main (round 5, second form): ONE stream of 288 MFMAs per wave and tile behind ONE workgroup barrier (pw_fwd2's recipe, lesson 56).
// 32-row tile / 8 waves, one workgroup per CU looping over its tiles of the row list; dW2 (256 x 256) lives in 128 accumulator registers
// per wave across the whole range, dW3 in 16.  The first form had three phases per tile with all eight waves at the same point and the
// pipe mostly empty: the d3 stage (no MFMA), a light phase between its two barriers (32 MFMAs behind LDS round trips) and the 16 d_h1
// stores at the tile's end -- ~4.5 of a 19.7 us tile.  Here every light piece of the tiles t+1 / t+2 sits INSIDE the MFMAs of tile t:
//   top   dW3 += h2(t)^T . d3(t)              16 MFMAs (operands: registers + a d3 tile staged two tiles ago) while the first dW2 operands
//                                              come from LDS; between them the 16 stores of d h1(t-1), masked before the barrier
//   I-a   dW2 += h1(t)^T . d2(t)              128 MFMAs; woven: W3, the four h1 row copies of tile t+1 (LDS-DMA), the d3 sources of tile
//                                              t+2, the h2(t+1) requests, the row ids of tile t+3, the d3 stage of tile t+2, then
//         d2(t+1) = d3(t+1) . W3^T             16 MFMAs, its mask (h2(t+1) > 0), column sums (d b2) and LDS store; the first W2 rows
//   I-b   d h1(t) = d2(t) . W2^T              128 MFMAs, W2 rows streamed from L2 six k-steps ahead; woven: d b3
//   tail  ReLU mask of d h1(t) (h1(t) is overwritten during the next tile), barrier
// Rules it follows: (1) everything with HBM latency is requested in the LDS-fed half (I-a) -- memory returns in order, so a slow request
// in front of the W2 stream would hold every one of its operands back; (2) h2 never goes through LDS: a wave touches only ITS OWN 32
// columns of it (the mask of its d2 piece, the A operand of its dW3 rows), so it is 16 plain loads per lane (lane = column, register r =
// tile row crow(r, half): two 128-byte segments per instruction) and the k-steps of dW3 pair the rows (crow(kk, 0), crow(kk, 1)) -- the
// registers ARE the operand; (3) a buffer written during tile t is read only behind the next barrier and was last read before the
// previous one: h1 x 2 (DMA), d2 x 2, d3 x 3 (+ their column sums), row ids x 4 = 150 KB of LDS; (4) 256 registers, no spill in the loop
// (a reload is a scratch load + s_waitcnt vmcnt(0) in the middle of the copies in flight: the order of the pieces above is what fits).
// Same d2 / dW2 / d h1 instruction sequences as the first form (same bits); dW3 sums its rows in another order.
// Measured (A/B on one box, GNET_LIB_AB): 2.53-2.57 -> 2.46-2.50 ms.  Per-wave time stamps (GNET_TRACE build, wave 0 and wave 4 = the two
// waves of SIMD 0): the older wave of a SIMD wins every issue slot and runs ~4 us ahead of its partner; alternating s_setprio per MFMA
// group evens that out and changes nothing (the pair's total is what counts): the tile is 36.9 k cycles of MFMA + ~430 vector
// instructions + 350 LDS instructions per SIMD at the 2.2 GHz the chip sustains here.
constexpr int PBP_D = 2 * 32 * LD256;
constexpr int PBP_D3 = 4 * 32 * LD256;
constexpr int PBP_G3 = PBP_D3 + 3 * 32 * LD32;                // (three d3 tiles: t -- dW3 at the tile's top --, t+1, t+2)
constexpr int PBP_ROWS = PBP_G3 + 3 * 256;                    // int [4][2][32]: list entries | the same with the slack row past the list
constexpr size_t kPwBwdSmem = (size_t)(PBP_ROWS + 4 * 64) * sizeof(float);

template <bool BIG>
__global__ void __launch_bounds__(512) pw_bwd_main(const PwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sD3 = smem + PBP_D3;
  float* sG3 = smem + PBP_G3;
  int* sRows = reinterpret_cast<int*>(smem + PBP_ROWS);
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 aW2[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) aW2[0][j] = zero16();
  f32x16 aW3 = zero16();
  float gb2 = 0.f, gb3 = 0.f;
  const int n_rows = *a.n_rows;
  const int ntiles = (n_rows + 31) / 32;
  const int G = (int)gridDim.x;
  float pq0 = 0.f, pq1 = 0.f, dq0 = 0.f, dq1 = 0.f;
  int ra = 0, rb = 0, rs = 0;
  f32x4 w3f[4];
  float h2r[16];
  f32x16 d2n;
  // Every piece of the stream derives its LDS / global offsets from the lane id AFRESH (an opaque copy: the compiler cannot hoist them out
  // of the loop).  Kept across the loop they are ~25 registers beside 144 accumulators + 16 h2 values + the operand rings: the compiler
  // spilled nine of them, and a reload is a scratch load followed by s_waitcnt vmcnt(0) -- in the middle of the copies and stores in flight.
#define PBP_LDG4(base_, off_) (*reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base_) + (off_)))
#define PBP_LANE()                                                                                       \
  int lane = lane0;                                                       \
  const int tid = lane + 64 * wave, col = lane & 31, half = lane >> 5; (void)tid; (void)col; (void)half
#define PBP_LOAD_IDS(tile_)                                                                              \
  do { PBP_LANE(); ra = PB_ROW(tile_, tid >> 5); rb = PB_ROW(tile_, (tid >> 5) + 16); rs = PB_ROW(tile_, tid & 31); } while (0)
#define PBP_REQUEST_D3()                                                                                 \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    const unsigned oa_ = (unsigned)ra * (D_E * 4u) + 4u * (tid & 31), ob_ = (unsigned)rb * (D_E * 4u) + 4u * (tid & 31);   \
    pq0 = ldg_b(a.pw, oa_); dq0 = ldg_b(a.d_pw, oa_);                                                    \
    pq1 = ldg_b(a.pw, ob_); dq1 = ldg_b(a.d_pw, ob_);                                                    \
  } while (0)
  // d3 tile (ReLU of fc3 applied; rows past the list: zero) and its per-wave column sums (d b3) -> LDS
#define PBP_STAGE_D3(tile_, b_)                                                                          \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    const int e0_ = (tile_) * 32, row0_ = tid >> 5, j_ = tid & 31;                                       \
    const float va_ = (e0_ + row0_ < n_rows && pq0 > 0.f) ? dq0 : 0.f;                                   \
    const float vb_ = (e0_ + row0_ + 16 < n_rows && pq1 > 0.f) ? dq1 : 0.f;                              \
    sD3[(b_) * 32 * LD32 + row0_ * LD32 + j_] = va_;                                                     \
    sD3[(b_) * 32 * LD32 + (row0_ + 16) * LD32 + j_] = vb_;                                              \
    unsigned lo_, hi_; half_bcast(__float_as_uint(va_ + vb_), lo_, hi_);                                 \
    if (lane < 32) sG3[(b_) * 256 + wave * 32 + lane] = __uint_as_float(lo_) + __uint_as_float(hi_);     \
  } while (0)
  // the tile's list entries, and the same with the slack row for positions past the list (every thread: rs is the entry of row
  // tid & 31 -- sixteen threads store the same value, no branch)
#define PBP_STAGE_IDS(tile_, q_)                                                                         \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    sRows[(q_) * 64 + (tid & 31)] = rs; sRows[(q_) * 64 + 32 + (tid & 31)] = (tile_) * 32 + (tid & 31) < n_rows ? rs : a.n_edge;   \
  } while (0)
#define PBP_LOAD_W3()                                                                                    \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    const unsigned o_ = (unsigned)((32 * wave + col) * D_E + 4 * half) * 4u;                             \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) w3f[k_] = PBP_LDG4(a.w3, o_ + 32u * k_);             \
  } while (0)
  // the wave's 32 columns of the tile's h2 rows: lane = column, register r = tile row crow(r, half)
#define PBP_REQUEST_H2(rows_)                                                                            \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    const int* rp_ = (rows_) + 4 * half;                                                                 \
    const unsigned lo_ = (unsigned)(32 * wave + col) * 4u;                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                  \
      const unsigned rid_ = (unsigned)rp_[crow(r_, 0)];                                                  \
      if (BIG) h2r[r_] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.h2) + ((unsigned long long)rid_ << 10) + lo_);   \
      else h2r[r_] = ldg_b(a.h2, (rid_ << 10) + lo_);                                                    \
    }                                                                                                    \
  } while (0)
  // mask of the d2 piece (h2 > 0), its column sums (d b2), the piece -> LDS
#define PBP_FINISH_D2(D_)                                                                                \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) d2n[r_] = h2r[r_] > 0.f ? d2n[r_] : 0.f;            \
    float p_ = d2n[0];                                                                                   \
    _Pragma("unroll") for (int r_ = 1; r_ < 16; ++r_) p_ += d2n[r_];                                     \
    unsigned lo_, hi_; half_bcast(__float_as_uint(p_), lo_, hi_);                                        \
    gb2 += __uint_as_float(lo_) + __uint_as_float(hi_);                                                  \
    float* d_ = (D_) + (4 * half) * LD256 + 32 * wave + col;                                             \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) d_[crow(r_, 0) * LD256] = d2n[r_];                 \
  } while (0)
  // d b3: the per-wave column sums of a staged d3 tile (every thread forms the sum of column tid & 31; the first 32 keep it: no branch)
#define PBP_GB3(b_)                                                                                      \
  do {                                                                                                   \
    PBP_LANE();                                                                                          \
    float p_ = sG3[(b_) * 256 + (tid & 31)];                                                             \
    _Pragma("unroll") for (int w8_ = 1; w8_ < 8; ++w8_) p_ += sG3[(b_) * 256 + w8_ * 32 + (tid & 31)];   \
    gb3 += tid < D_E ? p_ : 0.f;                                                                         \
  } while (0)
  // d(fc2 pre) piece of this wave before its mask = d3 . W3^T
#define PBP_D2_MFMA(k_, q_, av_) d2n = __builtin_amdgcn_mfma_f32_32x32x2f32((av_)[q_], w3f[k_][q_], d2n, 0, 0, 0)

  // d h1 of the previous tile: masked before the barrier, STORED at the next tile's top between the dW3 MFMAs
#define PBP_STORE_DH1(rows_)                                                                             \
  do {                                                                                                   \
    const int* rp_ = (rows_) + 4 * half;                                                                 \
    const unsigned lo_ = (unsigned)(32 * wave + col) * 4u;                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                  \
      if (BIG) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.d_h1) + ((unsigned long long)(unsigned)rp_[crow(r_, 0)] << 10) + lo_) = acc[r_];   \
      else stg_b(a.d_h1, ((unsigned)rp_[crow(r_, 0)] << 10) + lo_, acc[r_]);                             \
    }                                                                                                    \
  } while (0)
  if ((int)blockIdx.x < ntiles) {
    // ---- prologue: h1(t0) on its way, d3(t0) and d3(t0 + G) staged, then d2(t0)
    const int t0 = (int)blockIdx.x;
    f32x16 acc = zero16();                                    // (the first tile's "previous tile": zeros to the slack row)
    PBP_LOAD_IDS(t0);
    dma_rows32(smem, a.h1, a.rows, t0 * 32, n_rows, wave, lane0);
    PBP_LOAD_W3();
    PBP_REQUEST_D3();
    PBP_STAGE_D3(t0, 0);
    PBP_STAGE_IDS(t0, 0);
    { PBP_LANE(); sRows[3 * 64 + 32 + (tid & 31)] = a.n_edge; }
    PBP_LOAD_IDS(t0 + G);
    PBP_REQUEST_D3();
    PBP_STAGE_D3(t0 + G, 1);
    PBP_STAGE_IDS(t0 + G, 1);
    PBP_LOAD_IDS(t0 + 2 * G);
    __syncthreads();
    PBP_REQUEST_H2(sRows);
    {
      PBP_LANE();
      const float* ap3 = sD3 + col * LD32 + 4 * half;
      d2n = zero16();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap3 + 8 * k);
        PBP_D2_MFMA(k, 0, av); PBP_D2_MFMA(k, 1, av); PBP_D2_MFMA(k, 2, av); PBP_D2_MFMA(k, 3, av);
      }
    }
    PBP_FINISH_D2(smem + PBP_D);
    PBP_GB3(0);
    drain_vmem_before_loop();
    __syncthreads();

    int it = 0, i3 = 0;                                       // i3 = it % 3: d3(t) lives in buffer i3, d3(t+1) in the next, d3(t+2) is staged
    for (int t = t0; t < ntiles; t += G, ++it, i3 = i3 == 2 ? 0 : i3 + 1) {
      const int p = it & 1;
      const int i3n = i3 == 2 ? 0 : i3 + 1, i3w = i3n == 2 ? 0 : i3n + 1;
      const float* H1c = smem + p * 32 * LD256;
      float* H1n = smem + (p ^ 1) * 32 * LD256;
      const float* Dc = smem + PBP_D + p * 32 * LD256;
      float* Dn = smem + PBP_D + (p ^ 1) * 32 * LD256;
      const float* D3c = sD3 + i3 * 32 * LD32;
      const float* D3n = sD3 + i3n * 32 * LD32;
      const int* rowsN = sRows + ((it + 1) & 3) * 64;
      const int* rowsP = sRows + ((it + 3) & 3) * 64 + 32;    // the previous tile's rows, the slack row past the list
      if (it == 5) { GSTAMP(a, 0); GSTAMP_W(a, 8, 256); }
      constexpr int NS = D_H / 8, PF = 6;
      f32x4 ring[PF];                                         // W2 rows of phase I-b, six k-steps ahead
      // ---- I-a: dW2 += h1^T . d2 (rows [32w, 32w+32) of W2, all 256 columns)
      {
        PBP_LANE();
        const int4 drv = *reinterpret_cast<const int4*>(rowsN + 4 * wave);      // the four rows of tile t+1 this wave copies (uniform)
        const int dr[4] = {drv.x, drv.y, drv.z, drv.w};
        const float* X = H1c + 32 * wave + half * LD256 + col; const float* Y = Dc + half * LD256 + col;
        float xa = X[0], yb[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) yb[n] = Y[32 * n];
        // ---- II-b of this tile (its d2 was formed during the previous tile): dW3 += h2(t)^T . d3(t) -- operands in registers and in a d3
        // buffer staged two tiles ago, so its 16 MFMAs run while the first dW2 operands are on their way from LDS; between them the
        // 16 stores of the PREVIOUS tile's d h1 (masked before the barrier)
        {
          const float* b3p = D3c + (4 * half) * LD32 + col;
          PBP_STORE_DH1(rowsP);
#pragma unroll
          for (int kk = 0; kk < 16; kk += 4) {
            float b3[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b3[q] = b3p[crow(kk + q, 0) * LD32];
#pragma unroll
            for (int q = 0; q < 4; ++q) aW3 = __builtin_amdgcn_mfma_f32_32x32x2f32(h2r[kk + q], b3[q], aW3, 0, 0, 0);
          }
        }
        if (it == 5) { GSTAMP(a, 1); GSTAMP_W(a, 9, 256); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const int row = 2 * (kk + 1 < 16 ? kk + 1 : kk);
          const float xn = X[row * LD256];
          float yn[8];
#pragma unroll
          for (int n = 0; n < 8; ++n) yn[n] = Y[row * LD256 + 32 * n];
          // everything with HBM latency is requested HERE, in the LDS-fed half of the stream: memory returns in order, so a slow request
          // in front of the W2 stream of phase I-b would hold every one of its operands back
          if (kk == 8 && it == 5) { GSTAMP(a, 2); GSTAMP_W(a, 10, 256); }
          if (kk == 12 && it == 5) { GSTAMP(a, 3); GSTAMP_W(a, 11, 256); }
          if (kk == 14 && it == 5) { GSTAMP(a, 4); GSTAMP_W(a, 12, 256); }
          if (kk == 0) PBP_LOAD_W3();                       // (in front of the copies)
          if (kk < 4) {
            const int rq = __builtin_amdgcn_readfirstlane(dr[kk]);
            const char* src = reinterpret_cast<const char*>(a.h1 + (size_t)rq * D_H) + 16u * (unsigned)lane;
            float* dst = H1n + (wave * 4 + kk) * LD256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
          }
          if (kk == 4) { PBP_REQUEST_D3(); PBP_STAGE_IDS(t + 2 * G, (it + 2) & 3); }   // sources of tile t+2 (ids loaded during tile t-1)
          if (kk == 5) PBP_REQUEST_H2(rowsN);
          if (kk == 6) PBP_LOAD_IDS(t + 3 * G);
          if (kk == 12) {                                   // II-a: d2(t+1) = d3(t+1) . W3^T
            const float* ap3 = D3n + col * LD32 + 4 * half;
            d2n = zero16();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x4 av = *reinterpret_cast<const f32x4*>(ap3 + 8 * k);
              PBP_D2_MFMA(k, 0, av); PBP_D2_MFMA(k, 1, av); PBP_D2_MFMA(k, 2, av); PBP_D2_MFMA(k, 3, av);
            }
          }
          if (kk == 12) PBP_STAGE_D3(t + 2 * G, i3w);
          if (kk == 13) PBP_FINISH_D2(Dn);                  // its mask (h2(t+1), requested at kk = 5), column sums, the piece -> LDS
          if (kk >= 14) {
            const unsigned o_ = (unsigned)((32 * wave + col) * D_H + 4 * half) * 4u;
#pragma unroll
            for (int i = 3 * (kk - 14); i < 3 * (kk - 13); ++i) ring[i] = PBP_LDG4(a.w2, o_ + 32u * i);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n = 0; n < 8; ++n) aW2[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, yb[n], aW2[0][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          xa = xn;
#pragma unroll
          for (int n = 0; n < 8; ++n) yb[n] = yn[n];
        }
      }
      if (it == 5) { GSTAMP(a, 5); GSTAMP_W(a, 13, 256); }
      // ---- I-b: d(fc1 pre) tile = d2 . W2^T, the W2 rows streamed from L2 six k-steps ahead
      acc = zero16();
      {
        PBP_LANE();
        const float* ap = Dc + col * LD256 + 4 * half;
        const unsigned o_ = (unsigned)((32 * wave + col) * D_H + 4 * half) * 4u;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f32x4 b = ring[s % PF];
          if (s + PF < NS) ring[s % PF] = PBP_LDG4(a.w2, o_ + 32u * (s + PF));
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 8 * s);
          if (s == 16 && it == 5) { GSTAMP(a, 6); GSTAMP_W(a, 14, 256); }
          if (s == 12) PBP_GB3(i3n);
          __builtin_amdgcn_sched_barrier(0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (it == 5) { GSTAMP(a, 7); GSTAMP_W(a, 15, 256); }
      // ---- tail: the ReLU mask of d h1(t) (h1(t) is overwritten during the next tile); its stores follow behind the barrier
      {
        PBP_LANE();
        const float* hp = H1c + (4 * half) * LD256 + 32 * wave + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = hp[crow(r, 0) * LD256] > 0.f ? acc[r] : 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the h1 copies of tile t+1 (every later load has been consumed; the stores are a tile old)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    {                                                         // the last tile's d h1
      PBP_LANE();
      PBP_STORE_DH1(sRows + ((it + 3) & 3) * 64 + 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef PBP_STORE_DH1
  const int lane = lane0, tid = threadIdx.x, col = lane & 31;
#undef PBP_LANE
#undef PBP_LDG4
#undef PBP_LOAD_IDS
#undef PBP_REQUEST_D3
#undef PBP_STAGE_D3
#undef PBP_STAGE_IDS
#undef PBP_LOAD_W3
#undef PBP_REQUEST_H2
#undef PBP_FINISH_D2
#undef PBP_D2_MFMA
#undef PBP_GB3
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
#pragma unroll
  for (int j = 0; j < 8; ++j) store_acc(ar + a.o_w2 + (size_t)(32 * wave) * D_H + 32 * j, D_H, aW2[0][j], lane);
  store_acc(ar + a.o_w3 + (size_t)(32 * wave) * D_E, D_E, aW3, lane);
  if (lane < 32) ar[a.o_b2 + 32 * wave + col] = gb2;
  if (tid < D_E) ar[a.o_b3 + tid] = gb3;
}

// ------------------------------------------------------------------------------------------
// pw_bwd_bf (round 6): pw_bwd_main ON THE bf16 PIPE.  Every fp32 product of its four GEMMs -- d2 = d3 . W3^T, dW3 += h2^T . d3,
// dW2 += h1^T . d2, d h1 = d2 . W2^T -- is six bf16 products of exact three-term splits (common.hpp mma6's sequence,
// v_mfma_f32_32x32x16_bf16, fp32 accumulation): 216 MFMAs of 32 cycles per wave and tile instead of 288 of 64.  Same decomposition of
// the row list (32-row tiles, 8 waves, one workgroup per CU walking its tiles; dW2 in 128 accumulator registers per wave across the
// whole range, dW3 in 16), but the wave now owns COLUMNS [32 w, 32 w + 32) of d2 / W2 and all 256 rows of dW2 (8 blocks), because the
// B operand of dW2 -- the wave's own d2 piece -- then never leaves its registers.  Operands:
//   * weight-gradient products contract over the 32 tile rows, i.e. both operands want "lane = feature, eight k-slots = eight rows".
//     h2 is loaded that way (16 plain loads per lane: lane = column, register r = row crow(r, half) -- as pw_bwd_main) and h1 too: the
//     tile's rows are copied to an fp32 staging tile by LDS-DMA (no registers while in flight), and wave w reads ITS 32 columns of
//     it, keeps nothing but splits them into three bf16 terms and publishes them as the A-operand fragments of block w
//     (H1F[term][k-step][block][lane] x 16 bytes).  d2 (lane = column, registers = rows, masked by h2 > 0) is split in
//     registers: those ARE the B operand of dW2;
//   * d h1 = d2 . W2^T contracts over d2's 256 columns: its A operand wants "lane = row, eight k-slots = eight columns", the
//     transpose of what the wave holds -- the three terms are scattered into D2F[term][k-step][row] x 16 bytes as 48 two-byte
//     LDS stores per lane (the slot order of a k-step is frag_feat's, as in pw_fwd3), the B operand (W2 rows, three terms,
//     pack_pw_bf16's W2D) streams from L2 through a two-deep ring that is reloaded in place;
//   * d3 (32 x 32 per tile) is staged once in both fragment shapes (lane = row for d2, lane = column for dW3).
// TWO LDS-only barriers per tile: phase A (d2, dW3, dW2: reads H1F, writes D2F) | phase B (d h1: reads D2F; stages d3 and H1F of
// the next tile).  Every request with HBM latency (the h1 copies, h2, the d3 sources, row ids) is issued in phase A, whose operands
// come from LDS; phase B's W2 stream has nothing slow in front of it.
// The ReLU mask of d h1 comes from the high terms of the wave's own h1 fragments (h1 >= 0: positive <=> high term non-zero).
constexpr int PBB_H1F = 0;                              // 32-bit words: [3][2][8][64][4]
constexpr int PBB_H1F_T = 2 * 8 * 64 * 4;               //   words between the terms
constexpr int PBB_D2F = PBB_H1F + 3 * PBB_H1F_T;        // bytes: term * 17408 + k-step * 1088 + half * 544 + row * 16 (+ slot * 2): the pads
constexpr int PBB_D2F_TB = 16 * 1088;                   //   spread the two-byte scatter over the banks, the 16-byte reads stay aligned
constexpr int PBB_S = PBB_D2F + 3 * PBB_D2F_TB / 4;     // fp32 [32][260]: the next tile's h1 rows (LDS-DMA)
constexpr int PBB_D3A = PBB_S + 32 * LD256;             // [3][2][64][4]: d3, lane = row, slots = columns
constexpr int PBB_D3B = PBB_D3A + 3 * 2 * 64 * 4;       // [3][2][64][4]: d3, lane = column, slots = rows
constexpr int PBB_ROWS = PBB_D3B + 3 * 2 * 64 * 4;      // int [4][64]: list entries | the same with the slack row past the list
constexpr int PBB_RED = PBB_ROWS + 4 * 64;              // fp32 [32][32]: d b3 partials (epilogue)
constexpr size_t kPwBwdBfSmem = (size_t)(PBB_RED + 32 * 32) * 4;
static_assert(kPwBwdBfSmem <= 160 * 1024, "pw_bwd_bf LDS");

struct PwBwdBfArgs {
  PwBwdArgs p;
  const unsigned* wbf;       // pack_pw_bf16's arrays (common.hpp PWBF_*): W2D, W3D
};

#ifndef PBB_RD
#define PBB_RD 2     /* depth of the W2 ring of phase B in k-steps (3 measured equal, 4 spills) */
#endif
#ifdef PBB_TRACE2
#define PBB_T2(slot_) do { if (it == 5) GSTAMP(a, slot_); } while (0)
#define PBB_W(slot_) ((void)0)
#else
#define PBB_T2(slot_) ((void)0)
#define PBB_W(slot_) GSTAMP_W(a, slot_, 256)
#endif
#ifdef PBB_NOSB
#define PBB_SB() ((void)0)
#else
#define PBB_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef PBB_DB
#define PBB_DB 0         /* 1: the dW2 loop's A fragments fully double-buffered (measurement) */
#endif
#ifndef PBB_PRIME
#define PBB_PRIME 16     /* step of the dW2 loop in front of which phase B's W2 ring is primed (16 = behind the loop) */
#endif
#ifndef PBB_X
#define PBB_X 0      /* ablation mask of measurement builds (tools/pw_ablate.sh): the shipped kernel is PBB_X == 0 */
#endif
template <bool BIG>
__global__ void __launch_bounds__(512) pw_bwd_bf(const PwBwdBfArgs aa) {
  const PwBwdArgs& a = aa.p;
  extern __shared__ __attribute__((aligned(16))) unsigned smb[];
  float* sS = reinterpret_cast<float*>(smb + PBB_S);
  int* sRows = reinterpret_cast<int*>(smb + PBB_ROWS);
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  GSTAMP(a, 0);
  f32x16 aW2[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) aW2[m] = zero16();
  f32x16 aW3 = zero16();
  float gb2 = 0.f, g3x = 0.f, g3y = 0.f;
  const int n_rows = *a.n_rows;
  const int ntiles = (n_rows + 31) / 32;
  const int G = (int)gridDim.x;
  int rs = 0;
  float h2r[16];
  Bf3 w3q[2];                                            // the wave's W3^T fragments (B operand of d2): re-requested at the end of every phase B
  float2 pq = make_float2(0.f, 0.f), dq = make_float2(0.f, 0.f);
  const unsigned* w2d = aa.wbf + PWBF_W2D;               // [3][8][16][64][4]
  const unsigned* w3d = aa.wbf + PWBF_W3D;               // [3][8][2][64][4]
  constexpr unsigned W2D_T = 8u * 16 * 64 * 16, W3D_T = 8u * 2 * 64 * 16;      // bytes between the terms

  // every piece derives its offsets from an opaque copy of the lane id: nothing address-like is hoisted out of the tile loop (and spilled)
#define PBB_LANE()                                                                                       \
  int lane = lane0; asm volatile("" : "+v"(lane));                                                       \
  const int tid = lane + 64 * wave, col = lane & 31, half = lane >> 5; (void)tid; (void)col; (void)half
#define PBB_LDQ(base_, off_) (*reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base_) + (off_)))
#define PBB_LDS_Q(word_) (*reinterpret_cast<const u32x4*>(smb + (word_)))
#define PBB_LOAD_IDS(tile_) do { PBB_LANE(); rs = PB_ROW(tile_, tid & 31); } while (0)
#define PBB_STAGE_IDS(tile_, q_)                                                                         \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    sRows[(q_) * 64 + (tid & 31)] = rs; sRows[(q_) * 64 + 32 + (tid & 31)] = (tile_) * 32 + (tid & 31) < n_rows ? rs : a.n_edge;   \
  } while (0)
  // d3 sources of a tile: thread (row = tid >> 4, column pair = tid & 15); the row id comes from the staged list
#define PBB_REQUEST_D3(rows_)                                                                            \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const unsigned o_ = (unsigned)(rows_)[tid >> 4] * (D_E * 4u) + 8u * (tid & 15);                      \
    pq = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.pw) + o_);                     \
    dq = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.d_pw) + o_);                   \
  } while (0)
  // d3 = d_pw where fc3's output is positive (rows past the list: zero), split, both fragment shapes -> LDS; d b3 partials
#define PBB_STAGE_D3(tile_)                                                                              \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const int row_ = tid >> 4, op_ = tid & 15;                                                           \
    const bool in_ = (tile_) * 32 + row_ < n_rows;                                                       \
    const float v0_ = (in_ && pq.x > 0.f) ? dq.x : 0.f, v1_ = (in_ && pq.y > 0.f) ? dq.y : 0.f;          \
    g3x += v0_; g3y += v1_;                                                                              \
    unsigned ph_, pm_, pl_; split3_pk(v0_, v1_, ph_, pm_, pl_);                                          \
    unsigned* da_ = smb + PBB_D3A + (((op_ >> 3) * 64 + ((op_ >> 2) & 1) * 32 + row_) * 4 + (op_ & 3));  \
    da_[0] = ph_; da_[2 * 64 * 4] = pm_; da_[2 * 2 * 64 * 4] = pl_;                                      \
    unsigned short* db_ = reinterpret_cast<unsigned short*>(smb + PBB_D3B) +                             \
        (((row_ >> 4) * 64 + ((row_ >> 2) & 1) * 32 + 2 * op_) * 8 + (row_ & 3) + 4 * ((row_ >> 3) & 1)); \
    db_[0] = (unsigned short)ph_; db_[8] = (unsigned short)(ph_ >> 16);                                  \
    db_[2 * 64 * 8] = (unsigned short)pm_; db_[2 * 64 * 8 + 8] = (unsigned short)(pm_ >> 16);            \
    db_[2 * 2 * 64 * 8] = (unsigned short)pl_; db_[2 * 2 * 64 * 8 + 8] = (unsigned short)(pl_ >> 16);    \
  } while (0)
  // the wave's 32 columns of a tile's h2 rows: lane = column, register r = tile row crow(r, half)
#define PBB_REQUEST_H2(rows_)                                                                            \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const int* rp_ = (rows_) + 4 * half;                                                                 \
    const unsigned lo_ = (unsigned)(32 * wave + col) * 4u;                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                  \
      const unsigned rid_ = (unsigned)rp_[crow(r_, 0)];                                                  \
      if (BIG) h2r[r_] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.h2) + ((unsigned long long)rid_ << 10) + lo_);   \
      else h2r[r_] = ldg_b(a.h2, (rid_ << 10) + lo_);                                                    \
    }                                                                                                    \
  } while (0)
  // the four h1 rows of a tile this wave copies into the staging tile (one 1 KB row per wave-instruction)
#define PBB_DMA_H1(rows_)                                                                                \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const int4 drv_ = *reinterpret_cast<const int4*>((rows_) + 4 * wave);                                \
    const int dr_[4] = {drv_.x, drv_.y, drv_.z, drv_.w};                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                                   \
      const int rq_ = __builtin_amdgcn_readfirstlane(dr_[k_]);                                           \
      const char* src_ = reinterpret_cast<const char*>(a.h1 + (size_t)rq_ * D_H) + 16u * (unsigned)lane; \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                                       (__attribute__((address_space(3))) void*)(sS + (wave * 4 + k_) * LD256), 16, 0, 0);   \
    }                                                                                                    \
  } while (0)
  // the wave's 32 columns of the staged h1 tile -> three bf16 terms -> its block of H1F (k-step q = registers 8 q .. 8 q + 7)
#define PBB_STAGE_H1()                                                                                   \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const float* sp_ = sS + (4 * half) * LD256 + 32 * wave + col;                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                   \
      float v_[8];                                                                                       \
      _Pragma("unroll") for (int t_ = 0; t_ < 8; ++t_) v_[t_] = sp_[crow(8 * q_ + t_, 0) * LD256];       \
      const Bf3 f_ = split3_8(f32x4{v_[0], v_[1], v_[2], v_[3]}, f32x4{v_[4], v_[5], v_[6], v_[7]});     \
      unsigned* d_ = smb + PBB_H1F + ((q_ * 8 + wave) * 64 + lane) * 4;                                  \
      *reinterpret_cast<u32x4*>(d_) = f_.h; *reinterpret_cast<u32x4*>(d_ + PBB_H1F_T) = f_.m;            \
      *reinterpret_cast<u32x4*>(d_ + 2 * PBB_H1F_T) = f_.l;                                              \
    }                                                                                                    \
  } while (0)
#define PBB_STORE_DH1(rows_)                                                                             \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    const int* rp_ = (rows_) + 4 * half;                                                                 \
    const unsigned lo_ = (unsigned)(32 * wave + col) * 4u;                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                  \
      if (BIG) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.d_h1) + ((unsigned long long)(unsigned)rp_[crow(r_, 0)] << 10) + lo_) = acc[r_];   \
      else stg_b(a.d_h1, ((unsigned)rp_[crow(r_, 0)] << 10) + lo_, acc[r_]);                             \
    }                                                                                                    \
  } while (0)
#define PBB_LOAD_W3()                                                                                    \
  do {                                                                                                   \
    PBB_LANE();                                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                   \
      const unsigned o_ = (unsigned)((wave * 2 + q_) * 64 + lane) * 16u;                                 \
      w3q[q_].h = PBB_LDQ(w3d, o_); w3q[q_].m = PBB_LDQ(w3d, W3D_T + o_); w3q[q_].l = PBB_LDQ(w3d, 2 * W3D_T + o_);   \
    }                                                                                                    \
  } while (0)
#define PBB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  if ((int)blockIdx.x < ntiles) {
    const int t0 = (int)blockIdx.x;
    // ---- prologue: row ids of the first two tiles, d3(t0), h1(t0) as fragments, h2(t0) on its way
    PBB_LOAD_IDS(t0);
    PBB_STAGE_IDS(t0, 0);
    PBB_LOAD_IDS(t0 + G);
    PBB_STAGE_IDS(t0 + G, 1);
    PBB_LOAD_IDS(t0 + 2 * G);
    __syncthreads();
    PBB_DMA_H1(sRows);
    PBB_REQUEST_D3(sRows);
    PBB_REQUEST_H2(sRows);
    PBB_STAGE_D3(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PBB_STAGE_H1();
    PBB_LOAD_W3();
    drain_vmem_before_loop();
    __syncthreads();

    int it = 0;
    for (int t = t0; t < ntiles; t += G, ++it) {
      const int* rowsC = sRows + (it & 3) * 64 + 32;          // this tile's rows, the slack row past the list (d h1 stores)
      const int* rowsN = sRows + ((it + 1) & 3) * 64;         // the next tile's rows
      if (it == 5) { GSTAMP(a, 1); PBB_W(8); }
      // ======== phase A ========
      // the slow requests of the next tile, first: its four h1 rows (LDS-DMA) and its d3 sources
      PBB_DMA_H1(rowsN);
      PBB_T2(8);
      PBB_REQUEST_D3(rowsN);
      PBB_T2(9);
      PBB_STAGE_IDS(t + 2 * G, (it + 2) & 3);
      PBB_LOAD_IDS(t + 3 * G);
      PBB_T2(10);
      Bf3 bd2[2];
      unsigned long long hm[16];                              // lane masks (scalar registers), one per accumulator register of phase B
      {
        PBB_LANE();
        // the ReLU mask of d h1: positive h1 <=> its high term is non-zero (h1 >= 0); the wave's own block of H1F, slot t of k-step q = row
        // crow(8 q + t, half) -- the register order of phase B's accumulators.  Sixteen lane masks in scalar registers for the length of
        // the tile, applied by one v_cndmask each.  (Taken from the loop's own operand registers under `if (m == wave)` this
        // compiled to ~400 scalar mask merges per tile; as sixteen `bool`s the compiler kept the eight fragment words in vector
        // registers instead and spilled four of them.)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x4 hh = *reinterpret_cast<const u32x4*>(smb + PBB_H1F + lane * 4 + (q * 8 + wave) * 256);
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) { hm[8 * q + 2 * w4] = __ballot((hh[w4] & 0xffffu) != 0u); hm[8 * q + 2 * w4 + 1] = __ballot((hh[w4] >> 16) != 0u); }
        }
      }
      constexpr int RD = PBB_RD;                              // depth of the W2 ring in k-steps
      u32x4 rh[RD], rm[RD], rl[RD];
      {
        PBB_LANE();
        // ---- d2 = d3 . W3^T (this wave's 32 columns; W3's fragments were requested at the end of the previous phase B), masked by
        // h2 > 0; d b2; its three terms = the B operand of dW2
        f32x16 d2 = zero16();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          Bf3 da;
          da.h = PBB_LDS_Q(PBB_D3A + (q * 64 + lane) * 4); da.m = PBB_LDS_Q(PBB_D3A + 2 * 64 * 4 + (q * 64 + lane) * 4);
          da.l = PBB_LDS_Q(PBB_D3A + 2 * 2 * 64 * 4 + (q * 64 + lane) * 4);
          d2 = mma6(d2, da, w3q[q]);
        }
        PBB_T2(11);
        float p_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d2[r] = h2r[r] > 0.f ? d2[r] : 0.f; p_ += d2[r]; }
        gb2 += p_;
        PBB_T2(12);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          bd2[q] = split3_8(f32x4{d2[8 * q], d2[8 * q + 1], d2[8 * q + 2], d2[8 * q + 3]}, f32x4{d2[8 * q + 4], d2[8 * q + 5], d2[8 * q + 6], d2[8 * q + 7]});
        PBB_SB();
      }
      if (it == 5) { GSTAMP(a, 2); PBB_W(9); }
      {
        // ---- dW2 += h1^T . d2: 8 blocks x 2 k-steps of six products; the A fragments from LDS (high term a step ahead, the low and
        // the middle term reloaded in place behind their last product).
        // Woven into its 16 steps, six of the 48 two-byte stores of d2's TRANSPOSE per step in the first eight (phase B's A operand:
        // the terms of (row, column 32 w + col) go to D2F[term][k-step][row], slot frag^-1(col)), then dW3 += h2^T . d3 (rows
        // [32 w, 32 w + 32) of W3: A = the h2 registers' terms, B = d3 with lane = column) and the next tile's h2 requests
        PBB_LANE();
        const unsigned* hf = smb + PBB_H1F + lane * 4;
        unsigned short* sc_ = reinterpret_cast<unsigned short*>(smb + PBB_D2F) +
            ((2 * wave + (col >> 4)) * 1088 + ((col >> 2) & 1) * 544 + half * 64) / 2 + (col & 3) + 4 * ((col >> 3) & 1);
#if PBB_DB
        u32x4 ah[2], amv[2], alv[2];
        ah[0] = *reinterpret_cast<const u32x4*>(hf); amv[0] = *reinterpret_cast<const u32x4*>(hf + PBB_H1F_T); alv[0] = *reinterpret_cast<const u32x4*>(hf + 2 * PBB_H1F_T);
#else
        u32x4 ah[2], am, al;
        ah[0] = *reinterpret_cast<const u32x4*>(hf); am = *reinterpret_cast<const u32x4*>(hf + PBB_H1F_T); al = *reinterpret_cast<const u32x4*>(hf + 2 * PBB_H1F_T);
#endif
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int q = i >> 3, m = i & 7;
          const int qn = (i + 1) >> 3, mn = (i + 1) & 7;
          const unsigned* hn = hf + (qn * 8 + mn) * 256;
          if (i < 8 && !(PBB_X & 1)) {
            const int sq = i >> 2, p2 = i & 3;
            const int ro = (((2 * p2) & 3) + 8 * (2 * sq + ((2 * p2) >> 2))) * 8;      // (row crow(8 sq + 2 p2, 0)) * 16 bytes, in shorts
            sc_[ro] = (unsigned short)bd2[sq].h[p2]; sc_[ro + 8] = (unsigned short)(bd2[sq].h[p2] >> 16);
            sc_[PBB_D2F_TB / 2 + ro] = (unsigned short)bd2[sq].m[p2]; sc_[PBB_D2F_TB / 2 + ro + 8] = (unsigned short)(bd2[sq].m[p2] >> 16);
            sc_[PBB_D2F_TB + ro] = (unsigned short)bd2[sq].l[p2]; sc_[PBB_D2F_TB + ro + 8] = (unsigned short)(bd2[sq].l[p2] >> 16);
          }
          if ((i == 8 || i == 10) && !(PBB_X & 2)) {
            const int q3 = (i - 8) >> 1;
            const Bf3 ha = split3_8(f32x4{h2r[8 * q3], h2r[8 * q3 + 1], h2r[8 * q3 + 2], h2r[8 * q3 + 3]},
                                    f32x4{h2r[8 * q3 + 4], h2r[8 * q3 + 5], h2r[8 * q3 + 6], h2r[8 * q3 + 7]});
            Bf3 db;
            db.h = PBB_LDS_Q(PBB_D3B + (q3 * 64 + lane) * 4); db.m = PBB_LDS_Q(PBB_D3B + 2 * 64 * 4 + (q3 * 64 + lane) * 4);
            db.l = PBB_LDS_Q(PBB_D3B + 2 * 2 * 64 * 4 + (q3 * 64 + lane) * 4);
            PBB_SB();
            aW3 = mma6(aW3, ha, db);
          }
          if (i == 11 && !(PBB_X & 32)) PBB_REQUEST_H2(rowsN);   // (the registers of h2(t) are free: the next tile's, a whole tile ahead)
          if (i == PBB_PRIME) {
            // phase B's W2 ring is primed HERE, behind every slow request of the tile: the s_waitcnt in front of barrier A then
            // allows exactly these 3 RD loads to be outstanding -- the h1 copies, h2 and the d3 sources, all older, have landed
            const unsigned o_ = (unsigned)(wave * 16 * 64 + lane) * 16u;
#pragma unroll
            for (int k2 = 0; k2 < RD; ++k2) { rh[k2] = PBB_LDQ(w2d, o_ + 1024u * k2); rm[k2] = PBB_LDQ(w2d, W2D_T + o_ + 1024u * k2); rl[k2] = PBB_LDQ(w2d, 2 * W2D_T + o_ + 1024u * k2); }
          }
          PBB_SB();
#if PBB_DB
          if (i + 1 < 16) {
            alv[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(hn + 2 * PBB_H1F_T); ah[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(hn);
            amv[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(hn + PBB_H1F_T);
          }
          const u32x4 al = alv[i & 1], am = amv[i & 1];
          aW2[m] = mfma_bf16(al, bd2[q].h, aW2[m]);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].l, aW2[m]);
          aW2[m] = mfma_bf16(am, bd2[q].m, aW2[m]);
          aW2[m] = mfma_bf16(am, bd2[q].h, aW2[m]);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].m, aW2[m]);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].h, aW2[m]);
#else
          aW2[m] = mfma_bf16(al, bd2[q].h, aW2[m]);
          if (i + 1 < 16) al = *reinterpret_cast<const u32x4*>(hn + 2 * PBB_H1F_T);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].l, aW2[m]);
          if (i + 1 < 16) ah[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(hn);
          aW2[m] = mfma_bf16(am, bd2[q].m, aW2[m]);
          aW2[m] = mfma_bf16(am, bd2[q].h, aW2[m]);
          PBB_SB();
          if (i + 1 < 16) am = *reinterpret_cast<const u32x4*>(hn + PBB_H1F_T);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].m, aW2[m]);
          aW2[m] = mfma_bf16(ah[i & 1], bd2[q].h, aW2[m]);
#endif
        }
        PBB_SB();
      }
      if (it == 5) { GSTAMP(a, 3); PBB_W(10); }
      // ---- end of phase A: the DMA copies have landed (vmcnt counts in order: at most the ring's 3 RD loads, the youngest
      // requests, may still be on their way), the workgroup meets
      if (PBB_PRIME < 16) {
        static_assert(PBB_RD == 2 || PBB_RD == 3, "s_waitcnt immediate below");
        if (PBB_RD == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        if (it == 5) { GSTAMP(a, 4); PBB_W(11); }
      } else {
        PBB_LANE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (it == 5) { GSTAMP(a, 4); PBB_W(11); }
        const unsigned o_ = (unsigned)(wave * 16 * 64 + lane) * 16u;
#pragma unroll
        for (int k = 0; k < RD; ++k) { rh[k] = PBB_LDQ(w2d, o_ + 1024u * k); rm[k] = PBB_LDQ(w2d, W2D_T + o_ + 1024u * k); rl[k] = PBB_LDQ(w2d, 2 * W2D_T + o_ + 1024u * k); }
      }
      PBB_BARRIER();
      if (it == 5) { GSTAMP(a, 5); PBB_W(12); }
      // ======== phase B ========
      f32x16 acc = zero16();
      {
        // ---- d h1 = d2 . W2^T (columns [32 w, 32 w + 32)): 16 k-steps; A = D2F (lane = row), B = the ring
        PBB_LANE();
        const unsigned* df = smb + PBB_D2F + (half * 544 + col * 16) / 4;
        const unsigned o_ = (unsigned)(wave * 16 * 64 + lane) * 16u;
        u32x4 dh[2], dm, dl;
        dh[0] = *reinterpret_cast<const u32x4*>(df); dm = *reinterpret_cast<const u32x4*>(df + PBB_D2F_TB / 4); dl = *reinterpret_cast<const u32x4*>(df + 2 * (PBB_D2F_TB / 4));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const unsigned* dn = df + (s + 1) * (1088 / 4);
          if (s == 3 && !(PBB_X & 8)) PBB_STAGE_D3(t + G);
          if (s == 6 && !(PBB_X & 4)) PBB_STAGE_H1();
          if (s == 13) PBB_LOAD_W3();                          // (behind the ring's last requests)
          PBB_SB();
          acc = mfma_bf16(dl, rh[s % RD], acc);
          if (s + 1 < 16) dl = *reinterpret_cast<const u32x4*>(dn + 2 * (PBB_D2F_TB / 4));
          acc = mfma_bf16(dh[s & 1], rl[s % RD], acc);
          if (s + RD < 16) rl[s % RD] = PBB_LDQ(w2d, 2 * W2D_T + o_ + 1024u * (s + RD));
          if (s + 1 < 16) dh[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(dn);
          acc = mfma_bf16(dm, rm[s % RD], acc);
          acc = mfma_bf16(dm, rh[s % RD], acc);
          PBB_SB();
          if (s + 1 < 16) dm = *reinterpret_cast<const u32x4*>(dn + PBB_D2F_TB / 4);
          acc = mfma_bf16(dh[s & 1], rm[s % RD], acc);
          if (s + RD < 16) rm[s % RD] = PBB_LDQ(w2d, W2D_T + o_ + 1024u * (s + RD));
          acc = mfma_bf16(dh[s & 1], rh[s % RD], acc);
          if (s + RD < 16) rh[s % RD] = PBB_LDQ(w2d, o_ + 1024u * (s + RD));
          PBB_SB();
        }
      }
      if (it == 5) { GSTAMP(a, 6); PBB_W(13); }
      // ---- the ReLU mask of d h1, its stores, the workgroup meets
      // (inline asm is opaque to the hazard recogniser: the wait states between the last MFMA's write of the accumulators and a vector
      // read of them -- the compiler puts s_nop 11 there for this MFMA -- are spelled out)
      PBB_SB();
      asm volatile("s_nop 15" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!(PBB_X & 64)) { float o_; asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(o_) : "v"(acc[r]), "s"(hm[r])); acc[r] = o_; }
      }
      PBB_SB();
      if (!(PBB_X & 16)) PBB_STORE_DH1(rowsC);
      PBB_BARRIER();
      if (it == 5) { GSTAMP(a, 7); PBB_W(14); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef PBB_LDQ
#undef PBB_LDS_Q
#undef PBB_LOAD_IDS
#undef PBB_STAGE_IDS
#undef PBB_REQUEST_D3
#undef PBB_STAGE_D3
#undef PBB_REQUEST_H2
#undef PBB_DMA_H1
#undef PBB_STAGE_H1
#undef PBB_STORE_DH1
#undef PBB_LOAD_W3
  // ---- epilogue: the partial weight gradients of this workgroup
  __syncthreads();
  GSTAMP(a, 15);
  const int lane = lane0, tid = threadIdx.x, col = lane & 31;
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
#pragma unroll
  for (int m = 0; m < 8; ++m) store_acc(ar + a.o_w2 + (size_t)(32 * m) * D_H + 32 * wave, D_H, aW2[m], lane);
  store_acc(ar + a.o_w3 + (size_t)(32 * wave) * D_E, D_E, aW3, lane);
  { unsigned lo_, hi_; half_bcast(__float_as_uint(gb2), lo_, hi_); gb2 = __uint_as_float(lo_) + __uint_as_float(hi_); }
  if (lane < 32) ar[a.o_b2 + 32 * wave + col] = gb2;
  float* red = reinterpret_cast<float*>(smb + PBB_RED);
  red[(tid >> 4) * 32 + 2 * (tid & 15)] = g3x; red[(tid >> 4) * 32 + 2 * (tid & 15) + 1] = g3y;
  __syncthreads();
  if (tid < D_E) {
    float p_ = red[tid];
#pragma unroll 8
    for (int r = 1; r < 32; ++r) p_ += red[r * 32 + tid];
    ar[a.o_b3 + tid] = p_;
  }
#undef PBB_LANE
#undef PBB_BARRIER
}

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
  float mult;                       // cfg.gnet.pw_feat_multiplyer: the score columns of the features carry it too
};

__global__ void __launch_bounds__(256) pw_w1_nodesums(const PwW1Args a) {
  // rows in flight per wave.  The grid is bounded by the arena's partial slots (512 workgroups = two waves per SIMD), so the
  // bytes in flight are waves x NQ KB: 8 rows = 16 MB on the device = 4.7 TB/s at ~3 us per round trip (0.465 ms); 16: 0.417;
  // 24: 0.408; 32: 0.420 (173 registers at 24: still two waves per SIMD).  Same order of additions, same bits.
  constexpr int NQ = 24;
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
      // NQ rows in flight per wave (the sums are latency-bound: 2 rows per round trip kept the kernel at 2.5 TB/s,
      // 4 at 3.9, 8 at 4.7); rows are taken in ascending edge order, missing slots re-read the last row and are skipped.
      // The geometry columns of the 64 edges of this pass sit in two registers per lane (one coalesced load) and
      // reach the sums through v_readlane.
      float4 geA = make_float4(0.f, 0.f, 0.f, 0.f), geB = geA;
      if (in) {
        geA = *reinterpret_cast<const float4*>(a.geo + (size_t)el * 8);
        geB = *reinterpret_cast<const float4*>(a.geo + (size_t)el * 8 + 4);
      }
      while (mo) {
        int j[NQ]; bool hv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          hv[q] = mo != 0ull;
          j[q] = hv[q] ? __builtin_ctzll(mo) : (q ? j[q - 1] : 0);
          if (hv[q]) mo &= mo - 1;
        }
        float4 d[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[q] = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)(base + j[q]) * D_H + 4 * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (hv[q]) {                                      // wave-uniform
            const float4 dq = d[q];
            const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geA.x), j[q])), g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geA.y), j[q]));
            const float g2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geA.z), j[q])), g3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geA.w), j[q]));
            const float g4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geB.x), j[q])), g5 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geB.y), j[q]));
            const float g6 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geB.z), j[q]));
            S.x += dq.x; S.y += dq.y; S.z += dq.z; S.w += dq.w;
            ACC4(g[0], g0, dq); ACC4(g[1], g1, dq); ACC4(g[2], g2, dq); ACC4(g[3], g3, dq);
            ACC4(g[4], g4, dq); ACC4(g[5], g5, dq); ACC4(g[6], g6, dq);
          }
        }
      }
      while (mr) {
        int r[NQ]; bool hv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          hv[q] = mr != 0ull;
          const int jj = hv[q] ? __builtin_ctzll(mr) : 0;
          if (hv[q]) mr &= mr - 1;
          r[q] = hv[q] ? __builtin_amdgcn_readlane(tt, jj) : (q ? r[q - 1] : 0);
        }
        float4 tq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) tq[q] = *reinterpret_cast<const float4*>(a.d_h1 + (size_t)r[q] * D_H + 4 * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (hv[q]) { T.x += tq[q].x; T.y += tq[q].y; T.z += tq[q].z; T.w += tq[q].w; }
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
    const float sc = valid ? a.scores[i] * a.mult : 0.f;     // (x * 1.0f is exact)
    const bool member = valid && (!a.multiclass || a.classes[i] - 1 == k);
    unsigned long long mask = __ballot(member);
    while (mask) {
      // four member rows in flight (ascending index order; a missing slot re-reads the last row and is skipped)
      int j[4]; float s[4]; bool hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        hv[u] = mask != 0ull;                               // wave-uniform
        j[u] = hv[u] ? __builtin_ctzll(mask) : (u ? j[u - 1] : 0);
        if (hv[u]) mask &= mask - 1;
        s[u] = __shfl(sc, j[u]);
      }
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = src + (size_t)(base + j[u]) * D_H + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = row[64 * q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (hv[u]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaf(s[u], v[u][q], acc[q]);
        }
      }
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
// num_pwfeat_fc = 0: the 2C' score rows of a block's pw_fc1 (network.py:413-419 one-hot x score columns of the raw pairwise
// features, read by pw_fc1 directly).  The gradient of row k (centre column of class k) is sum_{i : class_i = k} s_i *
// (sum over i's own pairs of g1) = s_i d_rc[i]; of row C' + k: s_i * (sum over the pairs whose NEIGHBOUR is i of g1) =
// s_i (d_rn[i] + g1[self pair of i]) -- gather_winners leaves the self pair out of d_rn (the neighbour FEATURES of a self pair
// are zeroed, its score column is not).  Launched per block behind gather_winners; grid (2 C', chunks), one wave per workgroup,
// lane = output column; members of the class are picked by a ballot over 64 detections at a time and added in index order.
struct RawW1Args {
  int n_det, cprime, multiclass, nchunks;
  const float* d_rc; const float* d_rn; const float* g1c; const int* spos;
  const float* scores; const int* classes; float mult;
  float* arena; long long stride, o_w1;
};

__global__ void __launch_bounds__(64) raw_w1_classrows(const RawW1Args a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const bool nb = r >= a.cprime;
  const int k = nb ? r - a.cprime : r;
  const int per = (a.n_det + a.nchunks - 1) / a.nchunks;
  const int c0 = blockIdx.y * per, c1 = min(a.n_det, c0 + per);
  float acc = 0.f;
  for (int base = c0; base < c1; base += 64) {
    const int i = base + lane;
    const bool valid = i < c1;
    const float sc = valid ? a.scores[i] * a.mult : 0.f;
    const bool member = valid && (a.multiclass ? a.classes[i] - 1 == k : true);
    const int sp = (valid && nb) ? a.spos[i] : -1;
    unsigned long long mask = __ballot(member);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float s_ = __shfl(sc, j);
      const int spj = __shfl(sp, j);
      float v = (nb ? a.d_rn : a.d_rc)[(size_t)(base + j) * D_P + lane];
      if (spj >= 0) v += a.g1c[(size_t)spj * D_P + lane];
      acc = fmaf(s_, v, acc);
    }
  }
  a.arena[(size_t)blockIdx.y * a.stride + a.o_w1 + (size_t)r * D_P + lane] = acc;
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
  int kp;                 // pairwise rows of a block's pw_fc1 (32; 2C'+7 with num_pwfeat_fc = 0)
  int cls_rows, n_cls;    // of those, leading score rows written by raw_w1_classrows (n_cls partial copies); 0 with a pw-MLP
  float* grads;
};

// 256 parameters x 4 partial groups per workgroup: a lane owns four consecutive parameters (one 16-byte load per
// copy; the class boundaries and the arena stride are multiples of 4, so the four share their copy count), group g
// adds the copies g, g + 4, ... (four independent load streams per parameter instead of one dependent chain), the
// groups are folded in a fixed order.  The tail of the parameter vector (its length is odd) is handled per element.
__global__ void __launch_bounds__(256) reduce_partials(const ReduceArgs a) {
  __shared__ float4 part[4][64];
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long p = ((long long)blockIdx.x * 64 + px) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p < a.total) {
    int n;
    if (p < a.w1c_end) n = a.n_w1c;
    else if (p < a.pw1_end) n = a.n_w1;
    else if (p < a.pw_end) n = a.n_pw;
    else if (p < a.pw_end + a.blk_sz * a.nblocks) {
      const long long q = (p - a.pw_end) % a.blk_sz;
      const long long w1 = D_S * D_R + D_R;                       // start of pw_fc1 weights
      const long long w2 = w1 + (long long)(a.kp + 2 * D_R) * D_P + D_P;      // start of pw_fc2 weights
      const bool edge = (q >= w1 + (long long)a.cls_rows * D_P && q < w1 + (long long)a.kp * D_P) || (q >= w2 && q < w2 + D_P * D_P + D_P);
      n = (q >= w1 && q < w1 + (long long)a.cls_rows * D_P) ? a.n_cls : edge ? a.n_edge : a.n_node;
    } else n = a.n_head;
    // the arena rows are padded to the stride: the 16-byte load of the last, partial group stays inside its row
    const float* src = a.arena + p;
    int k = g;
    for (; k + 12 < n; k += 16) {
      const float4 x0 = *reinterpret_cast<const float4*>(src + (size_t)k * a.stride);
      const float4 x1 = *reinterpret_cast<const float4*>(src + (size_t)(k + 4) * a.stride);
      const float4 x2 = *reinterpret_cast<const float4*>(src + (size_t)(k + 8) * a.stride);
      const float4 x3 = *reinterpret_cast<const float4*>(src + (size_t)(k + 12) * a.stride);
      v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
      v.x += x1.x; v.y += x1.y; v.z += x1.z; v.w += x1.w;
      v.x += x2.x; v.y += x2.y; v.z += x2.z; v.w += x2.w;
      v.x += x3.x; v.y += x3.y; v.z += x3.z; v.w += x3.w;
    }
    for (; k < n; k += 4) {
      const float4 x0 = *reinterpret_cast<const float4*>(src + (size_t)k * a.stride);
      v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
    }
  }
  part[g][px] = v;
  __syncthreads();
  if (g == 0 && p < a.total) {
    const float4 p0 = part[0][px], p1 = part[1][px], p2 = part[2][px], p3 = part[3][px];
    const float4 r = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                 (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    if (p + 3 < a.total) *reinterpret_cast<float4*>(a.grads + p) = r;
    else { a.grads[p] = r.x; if (p + 1 < a.total) a.grads[p + 1] = r.y; if (p + 2 < a.total) a.grads[p + 2] = r.z; }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" int gnet_backward_prepare(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                                     const float* params, gnet_buffers* buf, int32_t phase, gnet_stream_t stream) {
  clear_hip_error();
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !in || !params || !buf) return GNET_ERR_INVALID;
  if (!buf->d_pw || !buf->blk_parg[1] || !buf->ewin || !buf->wlist || !buf->pw_rows || !buf->tpos || !buf->wrow) return GNET_ERR_INVALID;   // plan(training >= 1)
  if (cfg->num_pwfeat_fc == 0 && !buf->spos) return GNET_ERR_INVALID;
  const int E = (int)shape->n_edge;
  if (shape->n_det == 0 || E == 0) return GNET_OK;
  if (E > (1 << 24) - 128) return GNET_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (phase == 0 || phase == 1) {
    // zeroing: the edge stages accumulate into d_pw (winner rows only); the winner maps start empty
    HIP_CHECK_RET(hipMemsetAsync(buf->d_pw, 0, (size_t)E * D_E * sizeof(float), s));
    const int st = edge_stage_clear(cfg, shape, buf, s);
    if (st != GNET_OK) return st;
  }
  if (phase == 1) return GNET_OK;
  // winner maps / lists of ALL blocks (they depend on the forward pass only: off the backward chain)
  if (phase < 0 || phase > 3) return GNET_ERR_INVALID;
  return edge_stage_prepare(cfg, shape, make_layout(cfg), params, buf, s, phase == 2 ? 1 : phase == 3 ? 2 : 0);
}

extern "C" int gnet_backward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                             const float* params, gnet_buffers* buf, float* grads, int32_t prepared,
                             void* prepared_event, void* positions_event, gnet_stream_t stream) {
  clear_hip_error();
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !in || !params || !buf || !grads) return GNET_ERR_INVALID;
  if (!buf->arena || !buf->d_x || !buf->d_logits || !buf->blk_parg[1] || !buf->ewin || !buf->wlist || !buf->pw_rows || !buf->tpos || !buf->wrow) return GNET_ERR_INVALID;   // plan(training >= 1)
  if (cfg->num_pwfeat_fc == 0 ? !buf->spos : (!buf->pw_h1 || !buf->pw_h2 || !buf->d_h1 || !buf->w1_s || !buf->w1_t)) return GNET_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const ParamLayout L = make_layout(cfg);
  const int B = cfg->num_blocks;
  const int N = shape->n_det;
  const int E = (int)shape->n_edge;
  if (N == 0) {
    HIP_CHECK_RET(hipMemsetAsync(grads, 0, (size_t)L.total * sizeof(float), s));
    return GNET_OK;
  }
  if ((size_t)GNET_ARENA_PARTIALS * (size_t)arena_stride(L.total) > buf->arena_floats) return GNET_ERR_WORKSPACE;
  if (E > (1 << 24) - 128) return GNET_ERR_UNSUPPORTED;   // 32-bit byte offsets into the [E,64] fp32 arrays
  void* prof = buf->profiler;
  const long long stride = arena_stride(L.total);
  // one tile per workgroup (blk_bwd_node<.., true>) while the tiles fit the CUs.  Measured at 500 tiles (8 x 2000 detections):
  // two co-resident workgroups per CU finish in 23.2 us instead of 25.8 (their chains overlap except for the MFMA pipe, which is 40 %
  // of a chain) -- and reduce_partials then reads 500 instead of 256 partial copies of the node parameters: 0.05 ms gained, 0.05 ms
  // lost.  At up to 256 tiles (the reference's one-image step: 63) the partial count is the same and the shorter front is a gain
  // (19 -> 17 us per launch).
  const bool node_one = (N + 31) / 32 <= 256;
  const int g_node = node_one ? (N + 31) / 32 : min((N + 31) / 32, 256);                // node-kernel workgroups (32-detection tiles)
  const int g_head = min((N + 31) / 32, 256);
  const int etiles = (E + 31) / 32;
  const int g_edge = E > 0 ? 512 : 0;                                                  // edge_bwd_w workgroups (two per CU)
  const int g_pw = E > 0 ? min(etiles, 256) : 0;                                       // pw_bwd_main: one workgroup per CU, looping over its tiles
  const int g_w1 = E > 0 ? max(1, min(GNET_ARENA_PARTIALS, (N + 3) / 4)) : 0;          // node-sum workgroups
  // node chunks per class row: ~2048 workgroups in all (at 256 -- one chunk for 80 classes, 160 workgroups scanning all detections --
  // the kernel took 68 us; 1024: 23; 2048: 17; 4096: 18)
  const int g_w1c = E > 0 ? max(1, min(128, 2048 / (2 * L.cprime))) : 0;
  const int g_rawc = (E > 0 && L.raw) ? max(1, min(64, min((N + 255) / 256, 1024 / (2 * L.cprime)))) : 0;   // chunks per class row of raw_w1_classrows
  const EdgeGeom G = edge_geom(E, N);

  // dynamic-LDS limits are per device and cheap to set: no process-global "done" flag
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_bwd_main<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwBwdSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_bwd_main<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwBwdSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_bwd_bf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwBwdBfSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_bwd_bf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwBwdBfSmem));
  { const int st = edge_stage_set_attributes(); if (st != GNET_OK) return st; }

  if (E > 0 && !prepared) {
    const int st = gnet_backward_prepare(cfg, shape, in, params, buf, 0, stream);
    if (st != GNET_OK) return st;
  }
  {
    HeadBwdArgs h;
    h.n_det = N; h.d_logits = buf->d_logits; h.head2 = buf->head2; h.head1 = buf->head1; h.xb = buf->block_feats[B];
    h.hw1 = params + L.hw1; h.hw2 = params + L.hw2; h.hwl = params + L.hwl;
    h.d_x = buf->d_x; h.arena = buf->arena; h.stride = stride;
    h.o_hw1 = L.hw1; h.o_hb1 = L.hb1; h.o_hw2 = L.hw2; h.o_hb2 = L.hb2; h.o_hwl = L.hwl; h.o_hbl = L.hbl;
    GNET_LAUNCH(prof, GNET_K_HEAD_BWD, s, head_bwd<<<g_head, 256, 0, s>>>(h));
  }
  // backward chain: node(post B) -> edge(B) -> node(pre B + post B-1) -> edge(B-1) -> ... -> node(pre 1)
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)blk_bwd_node<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlkNodeSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)blk_bwd_node<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlkNodeSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)blk_bwd_node<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlkNodeSmem));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)blk_bwd_node<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlkNodeSmem));
  for (int b = B + 1; b >= 1; --b) {
    // node stage: pre of block b (b <= B), post of block b-1 (b >= 2)
    BlkNodeArgs n;
    n.n_det = N; n.do_pre = b <= B; n.do_post = b >= 2; n.want_dx0 = (b == 1 && buf->start_feat) ? 1 : 0;
    n.d_rc = E > 0 ? buf->d_rc : nullptr; n.d_rn = E > 0 ? buf->d_rn : nullptr;
    if (b == B && E > 0 && prepared && positions_event) {
      // first use of the reversed pairs' list positions (phase 3 of the preparation: it ran beside the first edge_bwd_w)
      HIP_CHECK_RET(hipStreamWaitEvent(s, (hipEvent_t)positions_event, 0));
    }
    if (b <= B && E > 0) {
      GNET_LAUNCH(prof, GNET_K_GATHER, s, gather_winners<<<(N + 3) / 4, 256, 0, s>>>(
          buf->d_g1, buf->row_ptr, buf->wrow + (size_t)(b - 1) * G.tf_stride, buf->tpos + (size_t)(b - 1) * G.wl_stride, N,
          buf->d_rc, buf->d_rn
#ifdef GNET_TRACE
          , b == B / 2 ? gnet_trace_ptr("GNET_TRACE_GATHER") : nullptr
#endif
          ));
    }
    if (b <= B && E > 0 && L.raw) {
      RawW1Args rw;
      rw.n_det = N; rw.cprime = L.cprime; rw.multiclass = cfg->num_classes > 1; rw.nchunks = g_rawc;
      rw.d_rc = buf->d_rc; rw.d_rn = buf->d_rn; rw.g1c = buf->d_g1; rw.spos = buf->spos + (size_t)(b - 1) * G.tf_stride;
      rw.scores = in->det_scores; rw.classes = in->det_classes; rw.mult = cfg->pw_feat_multiplyer;
      rw.arena = buf->arena; rw.stride = stride; rw.o_w1 = L.blk[b].w1;
      GNET_LAUNCH(prof, GNET_K_W1_CLASS, s, raw_w1_classrows<<<dim3(2 * L.cprime, g_rawc), 64, 0, s>>>(rw));
    }
    n.x_prev = b >= 2 ? buf->block_feats[b - 1] : buf->start_feat;
    if (b <= B) {
      const BlockLayout& K = L.blk[b];
      // (the kernel addresses the centre / neighbour rows of pw_fc1 as rows 32-95: with kp pairwise rows they start at row kp)
      const int64_t w1_cn = K.w1 + (int64_t)(L.kp - D_E) * D_P;
      n.r = buf->blk_r[b]; n.w1 = params + w1_cn; n.wr = params + K.wr;
      n.o_w1 = w1_cn; n.o_b1 = K.b1; n.o_wr = K.wr; n.o_br = K.br;
      n.r_nb = cfg->neighbor_feats ? buf->blk_rnb[b] : nullptr;
      n.wrn = cfg->neighbor_feats ? params + K.wrn : nullptr;
      n.o_wrn = K.wrn; n.o_brn = K.brn;
    } else { n.r = nullptr; n.w1 = n.wr = nullptr; n.o_w1 = n.o_b1 = n.o_wr = n.o_br = 0; n.r_nb = nullptr; n.wrn = nullptr; n.o_wrn = n.o_brn = 0; }
    if (b >= 2) {
      const BlockLayout& K = L.blk[b - 1];
      n.q = buf->blk_q[b - 1]; n.pm = (const unsigned long long*)buf->blk_pm[b - 1];
      n.w4 = params + K.w4; n.w3 = params + K.w3;
      n.o_w4 = K.w4; n.o_b4 = K.b4; n.o_w3 = K.w3; n.o_b3 = K.b3;
    } else { n.q = nullptr; n.pm = nullptr; n.w4 = n.w3 = nullptr; n.o_w4 = n.o_b4 = n.o_w3 = n.o_b3 = 0; }
    n.d_x = buf->d_x; n.d_pc = buf->d_pc; n.arena = buf->arena; n.stride = stride;
    GNET_TRACE_SET(n, "NODE_BWD", b == B / 2);
    if (node_one) {
      if (cfg->neighbor_feats) { GNET_LAUNCH(prof, GNET_K_NODE_BWD, s, blk_bwd_node<true, true><<<g_node, 256, kBlkNodeSmem, s>>>(n)); }
      else { GNET_LAUNCH(prof, GNET_K_NODE_BWD, s, blk_bwd_node<false, true><<<g_node, 256, kBlkNodeSmem, s>>>(n)); }
    } else {
      if (cfg->neighbor_feats) { GNET_LAUNCH(prof, GNET_K_NODE_BWD, s, blk_bwd_node<true, false><<<g_node, 256, kBlkNodeSmem, s>>>(n)); }
      else { GNET_LAUNCH(prof, GNET_K_NODE_BWD, s, blk_bwd_node<false, false><<<g_node, 256, kBlkNodeSmem, s>>>(n)); }
    }
    // edge stage of block b-1
    if (b >= 2 && E > 0) {
      if (b == B + 1 && prepared && prepared_event) {
        // first use of the prepared winner lists: head_bwd and the node kernel above needed none of them
        HIP_CHECK_RET(hipStreamWaitEvent(s, (hipEvent_t)prepared_event, 0));
      }
      const int st = edge_stage_block(cfg, shape, L, params, b - 1, buf, g_edge, s);
      if (st != GNET_OK) return st;
    }
  }
  if (E > 0 && !L.raw) {
    // rows of the pw-MLP backward = edges that won in at least one block (list B of edge_stage_prepare)
    const unsigned long long* eany = (const unsigned long long*)buf->ewin + (size_t)B * G.bm_stride;
    const int* wg_off = buf->rl_scratch + (size_t)(B + 1) * G.n_wg;
    PwBwdArgs p;
    p.rows = buf->pw_rows; p.n_rows = wg_off + (size_t)B * (G.n_wg + 1) + G.n_wg;
    p.n_edge = E; p.pw = buf->pw_feats; p.d_pw = buf->d_pw; p.h1 = buf->pw_h1; p.h2 = buf->pw_h2;
    p.w2 = params + L.pw2; p.w3 = params + L.pw3; p.d_h1 = buf->d_h1;
    p.arena = buf->arena; p.stride = stride; p.o_w2 = L.pw2; p.o_b2 = L.pb2; p.o_w3 = L.pw3; p.o_b3 = L.pb3;
    GNET_TRACE_SET(p, "PW_BWD", true);
    static const bool fp32_pipe = getenv("GNET_PW_FP32_PIPE") != nullptr;     // measurement only: round 5's pw_bwd_main (fp32 MFMA)
    bool big = (long long)E + 64 > (1ll << 22);
#ifdef PW_BWD_FORCE_BIG      /* verification builds: the 64-bit addressing variant on shapes the tests can afford */
    big = true;
#endif
    if (fp32_pipe) {
      if (big) { GNET_LAUNCH(prof, GNET_K_PW_BWD, s, pw_bwd_main<true><<<g_pw, 512, kPwBwdSmem, s>>>(p)); }
      else { GNET_LAUNCH(prof, GNET_K_PW_BWD, s, pw_bwd_main<false><<<g_pw, 512, kPwBwdSmem, s>>>(p)); }
    } else {
      PwBwdBfArgs pb; pb.p = p; pb.wbf = reinterpret_cast<const unsigned*>(buf->packed_t + packed_pwbf_off(L));
      if (big) { GNET_LAUNCH(prof, GNET_K_PW_BWD, s, pw_bwd_bf<true><<<g_pw, 512, kPwBwdBfSmem, s>>>(pb)); }
      else { GNET_LAUNCH(prof, GNET_K_PW_BWD, s, pw_bwd_bf<false><<<g_pw, 512, kPwBwdBfSmem, s>>>(pb)); }
    }
    PwW1Args w;
    w.n_det = N; w.cprime = L.cprime; w.multiclass = cfg->num_classes > 1;
    w.row_ptr = buf->row_ptr; w.edge_t = buf->edge_t; w.geo = buf->geo; w.d_h1 = buf->d_h1;
    w.scores = in->det_scores; w.classes = in->det_classes; w.eany = eany;
    w.w1_s = buf->w1_s; w.w1_t = buf->w1_t;
    w.arena = buf->arena; w.stride = stride; w.o_w1 = L.pw1; w.o_b1 = L.pb1; w.nchunks = g_w1c; w.mult = cfg->pw_feat_multiplyer;
    GNET_LAUNCH(prof, GNET_K_W1_SUMS, s, pw_w1_nodesums<<<g_w1, 256, 0, s>>>(w));
    GNET_LAUNCH(prof, GNET_K_W1_CLASS, s, pw_w1_classrows<<<dim3(2 * L.cprime, g_w1c), 256, 0, s>>>(w));
  }
  {
    ReduceArgs r;
    r.arena = buf->arena; r.stride = stride; r.total = L.total;
    r.w1c_end = (long long)2 * L.cprime * D_H; r.pw1_end = L.pw2; r.pw_end = L.blk[1].wr;
    r.blk_sz = (B > 1) ? (L.blk[2].wr - L.blk[1].wr) : (L.hw1 - L.blk[1].wr);
    r.nblocks = B;
    r.n_w1c = g_w1c; r.n_w1 = g_w1; r.n_pw = g_pw; r.n_edge = g_edge; r.n_node = g_node; r.n_head = g_head;
    r.kp = L.kp; r.cls_rows = L.raw ? 2 * L.cprime : 0; r.n_cls = g_rawc;
    if (L.raw) { r.w1c_end = 0; r.pw1_end = 0; }      // no pw-MLP parameters: the blocks start at offset 0
    r.grads = grads;
    GNET_LAUNCH(prof, GNET_K_REDUCE, s, reduce_partials<<<(int)((L.total + 255) / 256), 256, 0, s>>>(r));
  }
  return launch_status();
}
