// Forward pass of Gnet (network.py:197-273): pairwise geometry features, the pairwise-feature
// MLP, the stacked _block message passing, and the prediction head.
//
// Kernels
//   pack_transpose   W[in,out] -> Wt[out,in] for every FC (B operands are read 4-along-K)
//   edge_geometry    _geometry_feats (network.py:411-454): the 7 geometry columns per edge; the 2C' one-hot x score columns
//                    as two per-detection tables (the [E,167] feature matrix never exists)
//   pw_fwd2          _pw_feats_fc (:324-342): fc1 = two table rows + a K = 8 product, fc2 with its weights resident in
//                    registers, fc3 chained on fc2's accumulators (skipped with num_pwfeat_fc = 0)
//   edge_fwd         per block: build_context + pw_fc1 + pw_fc2 + segment_max (:367-388)
//   node_fwd         per block: fc1, fc2, shortcut (:390-408) of block b and reduce_dim (:348-354)
//                    + the per-node halves of pw_fc1 of block b+1; after the last block: head (:258-273)
// All dense layers run on v_mfma_f32_32x32x2_f32 (exact fp32).
#include "common.hpp"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------
// pack_transpose: matrix id -> (offset, in, out) computed from the flat layout.  raw (num_pwfeat_fc = 0): no pw-MLP matrices,
// and a block's pw_fc1 is [dpw + 64, 64] -- *kind = 1 marks it: its packed copy keeps the [64][96] shape every consumer reads
// (columns 0-6 = the seven geometry rows, 7-31 = zeros, 32-63 centre, 64-95 neighbour: the edge kernels then multiply the
// geometry columns, padded to 32 in pw_feats, by it -- exact zeros for the padding -- and the 2C' score rows enter through the
// per-detection tables node_fwd adds to rc / rn).
__device__ __forceinline__ void mat_info(int id, int dpw, int nblocks, int nf, int raw, long long* off, int* in, int* out, int* kind, long long* poff) {
  const int kp = raw ? dpw : D_E;
  const long long pw_sz = raw ? 0 : (long long)dpw * D_H + D_H + D_H * D_H + D_H + D_H * D_E + D_E;
  const long long blk_core = D_S * D_R + D_R + (long long)(kp + 2 * D_R) * D_P + D_P + D_P * D_P + D_P + D_P * D_P + D_P +
                             D_P * D_S + D_S;
  const long long blk_sz = blk_core + (nf ? D_S * D_R + D_R : 0);      // reduce_dim_neighbor follows fc2
  *kind = 0;
  if (!raw) {
    if (id == 0) { *off = 0; *in = dpw; *out = D_H; return; }
    if (id == 1) { *off = (long long)dpw * D_H + D_H; *in = D_H; *out = D_H; *kind = 2; return; }
    if (id == 2) { *off = (long long)dpw * D_H + D_H + D_H * D_H + D_H; *in = D_H; *out = D_E; return; }
    id -= 3;
  }
  const int mpb = 5 + (nf ? 1 : 0);
  if (id < mpb * nblocks) {
    const int b = id / mpb, m = id % mpb;
    long long o = pw_sz + (long long)b * blk_sz;
    if (m == 5) { *off = o + blk_core; *in = D_S; *out = D_R; return; }
    if (m == 0) { *off = o; *in = D_S; *out = D_R; return; }
    o += D_S * D_R + D_R;
    if (m == 1) {
      *off = o; *in = kp + 2 * D_R; *out = D_P; *kind = raw ? 1 : 0;
      // (raw: the [64][96] copy lives behind the parameters' copies -- common.hpp packed_w1_off)
      if (raw) *poff = pw_sz + (long long)nblocks * blk_sz + (D_S * D_HEAD + D_HEAD + D_HEAD * D_HEAD + D_HEAD + D_HEAD + 1) + (long long)b * (D_P * (D_E + 2 * D_R));
      return;
    }
    o += (long long)(kp + 2 * D_R) * D_P + D_P;
    if (m == 2) { *off = o; *in = D_P; *out = D_P; return; }
    o += D_P * D_P + D_P;
    if (m == 3) { *off = o; *in = D_P; *out = D_P; return; }
    o += D_P * D_P + D_P;
    *off = o; *in = D_P; *out = D_S; return;
  }
  id -= mpb * nblocks;
  long long o = pw_sz + (long long)nblocks * blk_sz;
  if (id == 0) { *off = o; *in = D_S; *out = D_HEAD; return; }
  o += D_S * D_HEAD + D_HEAD;
  *off = o; *in = D_HEAD; *out = D_HEAD;
}

constexpr int PACK_X = 32;   // workgroups per matrix: the 256 x 256 one is 32 strided 4-byte gathers per thread at 8 (23 us), 8 at 32 (14 us)
__global__ void __launch_bounds__(256) pack_transpose(const float* __restrict__ params, float* __restrict__ packed,
                                                      int dpw, int nblocks, int nf, int raw) {
  long long off, poff = -1; int in, out, kind;
  mat_info(blockIdx.y, dpw, nblocks, nf, raw, &off, &in, &out, &kind, &poff);
  const int total = in * out;
  if (kind == 2) {
    // pw_feats/fc2 (256 x 256) is streamed from L2 by pw_fwd as MFMA operand fragments: FRAGMENT-MAJOR, so that the 64
    // lanes of one load instruction read one contiguous 1 KB block (8 cache lines) instead of 32 bytes of each of 32 rows
    // (32 lines per instruction: the stream's tag look-ups, not its bytes, loaded the CU's texture path).  Wave w owns output
    // columns [32 w, 32 w + 32); k-step s covers k = 8 s .. 8 s + 7; lane (r, h) holds W[k = 8 s + 4 h + t][o = 32 w + r], t = 0..3:
    //   packed[(((w * 32 + s) * 64) + 32 h + r) * 4 + t]
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
      const int t = i & 3, lane = (i >> 2) & 63, s_ = (i >> 8) & 31, w = i >> 13;
      const int o = 32 * w + (lane & 31), k = 8 * s_ + 4 * (lane >> 5) + t;
      packed[off + i] = params[off + (long long)k * out + o];
    }
    return;
  }
  if (kind == 1) {
    // raw pw_fc1 [dpw + 64, 64] -> [64][96]: geometry rows | zeros | centre | neighbour
    const int g0 = dpw - 7;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < D_P * (D_E + 2 * D_R); i += gridDim.x * 256) {
      const int o = i / (D_E + 2 * D_R), k = i - o * (D_E + 2 * D_R);
      float v = 0.f;
      if (k < 7) v = params[off + (long long)(g0 + k) * D_P + o];
      else if (k >= D_E) v = params[off + (long long)(dpw + k - D_E) * D_P + o];
      packed[poff + i] = v;
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int o = i / in, k = i - o * in;   // packed[o][k] = W[k][o]
    packed[off + i] = params[off + (long long)k * out + o];
  }
}

// ------------------------------------------------------------------------------------------
// pack_pw_bf16: the pw-MLP's fc2 / fc3 weights as bf16 three-term operand fragments (common.hpp PWBF_*): every fp32 weight is split
// once per step -- hi, mid by truncation, lo the remainder: hi + mid + lo == w exactly (split3_pk) -- and stored where the lane that
// multiplies by it finds its eight k-slots of a k-step as one 16-byte load.  One thread per (array, wave, k-step, lane, slot pair).
__global__ void __launch_bounds__(256) pack_pw_bf16(const float* __restrict__ w2, const float* __restrict__ w3, unsigned* __restrict__ out) {
  constexpr int N2 = 8 * 16 * 64 * 4, N3 = 8 * 2 * 64 * 4;
  int i = blockIdx.x * 256 + threadIdx.x;
  float x0, x1;
  unsigned* dst; int n;
  if (i < 2 * N2) {
    const bool dgrad = i >= N2;                  // W2D (backward: d h1 = d2 . W2^T) behind W2A (forward: h2^T = W2^T . h1^T)
    if (dgrad) i -= N2;
    const int p = i & 3, lane = (i >> 2) & 63, s_ = (i >> 8) & 15, w = i >> 12;
    const int col = lane & 31, half = lane >> 5;
    const int k0 = 32 * (s_ >> 1) + frag_feat(s_ & 1, half, 2 * p);          // (slot 2 p + 1 is the next feature)
    if (!dgrad) { x0 = w2[(size_t)k0 * D_H + 32 * w + col]; x1 = w2[(size_t)(k0 + 1) * D_H + 32 * w + col]; }
    else { x0 = w2[(size_t)(32 * w + col) * D_H + k0]; x1 = w2[(size_t)(32 * w + col) * D_H + k0 + 1]; }
    dst = out + (dgrad ? PWBF_W2D : PWBF_W2A); n = N2;
  } else if (i < 2 * N2 + 2 * N3) {
    i -= 2 * N2;
    const bool dgrad = i >= N3;                  // W3D (backward: d2 = d3 . W3^T, either orientation) behind W3B (forward: fc3)
    if (dgrad) i -= N3;
    const int p = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 1, w = i >> 9;
    const int col = lane & 31, half = lane >> 5;
    if (!dgrad) {
      const int f0 = 32 * w + frag_feat(q, half, 2 * p);
      x0 = w3[(size_t)f0 * D_E + col]; x1 = w3[(size_t)(f0 + 1) * D_E + col];
    } else {
      const int o0 = 16 * q + 8 * half + 2 * p;
      x0 = w3[(size_t)(32 * w + col) * D_E + o0]; x1 = w3[(size_t)(32 * w + col) * D_E + o0 + 1];
    }
    dst = out + (dgrad ? PWBF_W3D : PWBF_W3B); n = N3;
  } else return;
  unsigned ph, pm, pl;
  split3_pk(x0, x1, ph, pm, pl);
  dst[i] = ph; dst[n + i] = pm; dst[2 * n + i] = pl;
}

// ------------------------------------------------------------------------------------------
struct GeoArgs {
  int n_edge;
  const int* edge_c; const int* edge_n; const float* edge_iou;
  const float4* dets; const float* scores; const int* classes;
  int cprime, multiclass;
  float* geo;       // [E,8]  (iou, x_dist, y_dist, l2_dist, w_diff, h_diff, aspect_diff, 0)
  // The 2C one-hot x score columns of _geometry_feats (network.py:413-419) enter pw_feats/fc1 as ONE row of its weight matrix
  // times the detection's score -- a per-DETECTION term, not a per-edge one: tc[i] = score_i W1[class_i - 1] + b1 (i as the
  // centre of a pair), tn[i] = score_i W1[C + class_i - 1] (i as the neighbour).  Two [N,256] tables per step; fc1 of an
  // edge (c, n) starts from tc[c] + tn[n] (pw_fwd).
  const float* w1; const float* b1;   // pw_feats/fc1 natural [dpw,256], [256]
  float* tc; float* tn;               // [N,256] each
  int* edge_nz;     // [E+64] n, or n_det for self pairs (their neighbour features are zeroed, network.py:371-374) and the tail
  int n_det;
  float mult;       // cfg.gnet.pw_feat_multiplyer (network.py:199-200: the whole feature row times it)
  // per detection, once per step (the same for all blocks): 1 = its segment-max records must start from zero -- it has
  // no edge, or its edges are split between two waves' ranges of edge_fwd_w (combined atomically there); 0 = one wave
  // writes them with a plain store.  node_fwd reads the flag instead of dividing row pointers by the range size 16 times
  // per thread and launch.
  const int* row_ptr; int* straddle; int ef_tiles, ef_waves;
  // num_pwfeat_fc = 0 (network.py:217-221: pw_feats = the raw feature columns): no tables here (the score columns enter every
  // block's pw_fc1 through per-block tables, node_fwd); the 7 geometry columns go to pw [E,32] padded with zeros -- the
  // edge kernels' pairwise operand -- and a self pair's neighbour row is row n_det + 1 + c of rn (its score term alone)
  int raw; float* pw;
};

// _geometry_feats (network.py:411-454), one thread per edge.  The 2C one-hot x score columns become the per-detection
// tables tc / tn (16 bytes of each per thread, grid-stride); the 7 geometry columns are evaluated in the reference's fp32
// operation order.
__global__ void __launch_bounds__(256) edge_geometry(const GeoArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  for (int i = e; i < a.n_det; i += gridDim.x * 256) {
    const int eb = a.row_ptr[i], ee = a.row_ptr[i + 1];
    a.straddle[i] = (ee == eb || efw_owner(eb >> 5, a.ef_tiles, a.ef_waves) != efw_owner((ee - 1) >> 5, a.ef_tiles, a.ef_waves)) ? 1 : 0;
  }
  for (int idx = e; !a.raw && idx < a.n_det * (D_H / 4); idx += gridDim.x * 256) {
    const int i = idx >> 6, q = idx & 63;
    float sc = a.scores[i] * a.mult, sn = sc;          // x * 1.0f is exact: the default multiplier changes no bit
    int rc = 0, rn = 1;
    if (a.multiclass) {                                // scatter_nd one-hot x score (network.py:413-419)
      const int cl = a.classes[i] - 1;
      if (cl >= 0 && cl < a.cprime) { rc = cl; rn = a.cprime + cl; } else { sc = 0.f; sn = 0.f; rn = a.cprime; }
    }
    const float4 wc = *reinterpret_cast<const float4*>(a.w1 + (size_t)rc * D_H + 4 * q);
    const float4 wn = *reinterpret_cast<const float4*>(a.w1 + (size_t)rn * D_H + 4 * q);
    const float4 bb = *reinterpret_cast<const float4*>(a.b1 + 4 * q);
    *reinterpret_cast<float4*>(a.tc + (size_t)i * D_H + 4 * q) = make_float4(sc * wc.x + bb.x, sc * wc.y + bb.y, sc * wc.z + bb.z, sc * wc.w + bb.w);
    *reinterpret_cast<float4*>(a.tn + (size_t)i * D_H + 4 * q) = make_float4(sn * wn.x, sn * wn.y, sn * wn.z, sn * wn.w);
  }
  if (e >= a.n_edge) {
    if (e < a.n_edge + 64) a.edge_nz[e] = a.n_det;
    return;
  }
  const float log2f_ = 0.69314718f;          // float32(np.log(2.0)) network.py:447
  const int c = a.edge_c[e], n = a.edge_n[e];
  const float4 cb = a.dets[c], nb = a.dets[n];
  const float c_w = cb.z - cb.x, c_h = cb.w - cb.y;
  const float n_w = nb.z - nb.x, n_h = nb.w - nb.y;
  const float c_scale = (c_w + c_h) / 2.0f;
  const float c_cx = cb.x + c_w / 2.0f, c_cy = cb.y + c_h / 2.0f;
  const float n_cx = nb.x + n_w / 2.0f, n_cy = nb.y + n_h / 2.0f;
  float xd = n_cx - c_cx, yd = n_cy - c_cy;
  const float l2 = sqrtf(xd * xd + yd * yd) / c_scale;
  xd = xd / c_scale; yd = yd / c_scale;
  const float wd = logf(n_w / c_w) / log2f_;
  const float hd = logf(n_h / c_h) / log2f_;
  const float ad = (logf(n_w / n_h) - logf(c_w / c_h)) / log2f_;
  float4* gp = reinterpret_cast<float4*>(a.geo + (size_t)e * 8);
  const float m = a.mult;                   // x * 1.0f is exact: the default changes no bit
  gp[0] = make_float4(a.edge_iou[e] * m, xd * m, yd * m, l2 * m);
  gp[1] = make_float4(wd * m, hd * m, ad * m, 0.f);
  if (a.raw) {
    float4* pp = reinterpret_cast<float4*>(a.pw + (size_t)e * D_E);
    pp[0] = gp[0]; pp[1] = gp[1];
#pragma unroll
    for (int k = 2; k < D_E / 4; ++k) pp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.edge_nz[e] = c == n ? a.n_det + 1 + c : n;
    return;
  }
  a.edge_nz[e] = c == n ? a.n_det : n;
}

struct PwFwdArgs {
  int n_edge;
  int cprime;
  const float* geo;
  const int* edge_c; const int* edge_n;
  const float* tc; const float* tn;     // [N,256] per-detection score terms of fc1 (edge_geometry): tc includes the bias
  const float* w1;                      // natural [dpw,256]: the 7 geometry rows are read here
  const float* w2t; const float* b2;    // fc2 as operand fragments (pack_transpose: fragment-major), [256]
  const float* w3; const float* b3;     // natural [256,32] (the resident fc3 operand), [32]
  float* h1; float* h2; float* pw;
};


// ------------------------------------------------------------------------------------------
// pw_fwd2: _pw_feats_fc with the WEIGHTS RESIDENT IN REGISTERS (round 5; it replaces pw_fwd, which streamed W2 from L2): one 8-wave workgroup per CU (two waves per
// SIMD, 256 registers each), wave w owns output features [32 w, 32 w + 32) of fc2 for every tile the workgroup walks and keeps
// its 256 x 32 slice of W2 (128 registers), its 32 x 32 slice of W3 (16) and its bias pieces (16) for the whole kernel -- fc2's
// weight stream from L2 (the 11 % of round 4's probe, and 32 operand registers of pipeline) is gone.
//   * fc2 is computed TRANSPOSED (h2^T = W2^T . h1^T: A = the resident weights, B = the h1 tile in LDS): the accumulators hold
//     lane = edge, registers = features, which IS the A operand of fc3 (as edge_fwd_w's layers) -- no h2 round trip through
//     LDS; fc3 is K-split over the eight waves (16 MFMAs each), the partial sums meet in LDS;
//   * a tile is 32 edges; its h1 is produced one tile ahead, INSIDE the previous tile's fc2 stream (8 table gathers requested
//     at the top of the stream, 4 MFMAs + 4 LDS stores in its middle), into the other of two h1 buffers; the partial sums of
//     tile t are reduced and stored inside tile t + 1's stream (two partial buffers): ONE workgroup barrier per tile, and
//     nothing but the fc3 tail (16 MFMAs, 16 LDS stores) runs without MFMAs of the same wave around it;
//   * training: h1 leaves through LDS as whole 1 KB rows (four per wave); h2 leaves straight from the accumulators, 16 bytes
//     per lane -- a store instruction covers 32 bytes of each of 32 rows, four of them complete a 128-byte line.
// No conditional memory operation in the tile loop (the compiler's wait insertion merges control-flow joins conservatively):
// the one-tile-ahead work of the last tile recomputes a clamped tile, the first tile's "previous" output row is a slack row.
constexpr int PW2_T = 32;
constexpr int PW2_LD = D_H + 4;                  // padded h1 row: conflict-free 16-byte reads at one column of 32 rows
constexpr int PW2_HF = PW2_T * PW2_LD;           // floats of one h1 tile
constexpr int PW2_PF = 8 * PW2_T * D_E;          // floats of one set of fc3 partial sums (8 waves x [32][32])
constexpr size_t kPwFwd2Smem = (size_t)(2 * PW2_HF + 2 * PW2_PF) * sizeof(float);

__device__ __forceinline__ void pw2_barrier() {
  // LDS only: this wave's LDS stores are done, then the workgroup meets.  (__syncthreads() also drains vmcnt: the h1 / h2 / pw
  // stores of a tile would be waited for at every barrier.)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// global accesses of pw_fwd2 as (wave-uniform 64-bit base, SGPRs) + (32-bit per-lane byte offset, one loop-invariant VGPR): the
// `saddr + voffset` form costs no vector instruction per access (base[row * ld + lane] costs three to five 64-bit ones)
__device__ __forceinline__ void pw2_st4(float* ubase, unsigned off, const float4& v) { *reinterpret_cast<float4*>(reinterpret_cast<char*>(ubase) + off) = v; }
// the same as a non-temporal (streaming) store: activations kept for the backward pass are read once, a whole forward pass later --
// they should not evict the table rows and weight fragments the kernel gathers from L2
#ifndef PW3_NT
#define PW3_NT 0      /* measured on MI355X: streaming stores of h1 / h2 make the training kernel 10 % SLOWER (1.34 -> 1.48 ms) */
#endif
__device__ __forceinline__ void pw3_st4(float* ubase, unsigned off, const float4& v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  f4v* p = reinterpret_cast<f4v*>(reinterpret_cast<char*>(ubase) + off);
  const f4v x = {v.x, v.y, v.z, v.w};
  if (PW3_NT) __builtin_nontemporal_store(x, p); else *p = x;
}
__device__ __forceinline__ void pw2_st2(float* ubase, unsigned off, const float2& v) { *reinterpret_cast<float2*>(reinterpret_cast<char*>(ubase) + off) = v; }

template <bool TRAINING>
__global__ void __launch_bounds__(512) pw_fwd2(const PwFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                              // [2][32][260]
  float* sP = smem + 2 * PW2_HF;                 // [2][8][32][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int nt = (a.n_edge + PW2_T - 1) / PW2_T, nwg = gridDim.x;
  // XCD-aware contiguous ranges (as edge_fwd_w): XCD x walks the x-th eighth of the edge list, its table rows stay in its L2
  const int lb = (nwg & 7) == 0 ? (int)((blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int t0 = range_begin(lb, nt, nwg), t1 = range_begin(lb + 1, nt, nwg);
  if (t0 >= t1) return;
  const int last = a.n_edge - 1;

  // ---- resident operands
  // W2 as the A operand of the transposed product: lane (feature col, half) supplies W2[8 s + 4 half + t][32 wave + col] at
  // step (s, t) -- exactly the lane's 16-byte piece of the fragment-major copy (pack_transpose)
  f32x4 w2r[32];
  {
    const float* bp = a.w2t + ((size_t)wave * (32 * 64) + lane) * 4;
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) w2r[s_] = *reinterpret_cast<const f32x4*>(bp + s_ * 256);
  }
  // fc3: A = the rectified accumulators (lane = edge, register r = feature 32 wave + crow(r, half)), B = W3[that feature][col]
  // (requested late in every tile's stream, when the registers of the next tile's gathers are free again: 32 registers
  // that would not fit beside the 128 of W2 otherwise; L1-resident after the first tile)
  float w3r[16];
  float4 b2q[4];
  const float* w3p = a.w3 + (size_t)(32 * wave + 4 * half) * D_E + col;
  const float* b2p = a.b2 + 32 * wave + 4 * half;
  const int geo_row0 = 2 * a.cprime;
  float wgA[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int k = 4 * half + s_;
    wgA[s_] = k < 7 ? a.w1[(size_t)(geo_row0 + k) * D_H + 32 * wave + col] : 0.f;
  }
  const int er = tid >> 4, j0 = (tid & 15) * 2;   // the (edge, output pair) of a tile this thread reduces
  const float b3a = a.b3[j0], b3b = a.b3[j0 + 1];
  const unsigned fo = (unsigned)(32 * wave + 4 * half) * 4u;   // the lane's first feature piece in a table row (bytes)
  const unsigned pw_lo = (unsigned)(er * D_E + j0) * 4u;       // the thread's output pair inside a tile of pw
  const unsigned h2_lo = (unsigned)(col * D_H + 4 * half) * 4u;   // the lane's row and half inside a tile of h2

#define PW2_EDGE(u_) min(min((u_), nt - 1) * PW2_T + col, last)
#define PW2_REQUEST(c_, n_)                                                                        \
  do {                                                                                             \
    const unsigned oc_ = (unsigned)(c_) * (D_H * 4u) + fo, on_ = (unsigned)(n_) * (D_H * 4u) + fo; \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) { tcv[g] = ldg4_b(a.tc, oc_ + 32u * g); tnv[g] = ldg4_b(a.tn, on_ + 32u * g); } \
  } while (0)
  // h1^T piece of this wave: accumulators start from the two table rows, 4 MFMAs add the geometry term (K = 8), one integer
  // max rectifies, the rows go to LDS in the [edge][feature] layout fc2 reads
#define PW2_FC1(dstH_)                                                                             \
  do {                                                                                             \
    f32x16 h_;                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                \
      h_[4 * g + 0] = tcv[g].x + tnv[g].x; h_[4 * g + 1] = tcv[g].y + tnv[g].y;                    \
      h_[4 * g + 2] = tcv[g].z + tnv[g].z; h_[4 * g + 3] = tcv[g].w + tnv[g].w;                    \
    }                                                                                              \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[0], gv.x, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[1], gv.y, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[2], gv.z, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[3], gv.w, h_, 0, 0, 0);                          \
    float* d_ = (dstH_) + col * PW2_LD + 32 * wave + 4 * half;                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                  \
      *reinterpret_cast<float4*>(d_ + 8 * g) = make_float4(relu_bits(h_[4 * g]), relu_bits(h_[4 * g + 1]), relu_bits(h_[4 * g + 2]), relu_bits(h_[4 * g + 3])); \
  } while (0)

  // forward-only: the fc3 partial product of a tile is DEFERRED into the next tile's stream (its rectified accumulators and W3
  // pieces kept across the barrier) -- the tile then ends with its last fc2 MFMAs instead of 16 MFMAs + 16 LDS stores + their
  // drain with nothing of this wave's in the pipe: 1.65 -> 1.56 ms.  Not in a training step: there the same move is +1 % (the
  // h2 stores at the tile's end want the fc3 MFMAs behind them), and with the h2 stores deferred as well +5 % (1.74 -> 1.82 ms).
  constexpr bool DEFER = !TRAINING;
  f32x16 hprev = zero16();        // (DEFER) the previous tile's rectified fc2 accumulators: fc3's A operand
  if (DEFER) {
#pragma unroll
    for (int r = 0; r < 16; ++r) w3r[r] = w3p[crow(r, 0) * D_E];
  }
  // ---- front: the first tile's h1, the records of the second
  float4 tcv[4], tnv[4];
  f32x4 gv;
  int c1, n1;
  {
    const unsigned e = (unsigned)PW2_EDGE(t0);
    const int c0 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e), n0 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e);
    { const float4 g_ = ldg4_b(a.geo, 32u * e + 16u * half); gv = f32x4{g_.x, g_.y, g_.z, g_.w}; }
    const unsigned e1 = (unsigned)PW2_EDGE(t0 + 1);
    c1 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e1); n1 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e1);
    PW2_REQUEST(c0, n0);
    PW2_FC1(sH);
  }
  pw2_barrier();

  for (int t = t0, it = 0; t < t1; ++t, ++it) {
    const int e0 = t * PW2_T;
    const float* Hc = sH + (it & 1) * PW2_HF;
    float* Hn = sH + ((it & 1) ^ 1) * PW2_HF;
    float* Pc = sP + (it & 1) * PW2_PF;
    const float* Pp = sP + ((it & 1) ^ 1) * PW2_PF;
    const float* hb = Hc + col * PW2_LD + 4 * half;
#define PW2_HB(f_) (*reinterpret_cast<const f32x4*>(hb + 8 * (f_)))
    f32x4 bq0 = PW2_HB(0), bq1 = PW2_HB(1), bq2;
    // requests of the next tile (its h1 is formed in the middle of this tile's stream) and the records of the one after it
    int c2, n2;
#define PW2_NEXT_REQUESTS()                                                                             \
    {                                                                                                   \
      PW2_REQUEST(c1, n1);                                                                              \
      const unsigned e1 = (unsigned)PW2_EDGE(t + 1);                                                    \
      { const float4 g_ = ldg4_b(a.geo, 32u * e1 + 16u * half); gv = f32x4{g_.x, g_.y, g_.z, g_.w}; }   \
      const unsigned e2 = (unsigned)PW2_EDGE(t + 2);                                                    \
      c2 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e2); n2 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e2); \
    }
    f32x16 pacc = zero16();
    if (!DEFER) PW2_NEXT_REQUESTS();        // (DEFER: behind the previous tile's fc3 MFMAs, whose operands need the registers first)
    f32x16 acc;
    float4 hrow;
#define PW2_MMA4(bq_, w_, first_)                                                                       \
  do {                                                                                                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((w_).x, (bq_).x, (first_) ? zero16() : acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((w_).y, (bq_).y, acc, 0, 0, 0);                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((w_).z, (bq_).z, acc, 0, 0, 0);                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((w_).w, (bq_).w, acc, 0, 0, 0);                          \
  } while (0)
    // ---- fc2: 32 fragments of 4 MFMAs; the h1 operand two fragments ahead; the rest of the tile's work in between
#pragma unroll
    for (int f = 0; f < 32; ++f) {
      if (f + 2 < 32) { if (f % 3 == 0) bq2 = PW2_HB(f + 2); else if (f % 3 == 1) bq0 = PW2_HB(f + 2); else bq1 = PW2_HB(f + 2); }
      if (DEFER) {
      // the PREVIOUS tile's fc3 partial product (its rectified accumulators and W3 pieces were kept across the barrier): four MFMAs
      // beside each of the first four fragments, the partial sums to LDS behind them -- the tile no longer ends with 16 MFMAs + 16 LDS
      // stores + their drain in front of the barrier with nothing of this wave's in the pipe
      if (f < 4) {
#pragma unroll
        for (int r = 4 * f; r < 4 * f + 4; ++r) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(hprev[r], w3r[r], r == 0 ? zero16() : pacc, 0, 0, 0);
      }
      if (f == 5) {
        float* d_ = sP + ((it & 1) ^ 1) * PW2_PF + (wave * PW2_T + 4 * half) * D_E + col;      // P[(it - 1) & 1]: the previous tile's
#pragma unroll
        for (int r = 0; r < 16; ++r) d_[crow(r, 0) * D_E] = pacc[r];
        PW2_NEXT_REQUESTS();
      }
      if (f == 9) {
        // the fc3 of the tile BEFORE the previous one (its partial sums were published by the last barrier)
        const float* pp = Pc + er * D_E + j0;                   // P[it & 1] = P[(it - 2) & 1]
        float2 s_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); s_.x += v.x; s_.y += v.y; }
        const int ep = it > 1 ? e0 - 2 * PW2_T : a.n_edge + 32;
        pw2_st2(a.pw + (size_t)ep * D_E, pw_lo, make_float2(fmaxf(s_.x + b3a, 0.f), fmaxf(s_.y + b3b, 0.f)));
      }
      } else if (f == 3) {
        // the previous tile's fc3: sum of the eight waves' partial sums, bias, ReLU, 8 bytes per thread (a tile's 4 KB in a row);
        // the first tile of the range has no predecessor: a slack row takes the store
        const float* pp = Pp + er * D_E + j0;
        float2 s_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); s_.x += v.x; s_.y += v.y; }
        const int ep = it > 0 ? e0 - PW2_T : a.n_edge + 32;
        pw2_st2(a.pw + (size_t)ep * D_E, pw_lo, make_float2(fmaxf(s_.x + b3a, 0.f), fmaxf(s_.y + b3b, 0.f)));
      }
      // training: four whole h1 rows per wave, read from LDS one fragment before they are stored (stored straight from fc1's
      // accumulators instead, as 16-byte pieces like h2: +1.6 %, measured)
      if (TRAINING && f >= 7 && f < 11) pw2_st4(a.h1 + (size_t)(e0 + 4 * wave + (f - 7)) * D_H, 16u * lane, hrow);
      if (TRAINING && f >= 6 && f < 10) hrow = *reinterpret_cast<const float4*>(Hc + (4 * wave + (f - 6)) * PW2_LD + 4 * lane);
      if (f == (DEFER ? 20 : 16)) PW2_FC1(Hn);
      if (f == (DEFER ? 26 : 24)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w3r[r] = w3p[crow(r, 0) * D_E];
#pragma unroll
        for (int g = 0; g < 4; ++g) b2q[g] = *reinterpret_cast<const float4*>(b2p + 8 * g);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (f % 3 == 0) PW2_MMA4(bq0, w2r[f], f == 0); else if (f % 3 == 1) PW2_MMA4(bq1, w2r[f], false); else PW2_MMA4(bq2, w2r[f], false);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- bias, ReLU, h2 rows, this wave's K = 32 slice of fc3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[4 * g + 0] = relu_bits(acc[4 * g + 0] + b2q[g].x); acc[4 * g + 1] = relu_bits(acc[4 * g + 1] + b2q[g].y);
      acc[4 * g + 2] = relu_bits(acc[4 * g + 2] + b2q[g].z); acc[4 * g + 3] = relu_bits(acc[4 * g + 3] + b2q[g].w);
    }
    if (TRAINING) {
      float* d_ = a.h2 + (size_t)e0 * D_H + 32 * wave;          // (uniform; the lane's row and half: h2_lo)
#pragma unroll
      for (int g = 0; g < 4; ++g) pw2_st4(d_, h2_lo + 32u * g, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]));
    }
    if (DEFER) {
      hprev = acc;
    } else {
      pacc = zero16();
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], w3r[r], pacc, 0, 0, 0);
      float* d_ = Pc + (wave * PW2_T + 4 * half) * D_E + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) d_[crow(r, 0) * D_E] = pacc[r];
    }
    c1 = c2; n1 = n2;
    pw2_barrier();
  }
  {
    const int itl = t1 - t0 - 1;
    if (DEFER) {
      // drain: the last tile's fc3 product, then the reduction of the tile before it (the last tile's own follows below)
      f32x16 pacc = zero16();
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(hprev[r], w3r[r], pacc, 0, 0, 0);
      {
        float* d_ = sP + (itl & 1) * PW2_PF + (wave * PW2_T + 4 * half) * D_E + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) d_[crow(r, 0) * D_E] = pacc[r];
      }
      pw2_barrier();
      if (itl > 0) {
        const float* pp = sP + ((itl & 1) ^ 1) * PW2_PF + er * D_E + j0;
        float2 s_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); s_.x += v.x; s_.y += v.y; }
        pw2_st2(a.pw + (size_t)((t1 - 2) * PW2_T) * D_E, pw_lo, make_float2(fmaxf(s_.x + b3a, 0.f), fmaxf(s_.y + b3b, 0.f)));
      }
    }
    // the last tile's fc3
    const float* pp = sP + (itl & 1) * PW2_PF + er * D_E + j0;
    float2 s_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); s_.x += v.x; s_.y += v.y; }
    pw2_st2(a.pw + (size_t)((t1 - 1) * PW2_T) * D_E, pw_lo, make_float2(fmaxf(s_.x + b3a, 0.f), fmaxf(s_.y + b3b, 0.f)));
  }
#undef PW2_EDGE
#undef PW2_NEXT_REQUESTS
#undef PW2_REQUEST
#undef PW2_FC1
#undef PW2_HB
#undef PW2_MMA4
}

// ------------------------------------------------------------------------------------------
// pw_fwd3 (round 6): pw_fwd2 ON THE bf16 PIPE.  fc2 and fc3 form every fp32 product as six bf16 products of exact three-term
// splits (common.hpp mma6: v_mfma_f32_32x32x16_bf16, fp32 accumulation, the fp32 MFMA's error against fp64 -- measured on this
// kernel's own operands by tests/test_gpu_bf16x3.py): 96 + 12 MFMAs of 32 cycles per wave and tile where pw_fwd2 issues 128 + 16 of 64.
// Same workgroup shape (one 8-wave workgroup per CU, wave w owns output features [32 w, 32 w + 32) of fc2, persistent over an
// XCD-aware tile range, ONE LDS-only barrier per tile, the next tile's fc1 formed inside this tile's stream), and:
//   * W2 as three bf16 terms is 192 registers per lane: the high and middle terms are RESIDENT (128 registers, loaded once per
//     kernel from pack_pw_bf16's fragment-major copy), the low term -- one product of the six -- streams from L2 through a four-deep
//     register ring (tools/pw_bf16x3_probe.hip: as fast as all three resident);
//   * the h1 tile lives in LDS as three bf16 terms in FRAGMENT-MAJOR order [term][k-step 16][lane 64] x 16 bytes: fc1's
//     accumulators (lane = edge, registers = features) are split in registers by the wave that formed them -- a k-step's slots are
//     the register order of an accumulator column (frag_feat), as in edge_fwd_w -- and land as six 16-byte stores; every wave's B
//     operand of k-step s is then ONE contiguous kilobyte per term: no swizzle, no padding, immediate offsets;
//   * fc2 is still computed transposed, so its rectified accumulators are fc3's A operand after one more in-register split;
//     fc3's B operand (the wave's 32 x 32 slice of W3, three terms) is re-requested late in every tile's stream;
//   * fc1 itself stays on the fp32 MFMA (4 per tile): h1 is bit-identical to pw_fwd2's;
//   * training: h1 and h2 leave straight from the accumulators as 16-byte pieces.
// LDS: 2 x 48 KB of h1 terms + 2 x 32 KB of fc3 partial sums = 160 KB, all of it.
// vmcnt is one in-order counter: a ring load issued behind a slow request (the table gathers, a store) returns behind it, so the
// slow requests of a tile are issued together at two places of the stream and the ring is four k-steps (~1 us) deep.
constexpr int PW3_TERM = 16 * 64 * 4;            // 32-bit words of one term of an h1 tile: [16 k-steps][64 lanes][4]
constexpr int PW3_HW = 3 * PW3_TERM;             // one h1 tile (three terms): 48 KB
constexpr size_t kPwFwd3Smem = (size_t)(2 * PW3_HW + 2 * PW2_PF) * 4;
static_assert(kPwFwd3Smem == 160 * 1024, "pw_fwd3 uses the whole LDS of a CU");

struct PwFwd3Args {
  PwFwdArgs p;
  const unsigned* wbf;     // pack_pw_bf16's arrays (common.hpp PWBF_*)
  GNET_TRACE_FIELD
};

__device__ __forceinline__ u32x4 lds_q(const unsigned* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ldg_q(const unsigned* __restrict__ ubase, unsigned byte_off) {
  return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ubase) + byte_off);
}

#ifndef PW3_X
#define PW3_X 0           /* ablation mask of measurement builds: 1 = no h1 stores, 2 = no h2 stores (wrong results, durations only) */
#endif
#ifndef PW3_S_FC1
#define PW3_S_FC1 9       /* k-step in front of which the next tile's fc1 is formed (5 / 9 / 12 / 13 measured: 1.38 / 1.35 / 1.40 / 1.41 ms) */
#define PW3_S_W3 13       /* k-step in front of which fc3's weight fragments are re-requested (behind fc1: its registers are free then) */
#endif
template <bool TRAINING>
__global__ void __launch_bounds__(512) pw_fwd3(const PwFwd3Args aa) {
  const PwFwdArgs& a = aa.p;
  extern __shared__ __attribute__((aligned(16))) unsigned smem3[];
  unsigned* sH = smem3;                                               // [2][3][16][64][4]
  float* sP = reinterpret_cast<float*>(smem3 + 2 * PW3_HW);           // [2][8][32][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int nt = (a.n_edge + PW2_T - 1) / PW2_T, nwg = gridDim.x;
  const int lb = (nwg & 7) == 0 ? (int)((blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int t0 = range_begin(lb, nt, nwg), t1 = range_begin(lb + 1, nt, nwg);
  if (t0 >= t1) return;
  const int last = a.n_edge - 1;
  GSTAMP(aa, 0);

  // ---- resident operands: the high and middle terms of the wave's 256 x 32 slice of W2 (A operand of the transposed product)
  u32x4 wh[16], wm[16];
  const unsigned* w2a = aa.wbf + PWBF_W2A + (size_t)wave * (16 * 64 * 4);
  constexpr unsigned TERM2 = 8u * 16 * 64 * 4 * 4;                     // bytes between the terms of W2A
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) { wh[s_] = ldg_q(w2a + 256 * s_, 16u * lane); wm[s_] = ldg_q(w2a + TERM2 / 4 + 256 * s_, 16u * lane); }
  // the low term's slice [16 k-steps][64 lanes] x 16 bytes and fc3's operand are read as (kernel-argument base, uniform) + (one lane
  // offset that carries the wave's slice, made opaque once per tile: the compiler otherwise hoists one 64-bit lane address per
  // k-step out of the loop and spills them)
  const unsigned* w2lo = aa.wbf + PWBF_W2A + 2 * (8 * 16 * 64 * 4);
  const unsigned* w3b = aa.wbf + PWBF_W3B;
  constexpr unsigned TERM3 = 8u * 2 * 64 * 4 * 4;
  unsigned ro = 16u * lane + (unsigned)wave * (16u * 64 * 16), ro3 = 16u * lane + (unsigned)wave * (2u * 64 * 16);
  Bf3 w3q[2];
  // fc2's bias rides in ONE more MFMA at the head of every tile's chain (sixteen registers of bias pieces do not fit beside the
  // weights): A = the three terms of b2[32 wave + col] in k-slots 0-2 of the lower half-wave, B = 1.0 in those slots -- the
  // accumulators start from hi + mid + lo = the bias, exactly
  u32x4 biasA;
  {
    unsigned ph, pm, pl;
    split3_pk(a.b2[32 * wave + col], 0.f, ph, pm, pl);
    biasA = half ? u32x4{0u, 0u, 0u, 0u} : u32x4{(ph & 0xffffu) | (pm << 16), pl & 0xffffu, 0u, 0u};
  }
  const int geo_row0 = 2 * a.cprime;
  float wgA[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int k = 4 * half + s_;
    wgA[s_] = k < 7 ? a.w1[(size_t)(geo_row0 + k) * D_H + 32 * wave + col] : 0.f;
  }
  const int er = tid >> 4, j0 = (tid & 15) * 2;   // the (edge, output pair) of a tile this thread reduces
  const float b3a = a.b3[j0], b3b = a.b3[j0 + 1];
  unsigned fo = (unsigned)(32 * wave + 4 * half) * 4u;   // the lane's first feature piece in a table row (bytes)
  unsigned pw_lo = (unsigned)(er * D_E + j0) * 4u;       // the thread's output pair inside a tile of pw
  unsigned h_lo = (unsigned)(col * D_H + 4 * half) * 4u; // the lane's row and half inside a tile of h1 / h2
  unsigned hmask = half ? 0u : 0xffffffffu;              // (the lower half-wave carries the bias slots)

#define PW3_EDGE(u_) min(min((u_), nt - 1) * PW2_T + col, last)
#define PW3_REQUEST(c_, n_)                                                                        \
  do {                                                                                             \
    const unsigned oc_ = (unsigned)(c_) * (D_H * 4u) + fo, on_ = (unsigned)(n_) * (D_H * 4u) + fo; \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) { tcv[g] = ldg4_b(a.tc, oc_ + 32u * g); tnv[g] = ldg4_b(a.tn, on_ + 32u * g); } \
  } while (0)
  // h1^T piece of this wave (fp32 MFMA, pw_fwd2's bits): table rows + the K = 8 geometry product, rectified; training: the fp32 rows
  // leave as four 16-byte pieces; then the in-register split and six 16-byte LDS stores: k-steps 2 wave, 2 wave + 1 of the three terms
#define PW3_FC1(dstH_, e0_)                                                                        \
  do {                                                                                             \
    f32x16 h_;                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                \
      h_[4 * g + 0] = tcv[g].x + tnv[g].x; h_[4 * g + 1] = tcv[g].y + tnv[g].y;                    \
      h_[4 * g + 2] = tcv[g].z + tnv[g].z; h_[4 * g + 3] = tcv[g].w + tnv[g].w;                    \
    }                                                                                              \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[0], gv.x, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[1], gv.y, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[2], gv.z, h_, 0, 0, 0);                          \
    h_ = __builtin_amdgcn_mfma_f32_32x32x2f32(wgA[3], gv.w, h_, 0, 0, 0);                          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) h_[r] = relu_bits(h_[r]);                       \
    if (TRAINING && !(PW3_X & 1)) {                                                                \
      float* g_ = a.h1 + (size_t)(e0_) * D_H + 32 * wave;                                          \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) pw3_st4(g_, h_lo + 32u * g, make_float4(h_[4 * g], h_[4 * g + 1], h_[4 * g + 2], h_[4 * g + 3])); \
    }                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                \
      const Bf3 t_ = split3_8(f32x4{h_[8 * q], h_[8 * q + 1], h_[8 * q + 2], h_[8 * q + 3]},       \
                              f32x4{h_[8 * q + 4], h_[8 * q + 5], h_[8 * q + 6], h_[8 * q + 7]});  \
      unsigned* d_ = (dstH_) + ((2 * wave + q) * 64 + lane) * 4;                                   \
      *reinterpret_cast<u32x4*>(d_) = t_.h;                                                        \
      *reinterpret_cast<u32x4*>(d_ + PW3_TERM) = t_.m;                                             \
      *reinterpret_cast<u32x4*>(d_ + 2 * PW3_TERM) = t_.l;                                         \
    }                                                                                              \
  } while (0)

  // ---- front: the first tile's h1, the records of the second, the first four pieces of the low-term ring
  float4 tcv[4], tnv[4];
  f32x4 gv;
  int c1, n1;
  u32x4 ring[4];
  {
    const unsigned e = (unsigned)PW3_EDGE(t0);
    const int c0 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e), n0 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e);
    { const float4 g_ = ldg4_b(a.geo, 32u * e + 16u * half); gv = f32x4{g_.x, g_.y, g_.z, g_.w}; }
    const unsigned e1 = (unsigned)PW3_EDGE(t0 + 1);
    c1 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e1); n1 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e1);
    PW3_REQUEST(c0, n0);
    PW3_FC1(sH, t0 * PW2_T);
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = ldg_q(w2lo, ro + 1024u * i);
  }
  pw2_barrier();

  for (int t = t0, it = 0; t < t1; ++t, ++it) {
    const int e0 = t * PW2_T;
    const unsigned* Hc = sH + (it & 1) * PW3_HW + lane * 4;
    unsigned* Hn = sH + ((it & 1) ^ 1) * PW3_HW;
    float* Pc = sP + (it & 1) * PW2_PF;
    const float* Pp = sP + ((it & 1) ^ 1) * PW2_PF;
    int c2, n2;
    if (it == 5) { GSTAMP(aa, 1); GSTAMP_W(aa, 8, 256); }
    // (lane offsets made opaque once per tile: what is derived from them stays inside the loop instead of being hoisted as 64-bit
    // lane addresses -- and spilled: a scratch reload in this loop is a vmcnt(0), i.e. a drain of the ring)
    asm volatile("" : "+v"(ro), "+v"(ro3), "+v"(fo), "+v"(pw_lo), "+v"(h_lo), "+v"(hmask));
    // the h1 operand of a k-step: three 16-byte LDS reads; the high term (read by the first and the last product) is requested a
    // whole k-step ahead, the low and the middle term into their own registers right behind the last product that reads them
    u32x4 bh[2], bm, bl;
    bh[0] = lds_q(Hc); bm = lds_q(Hc + PW3_TERM); bl = lds_q(Hc + 2 * PW3_TERM);
    f32x16 acc;
    {
      const u32x4 ones = u32x4{hmask & 0x3f803f80u, hmask & 0x00003f80u, 0u, 0u};
      acc = mfma_bf16(biasA, ones, zero16());
    }
    // ---- fc2: 16 k-steps of six MFMAs; the h1 operand one k-step ahead, the low weight term four; the rest of the tile's work between
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const u32x4 al = ring[s_ & 3];
      ring[s_ & 3] = ldg_q(w2lo, ro + 1024u * ((s_ + 4) & 15));          // (the last four: the next tile's first four)
      if (s_ == 0) {
        // the slow requests of the tile's first half, together: the next tile's table rows and geometry, the records of the tile
        // after it, the previous tile's fc3 (sum of the eight waves' partial sums, bias, ReLU, 8 bytes per thread; the first
        // tile of the range has no predecessor: a slack row takes the store)
        PW3_REQUEST(c1, n1);
        const unsigned e1 = (unsigned)PW3_EDGE(t + 1);
        { const float4 g_ = ldg4_b(a.geo, 32u * e1 + 16u * half); gv = f32x4{g_.x, g_.y, g_.z, g_.w}; }
        const unsigned e2 = (unsigned)PW3_EDGE(t + 2);
        c2 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * e2); n2 = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_n), 4u * e2);
        const float* pp = Pp + er * D_E + j0;
        float2 r_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); r_.x += v.x; r_.y += v.y; }
        const int ep = it > 0 ? e0 - PW2_T : a.n_edge + 32;
        pw2_st2(a.pw + (size_t)ep * D_E, pw_lo, make_float2(fmaxf(r_.x + b3a, 0.f), fmaxf(r_.y + b3b, 0.f)));
      }
      if (s_ == PW3_S_FC1 && it == 5) { GSTAMP(aa, 2); GSTAMP_W(aa, 9, 256); }
#ifdef PW3_STAGGER
      // (measurement: the two waves of a SIMD -- w and w + 4 -- form the next tile's fc1 and issue its h1 stores at DIFFERENT k-steps,
      //  so that one's wait for the stores' acknowledgement falls into the other's MFMA stream)
      if (s_ == PW3_STAGGER_A && wave < 4) PW3_FC1(Hn, min(t + 1, nt - 1) * PW2_T);
      if (s_ == PW3_STAGGER_B && wave >= 4) PW3_FC1(Hn, min(t + 1, nt - 1) * PW2_T);
#else
      if (s_ == PW3_S_FC1) PW3_FC1(Hn, min(t + 1, nt - 1) * PW2_T);
#endif
      if (s_ == PW3_S_FC1 + 1 && it == 5) { GSTAMP(aa, 3); GSTAMP_W(aa, 10, 256); }
      if (s_ == PW3_S_W3 && it == 5) { GSTAMP(aa, 4); GSTAMP_W(aa, 11, 256); }
      if (s_ == PW3_S_W3) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          w3q[q].h = ldg_q(w3b, ro3 + 1024u * q); w3q[q].m = ldg_q(w3b, ro3 + TERM3 + 1024u * q); w3q[q].l = ldg_q(w3b, ro3 + 2 * TERM3 + 1024u * q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // common.hpp mma6's sequence (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi), A = the weights, B = the h1 terms
      if (s_ + 1 < 16) bh[(s_ + 1) & 1] = lds_q(Hc + (s_ + 1) * 256);
      acc = mfma_bf16(al, bh[s_ & 1], acc);
      acc = mfma_bf16(wh[s_], bl, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ + 1 < 16) bl = lds_q(Hc + 2 * PW3_TERM + (s_ + 1) * 256);
      acc = mfma_bf16(wm[s_], bm, acc);
      acc = mfma_bf16(wm[s_], bh[s_ & 1], acc);
      acc = mfma_bf16(wh[s_], bm, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ + 1 < 16) bm = lds_q(Hc + PW3_TERM + (s_ + 1) * 256);
      acc = mfma_bf16(wh[s_], bh[s_ & 1], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (it == 5) { GSTAMP(aa, 5); GSTAMP_W(aa, 12, 256); }
    // ---- ReLU (the bias is in the chain), h2 rows, this wave's K = 32 slice of fc3 (two k-steps of the rectified accumulators, split in registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = relu_bits(acc[r]);
    if (TRAINING && !(PW3_X & 2)) {
      float* d_ = a.h2 + (size_t)e0 * D_H + 32 * wave;          // (uniform; the lane's row and half: h_lo)
#pragma unroll
      for (int g = 0; g < 4; ++g) pw3_st4(d_, h_lo + 32u * g, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]));
    }
    {
      f32x16 pacc = zero16();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const Bf3 ha = split3_8(f32x4{acc[8 * q], acc[8 * q + 1], acc[8 * q + 2], acc[8 * q + 3]},
                                f32x4{acc[8 * q + 4], acc[8 * q + 5], acc[8 * q + 6], acc[8 * q + 7]});
        __builtin_amdgcn_sched_barrier(0);
        pacc = mma6(pacc, ha, w3q[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
      float* d_ = Pc + (wave * PW2_T + 4 * half) * D_E + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) d_[crow(r, 0) * D_E] = pacc[r];
    }
    c1 = c2; n1 = n2;
    if (it == 5) { GSTAMP(aa, 6); GSTAMP_W(aa, 13, 256); }
    pw2_barrier();
    if (it == 5) { GSTAMP(aa, 7); GSTAMP_W(aa, 14, 256); }
  }
  {
    // the last tile's fc3
    const int itl = t1 - t0 - 1;
    const float* pp = sP + (itl & 1) * PW2_PF + er * D_E + j0;
    float2 r_ = *reinterpret_cast<const float2*>(pp);
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float2 v = *reinterpret_cast<const float2*>(pp + w * (PW2_T * D_E)); r_.x += v.x; r_.y += v.y; }
    pw2_st2(a.pw + (size_t)((t1 - 1) * PW2_T) * D_E, pw_lo, make_float2(fmaxf(r_.x + b3a, 0.f), fmaxf(r_.y + b3b, 0.f)));
  }
#undef PW3_EDGE
#undef PW3_REQUEST
#undef PW3_FC1
}

// ------------------------------------------------------------------------------------------
// edge_fwd: one wave = one 32-edge tile at a time, 4 independent waves per workgroup sharing the
// weights in LDS.  h1 = relu(P.Wp + rc[c] + (c != n) rn[n]);  h2 = relu(h1.W2 + b2);
// (max, tie count) per (centre, feature) streamed into pm[] (network.py:387-388).
constexpr int E_LD1 = D_E + 4;   // 36
constexpr int E_LD2 = D_P + 4;   // 68

struct EdgeFwdArgs {
  int n_edge;
  const int* edge_c; const int* edge_n;
  const float* pw;               // [E,32]
  const float* rc; const float* rn;   // [N,64]
  const float* w1t;              // transposed pw_fc1 [64][96]: columns 0-31 = pairwise rows
  const float* w2t; const float* b2;   // transposed pw_fc2 [64][64]
  unsigned long long* pm;        // [N,64], zeroed
  const int* row_ptr;            // [N+1]
  const int* edge_nz;            // [E+64] neighbour index, n_det (a zero row of rn) for self pairs and the tail
  float* h1_out;                 // [E+64,64] relu(pw_fc1) (KEEP variant only: tests / debugging)
  float* h2_out;                 // [E+64,64] h1.W2 + b2, the values the segment maximum is taken on (KEEP variant only)
  unsigned long long* parg;      // [N,64] (max bits << 32) | first edge attaining it (training), zeroed
  GNET_TRACE_FIELD
};

// edge_fwd_w: every wave owns whole 32-edge x 64-column tiles, no workgroup barriers in the tile loop, all gathers prefetched one
// tile ahead.  Its fp32 products are formed on the bf16 pipe: every operand as three bf16 terms (exact), six products per k-step of 16
// with fp32 accumulation (common.hpp: split3_8 / mma6) -- 72 v_mfma_f32_32x32x16_bf16 per tile for the 96 v_mfma_f32_32x32x2_f32 of the
// fp32 formulation (12 288 FLOP per edge), the fp32 MFMA's error against fp64 (DESIGN.md lesson 64).
// Segment handling is WAVE-UNIFORM: the rows of a tile are sorted by centre, a ballot yields the segment
// heads, and per segment the wave reduces (max, tie count) over its rows, folds the two half-waves with
// one cross-lane exchange, and flushes a finished centre ONCE -- by a plain 512-byte store when all of
// the centre's edges lie inside this wave's edge range (no other wave touches it), by the exact atomic
// combine only for the (at most two) centres that straddle the range.  No loads or returning atomics sit
// in divergent code, so the in-order vmcnt never drains the prefetches early.
// b = 2 b + (h == s): one compare and one add-with-carry.  (Written as C the compiler assembles the 16 hit bits
// with a select, a shift and an or per element; the carry form is pinned here.)
__device__ __forceinline__ void hit_bit(unsigned& b, float h, float s) {
  asm volatile("v_cmp_eq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(b) : "v"(h), "v"(s) : "vcc");
}

// branch-free (selects only): divergent control flow costs far more than the few extra VALU ops
__device__ __forceinline__ void segmax_merge(float& m, unsigned& k, float m2, unsigned k2) {
  const bool gt = m2 > m, eq = m2 == m;
  k = gt ? k2 : (eq ? k + k2 : k);
  m = gt ? m2 : m;
}

constexpr int EFW_WAVES = 4;      // waves per workgroup; 3 workgroups per CU (launch bounds below)
#ifndef EFW_X
#define EFW_X 0   /* ablation mask of measurement builds (WRONG results, durations only): 1 = no split of P, 2 = no split of h1, 4 = no hit bits, 8 = no ReLU / bias adds, 32 = no segment bookkeeping */
#endif
__device__ __forceinline__ Bf3 efw_fake3(const f32x4 a, const f32x4 b) {
  Bf3 t; t.h = __builtin_bit_cast(u32x4, a); t.m = __builtin_bit_cast(u32x4, b); t.l = t.h; return t;
}
// the three terms of a lane's eight k-slots (16 bytes each, one term-stride apart)
__device__ __forceinline__ Bf3 efw_w3(const unsigned* p, int tstride) {
  Bf3 t;
  t.h = *reinterpret_cast<const u32x4*>(p);
  t.m = *reinterpret_cast<const u32x4*>(p + tstride);
  t.l = *reinterpret_cast<const u32x4*>(p + 2 * tstride);
  return t;
}
// LDS: the two weight matrices as THREE bf16 terms each (common.hpp: Bf3), stored in the k-slot order of their MFMA operand so that a lane
// reads its eight slots of a k-step as one 16-byte word group:
//   sWpT [term][f 64][EFW_LD1]   word h * 8 + j * 4 + w     = slots 2w, 2w+1 of k-step j (of 2), half h:  pf      = 16 j + 8 (s >> 2) + 4 h + (s & 3)
//   sW2T [term][n 64][EFW_LD2]   word h * 16 + j * 4 + w    = slots 2w, 2w+1 of k-step j (of 4), half h:  feature = 32 (j >> 1) + 16 (j & 1) + 8 (s >> 2) + 4 h + (s & 3)
// (the slot order is the order of a lane's layer-1 accumulators: register 8 (j & 1) + s of block j >> 1 -- no shuffle between the layers;
//  row strides of 20 / 36 words: eight lanes' 16-byte reads cover the 32 banks once)
constexpr int EFW_LD1 = 20, EFW_LD2 = 36;
constexpr int EFW_T1 = D_P * EFW_LD1, EFW_T2 = D_P * EFW_LD2;         // words per term
constexpr size_t kEdgeFwdWSmem = (size_t)(3 * EFW_T1 + 3 * EFW_T2 + EFW_WAVES * 2 * D_P) * sizeof(float);

// TRAIN: record the arg-max edge of every (centre, column) for the sparse SegmentMax backward.  The pw_fc1
// activations are NOT kept: the backward pass recomputes them for the ~26 % of the edges that carry gradient
// (backward_edge.hip) -- storing them cost 0.37 GB of HBM writes per launch (2.8 TB/s on an MFMA-bound kernel)
// and 5.9 GB of workspace for the bench batch.  KEEP (tests / debugging) stores them after all.
template <bool TRAIN, bool KEEP>
#ifndef EFW_OCC
#define EFW_OCC 3     /* workgroups per CU (measurement builds: 4 = 128 registers per wave) */
#endif
__global__ void __launch_bounds__(64 * EFW_WAVES, EFW_OCC) edge_fwd_w(const EdgeFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* sWpT = reinterpret_cast<unsigned*>(smem);      // Wp^T as three bf16 terms (layout above)
  unsigned* sW2T = sWpT + 3 * EFW_T1;                      // W2^T likewise
  float* sHw = smem + 3 * EFW_T1 + 3 * EFW_T2;             // per wave: rc rows of the tile's first two centres [2][64]
  GSTAMP(a, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const float bias0 = a.b2[col], bias1 = a.b2[32 + col];
  const int ntiles = (a.n_edge + 31) / 32;
  const int nwaves = gridDim.x * EFW_WAVES;
  // XCD-aware range assignment: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its
  // own L2.  Giving XCD x the x-th contiguous eighth of the edge list keeps the rc / rn rows it gathers (one
  // image's worth when the batch has 8 images) resident in that XCD's L2 instead of all images in every L2.
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int gw = lb * EFW_WAVES + wave;
  const int t0 = efw_begin(gw, ntiles, nwaves), t1 = efw_begin(gw + 1, ntiles, nwaves);     // (common.hpp: sized by dispatch layer)
  float* sRC = sHw + wave * (2 * D_P);
  // Kernel front = two dependent round trips, overlapped with the staging of the weights (at one or two tiles per
  // wave -- a single image -- the front is a third of the kernel):
  //   (1) the first tile's edge records (centre, neighbour row, the edge in front of the range) and its P rows,
  //       then the weights as 16-byte loads into registers (6 per thread);
  //   (2) as soon as the records are back (the in-order counter lets the P rows and weights stay in flight): the rn rows
  //       and the rc rows of the first two centres;
  //   (3) weights -> LDS, barrier.  The gathers of (2) land during (3).
  int nx_c = -1, nx_nz = 0, c_before = -1;
  int n2_c = -1, n2_nz = 0;                                            // the records of the tile after the next one (see the loop)
  f32x4 pa[4];
  const bool have_tiles = t0 < t1;
  if (have_tiles) {
    { const int e = t0 * 32 + col; if (e < a.n_edge) nx_c = a.edge_c[e]; }
    nx_nz = a.edge_nz[t0 * 32 + col];                                  // (the tail of edge_nz is padded)
    {
      const int e = t0 * 32 + 32 + col;
      n2_c = a.edge_c[min(e, a.n_edge - 1)];
      n2_nz = a.edge_nz[min(e, a.n_edge + 63)];
    }
    if (t0 > 0) c_before = a.edge_c[t0 * 32 - 1];
    const float* ap = a.pw + (size_t)min(t0 * 32 + col, a.n_edge - 1) * D_E + 4 * half;
#pragma unroll
    for (int k = 0; k < 4; ++k) pa[k] = *reinterpret_cast<const f32x4*>(ap + 8 * k);
  }
  GSTAMP(a, 13);
  f32x4 wst[6];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int i = tid + 256 * j; wst[j] = *reinterpret_cast<const f32x4*>(a.w1t + (i >> 3) * (D_E + 2 * D_R) + 4 * (i & 7)); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; wst[2 + j] = *reinterpret_cast<const f32x4*>(a.w2t + 4 * i); }
  __builtin_amdgcn_sched_barrier(0);
  // ---- prefetch state for the first tile (lane = edge e0 + col, both half-waves alike)
  float4 rnv[8];                                                       // rn[n][8 g + 4 half .. + 3]
#pragma unroll
  for (int g = 0; g < 8; ++g) rnv[g] = ldg4_b(a.rn, (unsigned)nx_nz * (D_P * 4u) + 32u * g + 16u * half);
  // centre rows of the first two segments of the next tile (A = first centre, B = second or the same): lanes
  // 0-15 fetch the 16-byte chunks of A's row, lanes 16-31 those of B's
  int cA = __builtin_amdgcn_readfirstlane(nx_c), cB = cA, hiA = 32;
  {
    const int prev = __shfl_up(nx_c, 1);
    const unsigned hm = (unsigned)__ballot(col > 0 && nx_c != prev && nx_c >= 0);
    if (hm) { hiA = __builtin_ctz(hm); cB = __builtin_amdgcn_readlane(nx_c, hiA); }
  }
  float4 rcAB = ldg4_b(a.rc, (unsigned)max((lane & 16) ? cB : cA, 0) * (D_P * 4u) + 16u * (lane & 15));
  __builtin_amdgcn_sched_barrier(0);
  // weights -> three bf16 terms -> LDS in slot order: a thread's four consecutive k are slots 4 (g' & 1) .. + 3 of one (half, k-step)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    unsigned ph0, pm0, pl0, ph1, pm1, pl1;
    split3_pk(wst[j].x, wst[j].y, ph0, pm0, pl0);
    split3_pk(wst[j].z, wst[j].w, ph1, pm1, pl1);
    unsigned* dst; int tstride;
    if (j < 2) {
      const int i = tid + 256 * j, f = i >> 3, g = i & 7, kq = g >> 1;                    // pf = 8 kq + 4 (g & 1) + q
      dst = sWpT + f * EFW_LD1 + (g & 1) * 8 + (kq >> 1) * 4 + 2 * (kq & 1); tstride = EFW_T1;
    } else {
      const int i = tid + 256 * (j - 2), n = i >> 4, g = i & 15, kq = (g >> 1) & 3;        // feature = 32 (g >> 3) + 8 kq + 4 (g & 1) + q
      dst = sW2T + n * EFW_LD2 + (g & 1) * 16 + (2 * (g >> 3) + (kq >> 1)) * 4 + 2 * (kq & 1); tstride = EFW_T2;
    }
    *reinterpret_cast<uint2*>(dst) = make_uint2(ph0, ph1);
    *reinterpret_cast<uint2*>(dst + tstride) = make_uint2(pm0, pm1);
    *reinterpret_cast<uint2*>(dst + 2 * tstride) = make_uint2(pl0, pl1);
  }
  GSTAMP(a, 14);
  __syncthreads();
  GSTAMP(a, 1);
  if (!have_tiles) return;
  const int e_begin = t0 * 32, e_end = min(a.n_edge, t1 * 32);         // this wave's edge range
  // Vector instructions are paid beside the MFMA stream (the fp32 MFMA runs on the SIMD's FP32 lanes: tools/mfma_valu_overlap.hip; on
  // the bf16 pipe the kernel is bound by them: 8 per MFMA).  Hence:
  //   * LAYER 1 IS COMPUTED TRANSPOSED: h1^T[f][edge] = Wp^T . P^T (the MFMA's operands swapped), so that its
  //     accumulators -- lane = edge, registers = features 8 g + 4 half + q -- ARE the A operand of layer 2
  //     (a k-step's 16 slots may be ANY 16 features as long as both operands agree: the slot order is the register order).  No LDS
  //     round trip, no barrier and no layout shuffle between the two layers;
  //   * the accumulators start from rc[c] + rn[n] read as the lane's OWN rows (eight 16-byte gathers of rn per
  //     lane and tile instead of 32 4-byte ones; rc rows of the tile's first two centres go through 512 bytes of
  //     LDS and come back as broadcast reads); self pairs and the edge tail are resolved once per batch into
  //     edge_nz (index of a zero row of rn): no clamps, no (c != n) selects;
  //   * the segment maximum is taken on h2 + b2 and rectified once per segment, not per element.
  int cur = -1; float m0 = 0.f, m1 = 0.f; unsigned k0 = 0, k1 = 0;     // running segment (wave-uniform centre)
  int g0 = 0, g1 = 0;                                                  // edge that first attains m0 / m1 (TRAIN)
  // does the first centre of this range start in the previous wave's range?
  bool head_shared = e_begin > 0 && c_before == cA;
  drain_vmem_before_loop();
  GSTAMP(a, 2);
  // (trace builds: the phases of wave 0's FIFTH tile -- slot 3 top, 4 accumulator start (the rn / rc rows have arrived), 5 the next
  //  tile's row requests issued (the P rows have arrived), 6 layer 1, 7 ReLU + layer 2, 8 bias / stores, 9 segment bookkeeping)
#define EFW_STAMP(slot_) do { if (t - t0 == 4) GSTAMP(a, slot_); } while (0)
  for (int t = t0; t < t1; ++t) {
    const int e0 = t * 32;
    const int my_c = nx_c;
    const int nrows = min(32, a.n_edge - e0);
    EFW_STAMP(3);
    const int thiA = hiA;
    // segment heads of THIS tile (bit r set = row r starts a new centre)
    unsigned heads;
    {
      const int prev = __shfl_up(my_c, 1);
      heads = (unsigned)__ballot(half == 0 && col < nrows && (col == 0 || my_c != prev));
    }
    const int nseg = __popc(heads);
    // rc rows of centres A / B -> LDS (this wave's 512 bytes), read back as the lane's own centre row
    if (lane < 32) *reinterpret_cast<float4*>(sRC + 4 * lane) = rcAB;
    // (uniform base + 32-bit byte offset: a 64-bit vector instruction per address is dear beside the MFMA stream)
    // The edge records run TWO tiles ahead: the rows they address (rn, rc) are requested for the next tile right behind this tile's
    // accumulator start, a whole tile before they are read.  (One tile ahead, the records were requested here and consumed by the
    // gathers behind layer 1: a full memory round trip -- s_waitcnt vmcnt(0) -- in the middle of every tile's MFMA stream.)
    // (the loads of the tile after that sit behind layer 1, in front of the P rows: unconditional -- clamped indices, the range test
    // applied here -- so that the compiler's in-order vmcnt arithmetic is exact and no wait covers a request younger than its operand)
    nx_c = (t + 1 < t1 && e0 + 32 + col < a.n_edge) ? n2_c : -1;
    nx_nz = n2_nz;
    wave_lds_sync();
    f32x16 h1a, h1b;                                   // h1^T: lane = edge, register r = feature 8 (r >> 2) + 4 half + (r & 3) [+ 32]
    if (nseg <= 2) {                                   // centre rows were prefetched (A below thiA, B from it)
      const float* rp = sRC + (col < thiA ? 0 : D_P) + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ca = *reinterpret_cast<const float4*>(rp + 8 * g);
        const float4 cb = *reinterpret_cast<const float4*>(rp + 32 + 8 * g);
        h1a[4 * g + 0] = ca.x + rnv[g].x; h1a[4 * g + 1] = ca.y + rnv[g].y; h1a[4 * g + 2] = ca.z + rnv[g].z; h1a[4 * g + 3] = ca.w + rnv[g].w;
        h1b[4 * g + 0] = cb.x + rnv[4 + g].x; h1b[4 * g + 1] = cb.y + rnv[4 + g].y; h1b[4 * g + 2] = cb.z + rnv[4 + g].z; h1b[4 * g + 3] = cb.w + rnv[4 + g].w;
      }
    } else {                                           // many short segments: gather the centre row per edge
      const unsigned oc = (unsigned)max(my_c, 0) * (D_P * 4u) + 16u * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ca = ldg4_b(a.rc, oc + 32u * g);
        const float4 cb = ldg4_b(a.rc, oc + 128u + 32u * g);
        h1a[4 * g + 0] = ca.x + rnv[g].x; h1a[4 * g + 1] = ca.y + rnv[g].y; h1a[4 * g + 2] = ca.z + rnv[g].z; h1a[4 * g + 3] = ca.w + rnv[g].w;
        h1b[4 * g + 0] = cb.x + rnv[4 + g].x; h1b[4 * g + 1] = cb.y + rnv[4 + g].y; h1b[4 * g + 2] = cb.z + rnv[4 + g].z; h1b[4 * g + 3] = cb.w + rnv[4 + g].w;
      }
    }
    EFW_STAMP(4);
    // ---- the next tile's neighbour rows and first two centre rows (their registers are free again)
#pragma unroll
    for (int g = 0; g < 8; ++g) rnv[g] = ldg4_b(a.rn, (unsigned)nx_nz * (D_P * 4u) + 32u * g + 16u * half);
    {
      cA = __builtin_amdgcn_readfirstlane(nx_c); cB = cA; hiA = 32;
      const int prev = __shfl_up(nx_c, 1);
      const unsigned hm = (unsigned)__ballot(col > 0 && nx_c != prev && nx_c >= 0);
      if (hm) { hiA = __builtin_ctz(hm); cB = __builtin_amdgcn_readlane(nx_c, hiA); }
      rcAB = ldg4_b(a.rc, (unsigned)max((lane & 16) ? cB : cA, 0) * (D_P * 4u) + 16u * (lane & 15));
    }
    EFW_STAMP(5);
    __builtin_amdgcn_s_setprio(0);     // (the MFMA section: see the note behind layer 2)
    {                                                   // layer 1 (transposed): A = Wp^T terms from LDS, B = the lane's P row, split here
      const unsigned* w1p = sWpT + col * EFW_LD1 + half * 8;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const Bf3 pb = (EFW_X & 1) ? efw_fake3(pa[2 * j], pa[2 * j + 1]) : split3_8(pa[2 * j], pa[2 * j + 1]);
        h1a = mma6(h1a, efw_w3(w1p + j * 4, EFW_T1), pb);
        h1b = mma6(h1b, efw_w3(w1p + 32 * EFW_LD1 + j * 4, EFW_T1), pb);
      }
    }
    // ---- the records of the tile after the next one, the next tile's P rows (pinned here: the scheduler otherwise sinks the requests
    // to the end of layer 2 -- half a tile less for the P rows to arrive)
    __builtin_amdgcn_sched_barrier(0);
    {
      const unsigned e2 = (unsigned)(e0 + 64 + col);
      n2_c = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * min(e2, (unsigned)(a.n_edge - 1)));
      n2_nz = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_nz), 4u * min(e2, (unsigned)(a.n_edge + 63)));   // (padded: 64 rows of n_det)
    }
    {
      const unsigned po = (unsigned)min(e0 + 32 + col, a.n_edge - 1) * (D_E * 4u) + 16u * half;    // [E,32] fp32: < 2^31 bytes at the edge limit
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float4 v_ = ldg4_b(a.pw, po + 32u * k); pa[k] = f32x4{v_.x, v_.y, v_.z, v_.w}; }
    }
    __builtin_amdgcn_sched_barrier(0);
    EFW_STAMP(6);
    if (!(EFW_X & 8))
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1a[r] = relu_bits(h1a[r]); h1b[r] = relu_bits(h1b[r]); }
    if (KEEP) {
      // tests / debugging: relu(pw_fc1) rows, 16 bytes per (lane, feature group); rows past E land in the slack
      float* dst = a.h1_out + (size_t)(e0 + col) * D_P + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(h1a[4 * g], h1a[4 * g + 1], h1a[4 * g + 2], h1a[4 * g + 3]);
        *reinterpret_cast<float4*>(dst + 32 + 8 * g) = make_float4(h1b[4 * g], h1b[4 * g + 1], h1b[4 * g + 2], h1b[4 * g + 3]);
      }
    }
    f32x16 h2a = zero16(), h2b = zero16();
    {                                                   // layer 2: A = the h1^T registers (split per k-step), B = W2^T terms from LDS
      const unsigned* w2p = sW2T + col * EFW_LD2 + half * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x16& hs = j < 2 ? h1a : h1b;
        const int r0 = 8 * (j & 1);
        const Bf3 ha = (EFW_X & 2) ? efw_fake3(f32x4{hs[r0], hs[r0 + 1], hs[r0 + 2], hs[r0 + 3]}, f32x4{hs[r0 + 4], hs[r0 + 5], hs[r0 + 6], hs[r0 + 7]})
                                   : split3_8(f32x4{hs[r0], hs[r0 + 1], hs[r0 + 2], hs[r0 + 3]}, f32x4{hs[r0 + 4], hs[r0 + 5], hs[r0 + 6], hs[r0 + 7]});
        h2a = mma6(h2a, ha, efw_w3(w2p + j * 4, EFW_T2));
        h2b = mma6(h2b, ha, efw_w3(w2p + 32 * EFW_LD2 + j * 4, EFW_T2));
      }
    }
    // The segment bookkeeping (and the top of the next tile, up to its first MFMA) runs at a raised wave priority: the three
    // waves of a SIMD are at unrelated points of their tiles, and a wave in its vector section that gets its issue slots
    // ahead of the others' MFMA streams is back in its own MFMA section sooner (-1 %; raising the MFMA section instead: +-0).
    EFW_STAMP(7);
    __builtin_amdgcn_s_setprio(1);
    // pre-activations; relu is monotone, so max(relu(v)) = relu(max(v)): rectify once per segment.  The
    // tie count is the number of rows equal to the maximum (when the maximum is <= 0 every gradient through
    // it is zero and only count >= 1 matters).
    // (forward only: rounding is monotone, so max_r fl(h_r + b) = fl(max_r h_r + b) -- the bias is added to the segment's maximum;
    //  a training pass counts ties on the biased values, which two different raw values can share)
    if (TRAIN && !(EFW_X & 8)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { h2a[r] += bias0; h2b[r] += bias1; }
    }
    if (KEEP) {
      // tests: the exact bits the maxima, tie counts and arg-max edges below are derived from (rows past E: slack)
      float* dst = a.h2_out + (size_t)e0 * D_P + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dst[crow(r, half) * D_P] = h2a[r]; dst[crow(r, half) * D_P + 32] = h2b[r]; }
    }
    EFW_STAMP(8);
    // ---- wave-uniform segment loop
    unsigned hleft = heads;
    const bool whole = nseg == 1 && nrows == 32;       // one centre fills the tile: no row masks
    if (EFW_X & 32) { asm volatile("" :: "v"(h2a), "v"(h2b)); hleft = 0; }
    while (hleft) {
      const int lo = __builtin_ctz(hleft);
      hleft &= hleft - 1;
      const int hi = hleft ? __builtin_ctz(hleft) : nrows;
      const int cseg = __builtin_amdgcn_readlane(my_c, lo);
      float s0, s1;
      // hits of this lane's rows, one bit per accumulator slot (bit r = slot r attains the segment maximum): built
      // by b = 2 b + hit from slot 15 down (one compare + one add-with-carry per element); the tie count is its
      // population count, the arg-max slot its lowest set bit
      unsigned b0 = (EFW_X & 4) ? 1u : 0u, b1 = b0;
      if (whole) {
        s0 = h2a[0]; s1 = h2b[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { s0 = fmaxf(s0, h2a[r]); s1 = fmaxf(s1, h2b[r]); }
        s0 = half_fmax(s0);
        s1 = half_fmax(s1);
        if (TRAIN && !(EFW_X & 4)) {                    // (forward only: nobody reads the tie count or the arg-max)
#pragma unroll
          for (int r = 15; r >= 0; --r) {
            hit_bit(b0, h2a[r], s0);
            hit_bit(b1, h2b[r], s1);
          }
        }
      } else {
        // rows [lo, hi) of the tile.  Accumulator slots come in groups of four consecutive rows (8 g + 4 half + q):
        // a group of 8 rows lies outside the segment (skipped), inside it (no row masks) or across one of its
        // ends (masked) -- wave-uniform cases, so only the one or two boundary groups pay for selects.
        const unsigned rowmask = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        const unsigned mine = rowmask >> (4 * half);
        const float ninf = -__builtin_inff();
        s0 = ninf; s1 = ninf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (8 * g + 8 <= lo || 8 * g >= hi) continue;
          if (lo <= 8 * g && hi >= 8 * g + 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0 = fmaxf(s0, h2a[4 * g + q]); s1 = fmaxf(s1, h2b[4 * g + q]); }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bool in = (mine >> (8 * g + q)) & 1u;
              s0 = fmaxf(s0, in ? h2a[4 * g + q] : ninf);
              s1 = fmaxf(s1, in ? h2b[4 * g + q] : ninf);
            }
          }
        }
        s0 = half_fmax(s0);
        s1 = half_fmax(s1);
        if (TRAIN && !(EFW_X & 4))
#pragma unroll
        for (int g = 3; g >= 0; --g) {
          if (8 * g + 8 <= lo || 8 * g >= hi) { b0 <<= 4; b1 <<= 4; continue; }
          if (lo <= 8 * g && hi >= 8 * g + 8) {
#pragma unroll
            for (int q = 3; q >= 0; --q) {
              hit_bit(b0, h2a[4 * g + q], s0);
              hit_bit(b1, h2b[4 * g + q], s1);
            }
          } else {
#pragma unroll
            for (int q = 3; q >= 0; --q) {
              const bool in = (mine >> (8 * g + q)) & 1u;      // a masked row compares as -inf: never the (finite) maximum
              hit_bit(b0, in ? h2a[4 * g + q] : ninf, s0);
              hit_bit(b1, in ? h2b[4 * g + q] : ninf, s1);
            }
          }
        }
      }
      // (forward only: a count of one per half-wave -- the records' low word is then 2 and means nothing)
      unsigned q0 = TRAIN ? __popc(b0) : 1u, q1 = TRAIN ? __popc(b1) : 1u;
      int a0 = 0x7fffffff, a1 = 0x7fffffff;          // first edge attaining the maximum (no hit in this half-wave: INT_MAX)
      if (TRAIN) {
        // lowest hit slot r -> row crow(r, half) = r + (r & 12) + 4 half of the tile (v_ffbl of 0 is -1: masked by the select)
        const int r0 = (int)__builtin_ctz(b0 | 0x10000u), r1 = (int)__builtin_ctz(b1 | 0x10000u);
        const int ebase = e0 + 4 * half;
        a0 = b0 ? ebase + r0 + (r0 & 12) : a0;
        a1 = b1 ? ebase + r1 + (r1 & 12) : a1;
        a0 = half_min(a0);
        a1 = half_min(a1);
        if (EFW_X) { a0 = min(a0, a.n_edge - 1); a1 = min(a1, a.n_edge - 1); }   // (measurement builds: garbage maxima may have no hit)
      }
      if (!TRAIN) { s0 += bias0; s1 += bias1; }
      s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
      q0 = half_add(q0);
      q1 = half_add(q1);
      if (cseg == cur) {
        if (TRAIN) { g0 = s0 > m0 ? a0 : g0; g1 = s1 > m1 ? a1 : g1; }   // equal maxima keep the earlier edge
        segmax_merge(m0, k0, s0, q0);
        segmax_merge(m1, k1, s1, q1);
      } else {
        if (cur >= 0) {                                           // the previous centre is complete
          // edges are sorted by centre: only the first centre of this wave's range can have begun in
          // another wave's range; every later one that completes here lies entirely inside it
          unsigned long long* dst = a.pm + (size_t)cur * D_P + 32 * half + col;
          const float mm = half ? m1 : m0; const unsigned kk = half ? k1 : k0;
          if (!head_shared) *dst = ((unsigned long long)__float_as_uint(mm) << 32) | kk;
          else pm_flush(dst, mm, kk);
          if (TRAIN) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(mm) << 32) | (unsigned)(half ? g1 : g0);
            unsigned long long* ad = a.parg + (size_t)cur * D_P + 32 * half + col;
            if (!head_shared) *ad = key;
            else __hip_atomic_fetch_max(ad, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          head_shared = false;
        }
        cur = cseg; m0 = s0; k0 = q0; m1 = s1; k1 = q1; g0 = a0; g1 = a1;
      }
    }
    EFW_STAMP(9);
  }
  if (cur >= 0) {
    // the last centre may continue in the next wave's range
    const bool tail_shared = e_end < a.n_edge && a.edge_c[e_end] == cur;
    unsigned long long* dst = a.pm + (size_t)cur * D_P + 32 * half + col;
    const float mm = half ? m1 : m0; const unsigned kk = half ? k1 : k0;
    if (!head_shared && !tail_shared) *dst = ((unsigned long long)__float_as_uint(mm) << 32) | kk;
    else pm_flush(dst, mm, kk);
    if (TRAIN) {
      // shared centres: the larger maximum wins with its edge; for equal positive maxima (ties) any of the tied
      // edges may survive -- winners_mark finds all of them from the tie count
      const unsigned long long key = ((unsigned long long)__float_as_uint(mm) << 32) | (unsigned)(half ? g1 : g0);
      unsigned long long* ad = a.parg + (size_t)cur * D_P + 32 * half + col;
      if (!head_shared && !tail_shared) *ad = key;
      else __hip_atomic_fetch_max(ad, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  GSTAMP(a, 15);
}

// ------------------------------------------------------------------------------------------
// node_fwd: one workgroup (4 waves) per 32-detection tile.
constexpr int N_LD = D_S + 4;    // 132

struct NodeFwdArgs {
  int n_det;
  int do_post;                   // finish block b: p -> q -> y -> x_out
  int do_pre;                    // start block b+1: r, rc, rn
  int do_head;                   // after the last block: predict/fc1, fc2, logits
  int training;
  const unsigned long long* pm;  // [N,64] block b
  const float* x_prev;           // [N,128] block b input (NULL = zeros)
  const float* x0;               // [N,128] start features for the first launch (do_post == 0); NULL = zeros
  const float* w3t; const float* b3;   // fc1 transposed [64][64]
  const float* w4t; const float* b4;   // fc2 transposed [128][64]
  float* q; float* x_out;
  const float* wrt; const float* br;   // next block reduce_dim transposed [32][128]
  const float* w1t; const float* b1;   // next block pw_fc1 transposed [64][96]
  const float* wrnt; const float* brn; // next block reduce_dim_neighbor transposed [32][128] (neighbor_feats; else NULL)
  float* r; float* rc; float* rn; float* r_nb;
  // segment-max records of the NEXT block: rows of detections whose edges are split between two waves of
  // edge_fwd_w (combined atomically there) or that have no edge at all start from zero; every other row is
  // written by a plain store.
  unsigned long long* pm_next; unsigned long long* parg_next;
  const int* straddle;           // [N] see GeoArgs (NULL: the batch has no edge at all -- every record starts from zero)
  const float* hw1t; const float* hb1; const float* hw2t; const float* hb2; const float* hwl; const float* hbl;
  float* head1; float* head2; float* pred;
  // num_pwfeat_fc = 0: the 2C' one-hot x score columns of the raw pairwise features (network.py:413-419) enter the next block's
  // pw_fc1 as ONE row of its weight matrix per detection and role: rc[i] += s_i W1[class_i - 1], rn[i] += s_i W1[C' + class_i - 1],
  // and row n_det + 1 + i of rn = that neighbour term alone (i's self pair: its neighbour FEATURES are zeroed, network.py:371-374,
  // its neighbour score column is not).  raw_w1 = the next block's pw_fc1 in its natural layout [2C' + 7 + 64, 64]; NULL otherwise.
  const float* raw_w1; const float* scores; const int* classes; int cprime, multiclass; float mult;
  GNET_TRACE_FIELD
};

__global__ void __launch_bounds__(256) node_fwd(const NodeFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sX[32 * N_LD];
  __shared__ __attribute__((aligned(16))) float sY[32 * N_LD];
  __shared__ __attribute__((aligned(16))) float sR[4 * 32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * 32;
  GSTAMP(a, 0);
  // every global operand of the stages below is requested here, before the first barrier: the stages are a chain
  // of small products separated by barriers, and a load issued inside a stage costs that stage a memory latency
  BtRegs<D_P> gW3, gW4; BtRegs<32> gWr, gWrn, gW1;
  float gB3 = 0.f, gB4 = 0.f, gB1 = 0.f, gBr = 0.f, gBrn = 0.f, gXin[16];
  if (a.do_post) {
    if (wave < 2) { load_bt<D_P>(gW3, a.w3t + (size_t)(32 * wave) * D_P, D_P, lane); gB3 = a.b3[32 * wave + col]; }
    load_bt<D_P>(gW4, a.w4t + (size_t)(32 * wave) * D_P, D_P, lane);
    gB4 = a.b4[32 * wave + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int node = row0 + crow(r, half);
      gXin[r] = (a.x_prev && node < a.n_det) ? a.x_prev[(size_t)node * D_S + 32 * wave + col] : 0.f;
    }
  }
  // (the flags of the segment-max record initialisation below: requested here with everything else, not in the middle of
  //  the kernel where their round trip was exposed)
  int strad[8];
  if (a.do_pre && a.pm_next) {
#pragma unroll
    for (int j = 0; j < 8; ++j) strad[j] = a.straddle ? a.straddle[min(row0 + (tid >> 6) + 4 * j, a.n_det - 1)] : 1;
  }
  if (a.do_pre) {
    load_bt<32>(gWr, a.wrt + 32 * wave, D_S, lane);
    if (a.wrnt) load_bt<32>(gWrn, a.wrnt + 32 * wave, D_S, lane);
    load_bt<D_R>(gW1, a.w1t + (size_t)(32 * (wave & 1)) * (D_E + 2 * D_R) + D_E + (wave >> 1) * D_R, D_E + 2 * D_R, lane);
    if (wave < 2) gB1 = a.b1[32 * wave + col];
    gBr = a.br[col];
    if (a.wrnt) gBrn = a.brn[col];
  }
  if (a.do_post) {
    // p tile (segment max) -> sY[32][68]
    {
      unsigned long long pv[8];                          // all eight records requested before the first LDS store
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j, row = i >> 6, ff = i & 63;
        pv[j] = a.pm[(size_t)min(row0 + row, a.n_det - 1) * D_P + ff];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * j, row = i >> 6, ff = i & 63;
        sY[row * E_LD2 + ff] = __uint_as_float((unsigned)(pv[j] >> 32));
      }
    }
    __syncthreads();
    GSTAMP(a, 1);
    f32x16 acc = zero16();
    if (wave < 2) mma_abt_r<D_P>(acc, sY, E_LD2, gW3, lane);
    __syncthreads();
    if (wave < 2) {
      const float bb = gB3;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        const float v = fmaxf(acc[r] + bb, 0.f);
        sX[row * E_LD2 + 32 * wave + col] = v;       // q staged in sX with leading dim 68
        if (a.training && row0 + row < a.n_det) a.q[(size_t)(row0 + row) * D_P + 32 * wave + col] = v;
      }
    }
    __syncthreads();
    GSTAMP(a, 2);
    acc = zero16();
    mma_abt_r<D_P>(acc, sX, E_LD2, gW4, lane);
    __syncthreads();
    {
      const float bb = gB4;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        const int node = row0 + row;
        const float xin = gXin[r];
        const float v = fmaxf(xin + (acc[r] + bb), 0.f);   // relu(infeats + feats) network.py:408
        sX[row * N_LD + 32 * wave + col] = v;
        if (node < a.n_det) a.x_out[(size_t)node * D_S + 32 * wave + col] = v;
      }
    }
    __syncthreads();
  } else {
    if (a.x0) {                                               // start_feat from the image features (network.py:223-240)
      for (int i = tid; i < 32 * (D_S / 4); i += 256) {
        const int row = i >> 5, c4 = i & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < a.n_det) v = *reinterpret_cast<const float4*>(a.x0 + (size_t)(row0 + row) * D_S + 4 * c4);
        *reinterpret_cast<float4*>(sX + row * N_LD + 4 * c4) = v;
      }
    } else {
      for (int i = tid; i < 32 * N_LD; i += 256) sX[i] = 0.f;   // start_feat = zeros (network.py:241-246)
    }
    __syncthreads();
  }
  GSTAMP(a, 3);
  if (a.do_pre) {
    if (a.pm_next) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int node = row0 + (tid >> 6) + 4 * j;
        if (node < a.n_det && strad[j]) {
          a.pm_next[(size_t)node * D_P + (tid & 63)] = 0ull;
          if (a.parg_next) a.parg_next[(size_t)node * D_P + (tid & 63)] = 0ull;
        }
      }
    }
    // r = relu(x . Wr + br): K = 128 split over the 4 waves
    {
      f32x16 acc = zero16();
      mma_abt_r<32>(acc, sX + 32 * wave, N_LD, gWr, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[(wave * 32 + crow(r, half)) * 32 + col] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * D_R; i += 256) {
      const int row = i >> 5, ff = i & 31;
      float v = sR[(0 * 32 + row) * 32 + ff] + sR[(1 * 32 + row) * 32 + ff];
      v += sR[(2 * 32 + row) * 32 + ff];
      v += sR[(3 * 32 + row) * 32 + ff];
      v = fmaxf(v + gBr, 0.f);          // ff == col for every i of this thread
      sY[row * E_LD1 + ff] = v;
      if (a.training && row0 + row < a.n_det) a.r[(size_t)(row0 + row) * D_R + ff] = v;
    }
    __syncthreads();
    const float* sYn = sY;                       // reduced features of the NEIGHBOUR side (network.py:356-365)
    if (a.wrnt) {
      // neighbor_feats: r_n = relu(x . Wrn + brn), the same K-split product through the same partial buffer
      {
        f32x16 acc = zero16();
        mma_abt_r<32>(acc, sX + 32 * wave, N_LD, gWrn, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) sR[(wave * 32 + crow(r, half)) * 32 + col] = acc[r];
      }
      __syncthreads();
      float* sY2 = sY + 32 * E_LD1;
      for (int i = tid; i < 32 * D_R; i += 256) {
        const int row = i >> 5, ff = i & 31;
        float v = sR[(0 * 32 + row) * 32 + ff] + sR[(1 * 32 + row) * 32 + ff];
        v += sR[(2 * 32 + row) * 32 + ff];
        v += sR[(3 * 32 + row) * 32 + ff];
        v = fmaxf(v + gBrn, 0.f);
        sY2[row * E_LD1 + ff] = v;
        if (a.training && row0 + row < a.n_det) a.r_nb[(size_t)(row0 + row) * D_R + ff] = v;
      }
      __syncthreads();
      sYn = sY2;
    }
    GSTAMP(a, 4);
    // rc = r . W1[32:64] + b1 (waves 0,1) ; rn = r_n . W1[64:96] (waves 2,3; r_n = r without neighbor_feats)
    {
      const int part = wave >> 1, nt = wave & 1;
      f32x16 acc = zero16();
      mma_abt_r<D_R>(acc, part ? sYn : sY, E_LD1, gW1, lane);
      const float bb = gB1;
      float* dst = part == 0 ? a.rc : a.rn;
      if (a.raw_w1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int node = row0 + crow(r, half);
          if (node < a.n_det) {
            float sc = a.scores[node] * a.mult;                // x * 1.0f is exact
            int row = part ? 1 : 0;
            if (a.multiclass) {                                // scatter_nd one-hot x score; a class outside [1, C]: a zero column
              const int cl = a.classes[node] - 1;
              if (cl >= 0 && cl < a.cprime) row = (part ? a.cprime : 0) + cl; else { sc = 0.f; row = 0; }
            }
            const float tv = sc * a.raw_w1[(size_t)row * D_P + 32 * nt + col];
            dst[(size_t)node * D_P + 32 * nt + col] = (acc[r] + bb) + tv;
            if (part) a.rn[(size_t)(a.n_det + 1 + node) * D_P + 32 * nt + col] = tv;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int node = row0 + crow(r, half);
          if (node < a.n_det) dst[(size_t)node * D_P + 32 * nt + col] = acc[r] + bb;
        }
      }
    }
    // row n_det of rn stays zero: the edge kernels read it for self pairs (edge_nz)
    if (blockIdx.x == 0 && tid < D_P) a.rn[(size_t)a.n_det * D_P + tid] = 0.f;
  }
  GSTAMP(a, 5);
  if (a.do_head) {
    __syncthreads();
    f32x16 acc = zero16();
    mma_abt_gB<D_S, 8>(acc, sX, N_LD, a.hw1t + (size_t)(32 * wave) * D_S, D_S, lane);            // weight loads 8 k-steps ahead
    {
      const float bb = a.hb1[32 * wave + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        const float v = acc[r] + bb;                 // activation_fn=None (network.py:263)
        sY[row * N_LD + 32 * wave + col] = v;
        if (a.training && row0 + row < a.n_det) a.head1[(size_t)(row0 + row) * D_HEAD + 32 * wave + col] = v;
      }
    }
    __syncthreads();
    acc = zero16();
    mma_abt_gB<D_HEAD, 8>(acc, sY, N_LD, a.hw2t + (size_t)(32 * wave) * D_HEAD, D_HEAD, lane);
    __syncthreads();
    {
      const float bb = a.hb2[32 * wave + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        const float v = acc[r] + bb;
        sX[row * N_LD + 32 * wave + col] = v;
        if (a.training && row0 + row < a.n_det) a.head2[(size_t)(row0 + row) * D_HEAD + 32 * wave + col] = v;
      }
    }
    __syncthreads();
    {
      const int row = tid >> 3, part = tid & 7;      // 8 threads per detection
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(sX[row * N_LD + part * 16 + k], a.hwl[part * 16 + k], s);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if (part == 0 && row0 + row < a.n_det) a.pred[row0 + row] = s + a.hbl[0];
    }
  }
  GSTAMP(a, 15);
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" int gnet_forward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                            const float* params, gnet_buffers* buf, int training, gnet_stream_t stream) {
  clear_hip_error();
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !in || !params || !buf) return GNET_ERR_INVALID;
  if (shape->n_det < 0 || shape->n_edge < 0 || shape->n_img < 1) return GNET_ERR_INVALID;
  if (shape->n_det == 0) return GNET_OK;
  hipStream_t s = (hipStream_t)stream;
  const ParamLayout L = make_layout(cfg);
  const int B = cfg->num_blocks;
  const int N = shape->n_det;
  const int E = (int)shape->n_edge;
  float* pt = buf->packed_t;

  void* prof = buf->profiler;
  GNET_LAUNCH(prof, GNET_K_PACK, s, pack_transpose<<<dim3(PACK_X, (L.raw ? 0 : 3) + (5 + (cfg->neighbor_feats ? 1 : 0)) * B + 2), 256, 0, s>>>(params, pt, L.dpw, B, cfg->neighbor_feats, L.raw));
  unsigned* pwbf = reinterpret_cast<unsigned*>(pt + packed_pwbf_off(L));
  if (!L.raw)
    GNET_LAUNCH(prof, GNET_K_PACK, s, pack_pw_bf16<<<(2 * 8 * 16 * 64 * 4 + 2 * 8 * 2 * 64 * 4 + 255) / 256, 256, 0, s>>>(params + L.pw2, params + L.pw3, pwbf));

  if (E > 0) {
    // geometry columns + (row, score) pairs; kept in HBM for the backward pass when training
    GeoArgs g;
    g.n_edge = E; g.edge_c = buf->edge_c; g.edge_n = buf->edge_n; g.edge_iou = buf->edge_iou;
    g.dets = (const float4*)in->dets; g.scores = in->det_scores; g.classes = in->det_classes;
    g.cprime = L.cprime; g.multiclass = cfg->num_classes > 1;
    g.geo = buf->geo; g.edge_nz = buf->edge_nz; g.n_det = N; g.mult = cfg->pw_feat_multiplyer;
    g.w1 = params + L.pw1; g.b1 = params + L.pb1; g.tc = buf->pw_tc; g.tn = buf->pw_tn;
    g.raw = L.raw; g.pw = buf->pw_feats;
    g.row_ptr = buf->row_ptr; g.straddle = buf->scratch_i;
    g.ef_tiles = (E + 31) / 32; g.ef_waves = max(1, min(EFW_OCC * 256, ((E + 31) / 32 + EFW_WAVES - 1) / EFW_WAVES)) * EFW_WAVES;
    GNET_LAUNCH(prof, GNET_K_GEOMETRY, s, edge_geometry<<<(E + 64 + 255) / 256, 256, 0, s>>>(g));
  }
  if (E > 0 && !L.raw) {
    PwFwdArgs a;
    a.n_edge = E; a.cprime = L.cprime; a.geo = buf->geo; a.edge_c = buf->edge_c; a.edge_n = buf->edge_n;
    a.tc = buf->pw_tc; a.tn = buf->pw_tn; a.w1 = params + L.pw1;
    a.w2t = pt + L.pw2; a.b2 = params + L.pb2;
    a.w3 = params + L.pw3; a.b3 = params + L.pb3;
    a.h1 = buf->pw_h1; a.h2 = buf->pw_h2; a.pw = buf->pw_feats;
    // dynamic-LDS limits are per device and cheap to set: no process-global "done" flag
    {
      const int grid2 = min((E + PW2_T - 1) / PW2_T, 256);
      static const bool fp32_pipe = getenv("GNET_PW_FP32_PIPE") != nullptr;     // measurement only: round 5's pw_fwd2 (fp32 MFMA)
      if (fp32_pipe) {
        if (training) {
          HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_fwd2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwFwd2Smem));
          GNET_LAUNCH(prof, GNET_K_PW_FWD, s, pw_fwd2<true><<<grid2, 512, kPwFwd2Smem, s>>>(a));
        } else {
          HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_fwd2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwFwd2Smem));
          GNET_LAUNCH(prof, GNET_K_PW_FWD, s, pw_fwd2<false><<<grid2, 512, kPwFwd2Smem, s>>>(a));
        }
      } else {
        PwFwd3Args a3; a3.p = a; a3.wbf = pwbf;
        GNET_TRACE_SET(a3, "PW_FWD", true);
        if (training) {
          HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_fwd3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwFwd3Smem));
          GNET_LAUNCH(prof, GNET_K_PW_FWD, s, pw_fwd3<true><<<grid2, 512, kPwFwd3Smem, s>>>(a3));
        } else {
          HIP_CHECK_RET(hipFuncSetAttribute((const void*)pw_fwd3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwFwd3Smem));
          GNET_LAUNCH(prof, GNET_K_PW_FWD, s, pw_fwd3<false><<<grid2, 512, kPwFwd3Smem, s>>>(a3));
        }
      }
    }
  }

  const int ntile_n = (N + 31) / 32;
  // edge_fwd_w partition (2 workgroups per CU): wave-owned contiguous tile ranges
  const int ef_wg = max(1, min(EFW_OCC * 256, ((E + 31) / 32 + EFW_WAVES - 1) / EFW_WAVES));
  const bool keep_h1 = training == 2;
  if (E > 0) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_fwd_w<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeFwdWSmem));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_fwd_w<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeFwdWSmem));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_fwd_w<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeFwdWSmem));
  }

  for (int b = 0; b <= B; ++b) {
    // node stage between edge kernels: finish block b (b >= 1), start block b+1 (b < B)
    NodeFwdArgs n;
    n.n_det = N; n.do_post = b >= 1; n.do_pre = b < B; n.do_head = b == B; n.training = training;
    n.pm = b >= 1 ? (const unsigned long long*)buf->blk_pm[b] : nullptr;
    n.x_prev = b >= 2 ? buf->block_feats[b - 1] : buf->start_feat;
    n.x0 = b == 0 ? buf->start_feat : nullptr;
    if (b >= 1) {
      n.w3t = pt + L.blk[b].w3; n.b3 = params + L.blk[b].b3;
      n.w4t = pt + L.blk[b].w4; n.b4 = params + L.blk[b].b4;
      n.q = buf->blk_q[b]; n.x_out = buf->block_feats[b];
    } else { n.w3t = n.b3 = n.w4t = n.b4 = nullptr; n.q = n.x_out = nullptr; }
    if (b < B) {
      n.wrt = pt + L.blk[b + 1].wr; n.br = params + L.blk[b + 1].br;
      n.w1t = pt + packed_w1_off(L, b + 1); n.b1 = params + L.blk[b + 1].b1;
      n.r = buf->blk_r[b + 1]; n.rc = buf->blk_rc[b + 1]; n.rn = buf->blk_rn[b + 1];
      n.wrnt = cfg->neighbor_feats ? pt + L.blk[b + 1].wrn : nullptr;
      n.brn = cfg->neighbor_feats ? params + L.blk[b + 1].brn : nullptr;
      n.r_nb = buf->blk_rnb[b + 1];
    } else { n.wrt = n.br = n.w1t = n.b1 = nullptr; n.r = n.rc = n.rn = nullptr; n.wrnt = n.brn = nullptr; n.r_nb = nullptr; }
    n.pm_next = b < B ? (unsigned long long*)buf->blk_pm[b + 1] : nullptr;
    n.parg_next = (b < B && training) ? (unsigned long long*)buf->blk_parg[b + 1] : nullptr;
    n.straddle = E > 0 ? buf->scratch_i : nullptr;
    n.hw1t = pt + L.hw1; n.hb1 = params + L.hb1; n.hw2t = pt + L.hw2; n.hb2 = params + L.hb2;
    n.hwl = params + L.hwl; n.hbl = params + L.hbl;
    n.head1 = buf->head1; n.head2 = buf->head2; n.pred = buf->prediction;
    n.raw_w1 = (L.raw && b < B) ? params + L.blk[b + 1].w1 : nullptr;
    n.scores = in->det_scores; n.classes = in->det_classes; n.cprime = L.cprime; n.multiclass = cfg->num_classes > 1;
    n.mult = cfg->pw_feat_multiplyer;
    GNET_TRACE_SET(n, "NODE_FWD", b == B / 2);
    GNET_LAUNCH(prof, GNET_K_NODE_FWD, s, node_fwd<<<ntile_n, 256, 0, s>>>(n));
    if (b < B) {
      if (E > 0) {
        EdgeFwdArgs e;
        e.n_edge = E; e.edge_c = buf->edge_c; e.edge_n = buf->edge_n; e.pw = buf->pw_feats;
        e.rc = buf->blk_rc[b + 1]; e.rn = buf->blk_rn[b + 1];
        e.w1t = pt + packed_w1_off(L, b + 1); e.w2t = pt + L.blk[b + 1].w2; e.b2 = params + L.blk[b + 1].b2;
        e.pm = (unsigned long long*)buf->blk_pm[b + 1]; e.row_ptr = buf->row_ptr;
        e.edge_nz = buf->edge_nz;
        e.h1_out = keep_h1 ? buf->blk_h1[b + 1] : nullptr;
        e.h2_out = keep_h1 ? buf->blk_h2[b + 1] : nullptr;
        e.parg = training ? (unsigned long long*)buf->blk_parg[b + 1] : nullptr;
        GNET_TRACE_SET(e, "EDGE_FWD", b == B / 2);
        const int wg = ef_wg;
        if (keep_h1) {
          if (!buf->blk_h1[b + 1] || !buf->blk_h2[b + 1]) return GNET_ERR_INVALID;   // planned without training == 2
          GNET_LAUNCH(prof, GNET_K_EDGE_FWD, s, edge_fwd_w<true, true><<<wg, 64 * EFW_WAVES, kEdgeFwdWSmem, s>>>(e));
        } else if (training) { GNET_LAUNCH(prof, GNET_K_EDGE_FWD, s, edge_fwd_w<true, false><<<wg, 64 * EFW_WAVES, kEdgeFwdWSmem, s>>>(e)); }
        else { GNET_LAUNCH(prof, GNET_K_EDGE_FWD, s, edge_fwd_w<false, false><<<wg, 64 * EFW_WAVES, kEdgeFwdWSmem, s>>>(e)); }
      }
    }
  }
  return launch_status();
}
