// Edge stage of the backward pass of a _block (network.py:367-388, TF autodiff): build_context gathers,
// pw_fc1, pw_fc2 and the SegmentMax.
//
// The gradient of tf.segment_max reaches, per (detection, column), only the edge that attained the maximum
// (ties: all of them, split evenly -- _SegmentMinOrMaxGrad).  d h2 is therefore 96 % zeros even on the
// "winner" edges (64 non-zeros per detection spread over ~23 winner rows at E/N = 90), and every product with
// it is done sparsely here instead of as a dense GEMM on zeros:
//
//   winners_mark    (all blocks, one launch) arg-max record of the forward pass -> 1-bit-per-edge winner map
//                   + a per-detection "has a tied maximum" flag
//   winners_ties    exact positive ties (rare: duplicate boxes, fp32 coincidences) are resolved by recomputing
//                   the flagged detection's pw_fc1/pw_fc2 with the forward kernel's exact MFMA sequence ->
//                   per-edge extra-winner masks
//   winner_lists    bitmap -> ascending list of winner edges + per-word prefix counts (count / scan / fill,
//                   all blocks in one launch each; the OR over the blocks gives the rows of the pw-MLP backward)
//   winner_positions  list position of every (detection, column)'s arg-max edge
//   edge_bwd_w      every wave owns whole 32-winner tiles, no workgroup barriers in the tile loop:
//                     h1   = relu(P.Wp + rc[c] + rn[n])   recomputed with edge_fwd_w's sequence (24 bf16 MFMAs on split operands) -- the forward pass keeps no
//                            per-edge activations (16 x 0.37 GB of stores and 5.9 GB of workspace gone)
//                     dW2 += d_pc[c][j] * h1[arg(c,j)]    per (detection, column), lane = column: 64 FMAs
//                     g1   = (h1 > 0) * (d h2 . W2^T)     64 MFMAs, d h2 built in A-operand registers
//                     dP   = g1 . Wp^T, dWp += P^T . g1   (32 + 32 MFMAs)
//                   24 bf16 + 128 fp32 MFMAs per 32 rows (160 in the fp32 formulation, 224 for the dense-on-winner-rows one).
//   (d_rc / d_rn -- centre sums over a contiguous range of the compact g1 rows, reversed-pair sums -- are taken
//   inside the next node kernel, backward.hip blk_bwd_node)
// No float atomics, static work assignment: gradients stay bitwise reproducible.
#include "common.hpp"
#include "backward_edge.hpp"

namespace {

constexpr int LD32 = D_E + 4;    // 36
constexpr int LD64 = D_P + 4;    // 68

__device__ __forceinline__ unsigned long long low_mask(int bit) { return bit ? (~0ull >> (64 - bit)) : 0ull; }

// position of edge e in the ascending winner list = number of winner edges before it
__device__ __forceinline__ int winner_pos(const unsigned long long* __restrict__ ewin, const int* __restrict__ wprefix, int e) {
  const int w = e >> 6;
  return wprefix[w] + __popcll(ewin[w] & low_mask(e & 63));
}

// ------------------------------------------------------------------------------------------
struct WinArgs {
  int n_det;
  long long bm_stride, xm_stride, tf_stride, tl_stride;
  unsigned long long* ewin;         // [B][bm_stride] zeroed
  unsigned long long* xmask;        // [B][xm_stride] extra winners of tied columns (valid on rows of flagged detections only)
  unsigned char* tflag;             // [B][tf_stride] detection has a tied column in this block
  int* tlist;                       // [B][tl_stride] indices of the flagged detections
  int* tcount;                      // [B] their number (zeroed)
  const int* row_ptr; const int* edge_nz; const float* pw;
  const unsigned long long* pm[GNET_MAX_BLOCKS];     // [N,64] (max bits << 32) | tie count
  const unsigned long long* parg[GNET_MAX_BLOCKS];   // [N,64] (max bits << 32) | first edge attaining it
  const float* rc[GNET_MAX_BLOCKS]; const float* rn[GNET_MAX_BLOCKS];
  const float* w1t[GNET_MAX_BLOCKS]; const float* w2t[GNET_MAX_BLOCKS]; const float* b2[GNET_MAX_BLOCKS];
};

constexpr int WM_WORDS = 8;   // bitmap words a detection's edges may span on the fast path (<= 449 edges)

__global__ void __launch_bounds__(256) winners_mark(const WinArgs a) {
  __shared__ unsigned long long sbm[4][WM_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  const int blk = blockIdx.y;
  unsigned long long* ewin = a.ewin + (size_t)blk * a.bm_stride;
  unsigned char* tflag = a.tflag + (size_t)blk * a.tf_stride;
  if (lane < WM_WORDS) sbm[wave][lane] = 0ull;
  wave_lds_sync();
  for (int node = blockIdx.x * 4 + wave; node < a.n_det; node += nwaves) {
    const unsigned long long pv = a.parg[blk][(size_t)node * D_P + lane];
    const unsigned long long pc = a.pm[blk][(size_t)node * D_P + lane];
    const bool valid = (pv >> 32) != 0ull;                  // maximum > 0: the ReLU passes the gradient
    const int arg = (int)(unsigned)pv;
    const unsigned long long ties = __ballot(valid && (unsigned)pc > 1u);
    const int eb = a.row_ptr[node], ee = a.row_ptr[node + 1];
    const int w0 = eb >> 6;
    const bool fits = ((ee - 1) >> 6) - w0 < WM_WORDS;      // wave-uniform
    // winner bits: combined per bitmap word in LDS, one global atomic per word (words at the row boundaries are
    // shared with the neighbouring detections)
    if (valid) {
      const unsigned long long bit = 1ull << (arg & 63);
      if (fits) atomicOr(&sbm[wave][(arg >> 6) - w0], bit);
      else atomicOr(ewin + (arg >> 6), bit);
    }
    wave_lds_sync();
    if (fits && lane < WM_WORDS) {
      const unsigned long long m = sbm[wave][lane];
      if (m) { atomicOr(ewin + w0 + lane, m); sbm[wave][lane] = 0ull; }
    }
    wave_lds_sync();
    if (lane == 0) {
      tflag[node] = ties ? 1 : 0;
      if (ties) {                      // (order of the list is irrelevant: winners_ties only sets bits)
        a.tlist[(size_t)blk * a.tl_stride + atomicAdd(a.tcount + blk, 1)] = node;
      }
    }
  }
}

// Rare (about one detection per block: duplicate boxes, fp32 coincidences): some column's maximum is attained by
// several edges, and the forward pass recorded only the first.  Recompute pw_fc1 / pw_fc2 for the flagged
// detection's edges with the forward kernel's exact operation sequence (edge_fwd_w: accumulator initialised with
// rc + rn, operand fragments and k order of its MFMAs, so the bits match) and mark every OTHER edge that attains
// a tied maximum; d_pc already carries the 1 / count split (network.py:387-388, TF SegmentMax gradient).
// A kernel of its own: its registers must not cost winners_mark its occupancy.
__global__ void __launch_bounds__(256) winners_ties(const WinArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int nwaves = gridDim.x * 4;
  const int blk = blockIdx.y;
  unsigned long long* ewin = a.ewin + (size_t)blk * a.bm_stride;
  unsigned long long* xmask = a.xmask + (size_t)blk * a.xm_stride;
  const int* tl = a.tlist + (size_t)blk * a.tl_stride;
  const int n_tied = a.tcount[blk];
  // eight waves share a flagged detection, each taking every eighth 32-edge tile of it: the kernel is a handful of
  // waves' worth of dependent work per block, so its duration is one wave's chain
  const int gw = blockIdx.x * 4 + wave, sub = gw & 7;
  for (int li = gw >> 3; li < n_tied; li += nwaves >> 3) {
    const int node = __builtin_amdgcn_readfirstlane(tl[li]);
    const unsigned long long pv = a.parg[blk][(size_t)node * D_P + lane];
    const unsigned long long pc = a.pm[blk][(size_t)node * D_P + lane];
    const bool valid = (pv >> 32) != 0ull;
    const int arg = (int)(unsigned)pv;
    const unsigned long long ties = __ballot(valid && (unsigned)pc > 1u);
    const int eb = a.row_ptr[node], ee = a.row_ptr[node + 1];
    const float* rcp = a.rc[blk] + (size_t)node * D_P;
    const float* rnp = a.rn[blk];
    const float* w1tp = a.w1t[blk]; const float* w2tp = a.w2t[blk];
    const float bias0 = a.b2[blk][col], bias1 = a.b2[blk][32 + col];
    const float mx = __uint_as_float((unsigned)(pv >> 32));
    for (int e0 = eb + 32 * sub; e0 < ee; e0 += 32 * 8) {
      const int nrows = min(32, ee - e0);
      if (lane < nrows) xmask[e0 + lane] = 0ull;            // this tile's masks (every edge belongs to one tile of one wave)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      // pw_fc1 / pw_fc2 of the tile with edge_fwd_w's operation sequence: layer 1 transposed (lane = edge, registers = features,
      // accumulators start from rc + rn), every product as six bf16 products of the split operands in the same k-slot order
      // (common.hpp: split3_8 / mma6; the slot order is edge_fwd_w's LDS layout)
      f32x16 h1a, h1b;
      {
        const int nz = a.edge_nz[min(e0 + col, ee - 1)];
        const float* rnl = rnp + (size_t)nz * D_P + 4 * half;
        const float* rcl = rcp + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 ca = *reinterpret_cast<const float4*>(rcl + 8 * g), cb = *reinterpret_cast<const float4*>(rcl + 32 + 8 * g);
          const float4 na = *reinterpret_cast<const float4*>(rnl + 8 * g), nb = *reinterpret_cast<const float4*>(rnl + 32 + 8 * g);
          h1a[4 * g + 0] = ca.x + na.x; h1a[4 * g + 1] = ca.y + na.y; h1a[4 * g + 2] = ca.z + na.z; h1a[4 * g + 3] = ca.w + na.w;
          h1b[4 * g + 0] = cb.x + nb.x; h1b[4 * g + 1] = cb.y + nb.y; h1b[4 * g + 2] = cb.z + nb.z; h1b[4 * g + 3] = cb.w + nb.w;
        }
      }
      {
        const float* ap = a.pw + (size_t)min(e0 + col, ee - 1) * D_E + 4 * half;
        const float* wl = w1tp + (size_t)col * (D_E + 2 * D_R) + 4 * half;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const Bf3 pb = split3_8(*reinterpret_cast<const f32x4*>(ap + 16 * j), *reinterpret_cast<const f32x4*>(ap + 16 * j + 8));
          const Bf3 wa0 = split3_8(*reinterpret_cast<const f32x4*>(wl + 16 * j), *reinterpret_cast<const f32x4*>(wl + 16 * j + 8));
          const float* wl1 = wl + 32 * (D_E + 2 * D_R);
          const Bf3 wa1 = split3_8(*reinterpret_cast<const f32x4*>(wl1 + 16 * j), *reinterpret_cast<const f32x4*>(wl1 + 16 * j + 8));
          h1a = mma6(h1a, wa0, pb);
          h1b = mma6(h1b, wa1, pb);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { h1a[r] = relu_bits(h1a[r]); h1b[r] = relu_bits(h1b[r]); }
      f32x16 h2a = zero16(), h2b = zero16();
      {
        const float* vl = w2tp + (size_t)col * D_P + 4 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x16& hs = j < 2 ? h1a : h1b;
          const int r0 = 8 * (j & 1), fo = 32 * (j >> 1) + 16 * (j & 1);
          const Bf3 ha = split3_8(f32x4{hs[r0], hs[r0 + 1], hs[r0 + 2], hs[r0 + 3]}, f32x4{hs[r0 + 4], hs[r0 + 5], hs[r0 + 6], hs[r0 + 7]});
          const Bf3 wb0 = split3_8(*reinterpret_cast<const f32x4*>(vl + fo), *reinterpret_cast<const f32x4*>(vl + fo + 8));
          const Bf3 wb1 = split3_8(*reinterpret_cast<const f32x4*>(vl + 32 * D_P + fo), *reinterpret_cast<const f32x4*>(vl + 32 * D_P + fo + 8));
          h2a = mma6(h2a, ha, wb0);
          h2b = mma6(h2b, ha, wb1);
        }
      }
      unsigned long long tleft = ties;
      while (tleft) {
        const int j = __builtin_ctzll(tleft);
        tleft &= tleft - 1;
        const float mj = __shfl(mx, j);                     // lane j holds column j's maximum
        const int aj = __shfl(arg, j);                      // ... and its recorded winner (the primary)
        if (col == (j & 31)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = j < 32 ? h2a[r] + bias0 : h2b[r] + bias1;
            const int e = e0 + crow(r, half);
            if (crow(r, half) < nrows && v == mj && e != aj) {
              atomicOr(xmask + e, 1ull << j);
              atomicOr(ewin + (e >> 6), 1ull << (e & 63));
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// bitmap -> ascending list + per-word prefix counts (count -> scan -> fill: deterministic order).  blockIdx.y
// selects the bitmap (one per block, the last one = OR over the blocks).
struct ListArgs {
  int n_words, n_edge, n_wg, n_lists;
  long long bm_stride, wl_stride;
  const unsigned long long* bits;   // [n_lists][bm_stride]
  int* wg_count;                    // [n_lists][n_wg]
  int* wg_off;                      // [n_lists][n_wg + 1] (total last)
  int* rows;                        // [n_lists - 1][wl_stride] winner lists of the blocks
  int* rows_any;                    // list of the last bitmap
  int* wprefix;                     // [n_lists][bm_stride]
};

__device__ __forceinline__ unsigned long long list_word(const unsigned long long* __restrict__ bits, int w, int n_words, int n_edge) {
  if (w >= n_words) return 0ull;
  unsigned long long b = bits[w];
  if (w == n_words - 1 && (n_edge & 63)) b &= (1ull << (n_edge & 63)) - 1ull;     // bits past the last edge
  return b;
}

__global__ void __launch_bounds__(256) ewin_or(unsigned long long* __restrict__ ewin, long long bm_stride, int n_blocks, int n_words) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  unsigned long long v = 0ull;
  for (int b = 0; b < n_blocks; ++b) v |= ewin[(size_t)b * bm_stride + w];
  ewin[(size_t)n_blocks * bm_stride + w] = v;
}

__global__ void __launch_bounds__(256) list_count(const ListArgs a) {
  __shared__ int red[4];
  const unsigned long long* bits = a.bits + (size_t)blockIdx.y * a.bm_stride;
  const int w = blockIdx.x * 256 + threadIdx.x;
  int c = __popcll(list_word(bits, w, a.n_words, a.n_edge));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) a.wg_count[(size_t)blockIdx.y * a.n_wg + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the workgroup counts of every list; one workgroup per list
__global__ void __launch_bounds__(1024) list_scan(const ListArgs a) {
  __shared__ int part[1024];
  const int* cnt = a.wg_count + (size_t)blockIdx.x * a.n_wg;
  int* out = a.wg_off + (size_t)blockIdx.x * (a.n_wg + 1);
  const int n = a.n_wg, t = threadIdx.x;
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

__global__ void __launch_bounds__(256) list_fill(const ListArgs a) {
  __shared__ int part[256];
  const int y = blockIdx.y;
  const unsigned long long* bits = a.bits + (size_t)y * a.bm_stride;
  int* rows = y < a.n_lists - 1 ? a.rows + (size_t)y * a.wl_stride : a.rows_any;
  int* wprefix = a.wprefix + (size_t)y * a.bm_stride;
  const int t = threadIdx.x, w = blockIdx.x * 256 + t;
  unsigned long long b = list_word(bits, w, a.n_words, a.n_edge);
  const int c = __popcll(b);
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int pos = a.wg_off[(size_t)y * (a.n_wg + 1) + blockIdx.x] + part[t] - c;
  wprefix[w] = pos;                                   // winners before word w (w runs to n_wg * 256 <= bm_stride)
  while (b) {
    const int j = __builtin_ctzll(b);
    b &= b - 1;
    rows[pos++] = 64 * w + j;
  }
}

// ------------------------------------------------------------------------------------------
// apos[node][j] = (position in the block's winner list of the edge recorded as column j's arg-max) + 1 in bits 0-23, 0
// when the maximum is not positive (no gradient); bit 30 = the detection's tie flag (edge_bwd_w reads one record per
// (detection, column) instead of a record and a flag).  All blocks in one launch; off the backward chain.
struct PosArgs {
  int n_det;
  long long bm_stride, ap_stride;
  const unsigned long long* ewin; const int* wprefix;
  const unsigned long long* parg[GNET_MAX_BLOCKS];
  int* apos;                        // [B][ap_stride]
  const unsigned char* tflag; long long tf_stride;
};

__global__ void __launch_bounds__(256) winner_positions(const PosArgs a) {
  const int blk = blockIdx.y;
  const unsigned long long* ewin = a.ewin + (size_t)blk * a.bm_stride;
  const int* wprefix = a.wprefix + (size_t)blk * a.bm_stride;
  int* apos = a.apos + (size_t)blk * a.ap_stride;
  const long long total = (long long)a.n_det * D_P;
  // four elements per thread and step: record -> bitmap word + prefix count are dependent round trips, and one element at a
  // time left each thread waiting for them five times over (62 -> 44 us for the 16 blocks, on the side stream).  The same
  // unrolling of winner_tpos changed nothing (96 us: its random 8-byte gathers are bound by their number, not their latency).
  const long long step = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * step) {
    unsigned long long pv[4], wd[4]; int tie[4], pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long i = min(i0 + q * step, total - 1);
      pv[q] = a.parg[blk][i];
      tie[q] = a.tflag[(size_t)blk * a.tf_stride + (i >> 6)] ? (1 << 30) : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = (int)(unsigned)pv[q]; const bool on = (pv[q] >> 32) != 0ull; wd[q] = on ? ewin[e >> 6] : 0ull; pre[q] = on ? wprefix[e >> 6] : 0; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long i = i0 + q * step;
      const int e = (int)(unsigned)pv[q];
      if (i < total) apos[i] = ((pv[q] >> 32) != 0ull ? pre[q] + __popcll(wd[q] & low_mask(e & 63)) + 1 : 0) | tie[q];
    }
  }
}

// tpos[e] = list position of the reversed pair of edge e if that pair is a winner of the block (not for self pairs:
// their rn row is the zero row), wrow[i] = list position of the first winner at or after detection i's first edge.
// gather_winners reads these instead of chasing edge_t -> bitmap word -> prefix count per edge.  All blocks in one launch.
struct TposArgs {
  int n_det, n_edge, n_wg;
  const int* wlist; const int* wg_off;    // the blocks' winner lists and their lengths (list_scan's totals)
  long long bm_stride, wl_stride, tf_stride;
  const unsigned long long* ewin; const int* wprefix;
  const int* row_ptr; const int* edge_c; const int* edge_n; const int* edge_t;
  int* tpos; int* wrow;
  int* spos; long long sp_stride;         // num_pwfeat_fc = 0: list position of every detection's self pair (filled with -1); else NULL
};

__global__ void __launch_bounds__(256) winner_tpos(const TposArgs a) {
  const int blk = blockIdx.y;
  const unsigned long long* ewin = a.ewin + (size_t)blk * a.bm_stride;
  const int* wprefix = a.wprefix + (size_t)blk * a.bm_stride;
  int* tpos = a.tpos + (size_t)blk * a.wl_stride;
  int* wrow = a.wrow + (size_t)blk * a.tf_stride;
  const int stride = gridDim.x * 256;
  // SCATTER over the winners (a quarter of the edges): winner t at list position p is the reversed pair of edge edge_t[t]
  // (edge_t is an involution; a self pair is its own reverse and has no neighbour row).  tpos was filled with -1 beside the
  // forward pass.  The gather it replaces -- every edge looks its reversed pair up in the bitmap -- made 28 M random L2 requests
  // per step for the 16 blocks (96 us; the CU's address path takes one lane-address per clock); this makes 11 M.
  const int* wl = a.wlist + (size_t)blk * a.wl_stride;
  const int W = a.wg_off[(size_t)blk * (a.n_wg + 1) + a.n_wg];
  for (int p = blockIdx.x * 256 + threadIdx.x; p < W; p += stride) {
    const int t = wl[p];
    const int e = a.edge_t[t];
    if (e != t) tpos[e] = p;          // (a self pair is the involution's fixed point)
    else if (a.spos) a.spos[(size_t)blk * a.sp_stride + a.edge_c[t]] = p;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i <= a.n_det; i += stride) wrow[i] = winner_pos(ewin, wprefix, a.row_ptr[i]);
}

// ------------------------------------------------------------------------------------------
struct EdgeBwdWArgs {
  int n_edge, n_det;
  const int* wcount;                // device: number of winner rows of this block
  const int* wlist;                 // ascending winner edges
  const int* edge_c; const int* edge_nz;
  const int* apos;                  // [N,64] (list position of column j's arg-max edge) + 1, 0 = none; bit 30 = the detection's tie flag
  const unsigned long long* xmask;
  const float* pw; const float* rc; const float* rn; const float* d_pc;
  const float* w1t; const float* w2;      // pw_fc1 transposed [64][96]; pw_fc2 natural [64 (in f)][64 (out j)]
  float* d_pw;                      // [E,32] += d P on winner rows
  float* g1c;                       // [W,64] g1 of the winner rows, list order
  float* arena; long long stride;
  long long o_w1, o_w2, o_b2;
  int w1_rows;                      // pairwise rows of pw_fc1 this kernel owns: 32 (at o_w1), or the 7 geometry rows with num_pwfeat_fc = 0 (pw = the geometry columns padded to 32: the other 25 rows of P^T . g1 are exact zeros and belong to nobody)
  GNET_TRACE_FIELD
};

// Top of an edge_bwd_w tile: h1 = relu(P . Wp + (rc[c] + rn[n])) with the forward kernel's operation sequence (same bits),
// computed TRANSPOSED exactly as edge_fwd_w does (h1^T = Wp^T . P^T: lane = tile row, register r = feature 8 (r >> 2) + 4 half +
// (r & 3) [+ 32]): the accumulators start from the lane's OWN bias-row pieces x = rc[c], y = rn[n] (eight 16-byte gathers each,
// no cross-lane shuffles, no LDS staging of the bias tile) and go to the wave's LDS tile as whole 16-byte pieces of row-major
// rows (the d W2 gathers read rows; g1 reads its ReLU mask back as the same pieces).  pa = the lane's P row (B operand).
__device__ __forceinline__ void ebw_stage_h1(float* sH, int* sE, const float* sWpT, const float4 (&x)[8], const float4 (&y)[8],
                                             const f32x4 (&pa)[4], int my_e, int lane) {
  const int col = lane & 31, half = lane >> 5;
  if (half == 0) sE[col] = my_e;
  f32x16 h1a, h1b;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    h1a[4 * g + 0] = x[g].x + y[g].x; h1a[4 * g + 1] = x[g].y + y[g].y; h1a[4 * g + 2] = x[g].z + y[g].z; h1a[4 * g + 3] = x[g].w + y[g].w;
    h1b[4 * g + 0] = x[4 + g].x + y[4 + g].x; h1b[4 * g + 1] = x[4 + g].y + y[4 + g].y; h1b[4 * g + 2] = x[4 + g].z + y[4 + g].z; h1b[4 * g + 3] = x[4 + g].w + y[4 + g].w;
  }
  const float* b0 = sWpT + col * (D_E + 4) + 4 * half;
  const float* b1 = b0 + 32 * (D_E + 4);
  // edge_fwd_w's sequence: per k-step of 16 the six bf16 products of the split operands (common.hpp: mma6), k-slots 0-3 = pf 16 j + 4 half + q,
  // 4-7 = 8 further.  The weights are split here from their fp32 LDS copy (which the d P product reads as fp32): 32 values per lane and
  // tile, about what the 24 bf16 MFMAs save over 32 fp32 ones -- the point is the bits, not the time.
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const Bf3 pb = split3_8(pa[2 * j], pa[2 * j + 1]);
    const Bf3 wa0 = split3_8(*reinterpret_cast<const f32x4*>(b0 + 16 * j), *reinterpret_cast<const f32x4*>(b0 + 16 * j + 8));
    const Bf3 wa1 = split3_8(*reinterpret_cast<const f32x4*>(b1 + 16 * j), *reinterpret_cast<const f32x4*>(b1 + 16 * j + 8));
    h1a = mma6(h1a, wa0, pb);
    h1b = mma6(h1b, wa1, pb);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { h1a[r] = relu_bits(h1a[r]); h1b[r] = relu_bits(h1b[r]); }
  float* hp = sH + col * (D_P + 4) + 4 * half;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<float4*>(hp + 8 * g) = make_float4(h1a[4 * g], h1a[4 * g + 1], h1a[4 * g + 2], h1a[4 * g + 3]);
    *reinterpret_cast<float4*>(hp + 32 + 8 * g) = make_float4(h1b[4 * g], h1b[4 * g + 1], h1b[4 * g + 2], h1b[4 * g + 3]);
  }
}

constexpr int EBW_WAVES = 4;
constexpr int EBW_SLOTS = 3;                                    // detections of a tile handled between two LDS hand-offs
constexpr int EBW_WAVE_FLOATS = 32 * LD64 + 32 + EBW_SLOTS * 128;   // h1/g1 tile, edge ids [32], RJ/DV [slots][64]
// g1 = d h2 . W2^T on the bf16 pipe (round 6): W2 as three bf16 terms in the k-slot order of its MFMA operand, [term][f 64][EBW_LDW] words --
// word h * 16 + q * 4 + w = slots 2 w, 2 w + 1 of k-step q, half h: column j = 16 q + 8 (s >> 2) + 4 h + (s & 3), the order of the lane's d h2
// registers (edge_fwd_w's layout for its second layer; 36-word rows: eight lanes' 16-byte reads cover the banks once)
#ifndef EBW_G1_BF16
#define EBW_G1_BF16 1
#endif
constexpr int EBW_LDW = 36;
constexpr int EBW_W2_WORDS = EBW_G1_BF16 ? 3 * D_P * EBW_LDW : D_P * LD64;
constexpr size_t kEdgeBwdWSmem = (size_t)(D_P * LD32 + EBW_W2_WORDS + EBW_WAVES * EBW_WAVE_FLOATS) * sizeof(float);   // 78 KB (68 with the fp32 g1): two workgroups per CU

// Every wave owns whole 32-winner tiles of the block's winner list (rows sorted by centre), no workgroup barrier
// in the tile loop.  Per tile:
//   h1   = relu(P . Wp + (rc[c] + rn[n]))            24 bf16 MFMAs, the forward kernel's operation sequence (split operands, common.hpp: mma6)
//   per detection (segment of rows) of the tile, LANE = COLUMN j of the detection:
//     row_j = apos[c][j] - first list position of the tile;  v_j = d_pc[c][j]  (0 when the winner is not in the tile)
//     dW2[:, j] += v_j * h1[row_j][:]                64 FMAs on 16 gathered LDS quads, accumulators = registers
//     D[row_j][j] = v_j                              d h2 of the tile, built in MFMA A-operand registers
//   g1   = (h1 > 0) * (D . W2^T)                     64 MFMAs
//   dP   = g1 . Wp^T -> d_pw[e] +=                   32 MFMAs;  dWp += P^T . g1   32 MFMAs
// Latency discipline (vmcnt is one in-order counter for loads and stores): everything the NEXT tile reads from
// HBM/L2 is requested before this tile's stores -- the list entries two tiles ahead, the row records and P rows one
// tile ahead (right after the h1 MFMAs), the bias rows rc[c] / rn[n] and the first two detections' column records
// right after the g1 MFMAs (in flight during the 64 d Wp / d P MFMAs).  d_pw[e] += d P uses returnless float
// atomics: an edge is a row of exactly one tile per block and the blocks are separate launches, so every element
// receives its additions in launch order -- the sum stays bitwise reproducible, and no old value has to be fetched.
__global__ void __launch_bounds__(64 * EBW_WAVES, 2) edge_bwd_w(const EdgeBwdWArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sWpT = smem;                          // [64][36]  Wp^T[f][pf]: B operand of h1 (16-byte reads along pf)
  float* sW2 = sWpT + D_P * LD32;              // [64][68]  W2[f][j]:    B operand of g1 (16-byte reads along j)
  GSTAMP(a, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sH = sW2 + EBW_W2_WORDS + wave * EBW_WAVE_FLOATS;   // [32][68] bias -> h1 -> g1 of this wave's tile
  int* sE = reinterpret_cast<int*>(sH + 32 * LD64);        // [32] edge of every tile row
  int* sRJ = sE + 32;                          // [slots][64] tile row of column j's winner (-1: not in this tile)
  float* sDV = reinterpret_cast<float*>(sRJ + EBW_SLOTS * 64);   // [slots][64] d_pc[c][j]
  const int col = lane & 31, half = lane >> 5;
  float w2acc[D_P];                            // lane j: d W2[f][j], f = register index
#pragma unroll
  for (int f = 0; f < D_P; ++f) w2acc[f] = 0.f;
  f32x16 aWp0 = zero16(), aWp1 = zero16();     // d Wp[pf][f], f tiles 0 / 1
  float gb2 = 0.f;                             // lane j: d b2[j]
  const int W = *a.wcount;
  const int ntiles = (W + 31) / 32;
  const int nwaves = gridDim.x * EBW_WAVES;
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; XCD x gets the x-th contiguous eighth of the list,
  // so the rc / rn / d_pc / apos rows it gathers (one image's worth for an 8-image batch) stay in its L2
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int gw = lb * EBW_WAVES + wave;
  const int t0 = range_begin(gw, ntiles, nwaves), t1 = range_begin(gw + 1, ntiles, nwaves);     // balanced: every wave has work
  const int q4 = lane >> 4, f4 = lane & 15;    // row-layout accesses: rows 4 i + q4, 16-byte chunk f4
  // records of the next tile: edge, centre, neighbour row (lane = row, both half-waves alike), P in A layout
  int nx2_e = 0;                               // list entry two tiles ahead
  int nx_e = 0, nx_c = -1, nx_nz = 0;
  // column records (lane = column j) of the first two detections of the next tile
  int apA = 0, apB = 0, apC = 0;
  float dvA = 0.f, dvB = 0.f, dvC = 0.f;
#define EBW_LOAD_LIST(tile_) do { nx2_e = (int)ldg_b(reinterpret_cast<const unsigned*>(a.wlist), 4u * (unsigned)min((tile_) * 32 + col, W - 1)); } while (0)   /* (uniform base + 32-bit offsets throughout: no 64-bit vector address arithmetic) */
#define EBW_LOAD_ROWS(tile_)     /* uses nx2_e = list entries of tile_ */                               \
  do {                                                                                                  \
    nx_e = nx2_e;                                                                                       \
    nx_c = (tile_) * 32 + col < W ? (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_c), 4u * (unsigned)nx_e) : -1; \
    nx_nz = (int)ldg_b(reinterpret_cast<const unsigned*>(a.edge_nz), 4u * (unsigned)nx_e);              \
  } while (0)
#define EBW_LOAD_BIAS()          /* column records of the next tile's first detections; uses nx_c */                               \
  do {                                                                                                  \
    int cA_ = __builtin_amdgcn_readfirstlane(nx_c), cB_ = cA_, cC_ = cA_;                               \
    {                                                                                                   \
      const int prev_ = __shfl_up(nx_c, 1);                                                             \
      unsigned hm_ = (unsigned)__ballot(half == 0 && col > 0 && nx_c != prev_ && nx_c >= 0);            \
      if (hm_) { cB_ = __builtin_amdgcn_readlane(nx_c, __builtin_ctz(hm_)); cC_ = cB_; hm_ &= hm_ - 1; } \
      if (hm_) cC_ = __builtin_amdgcn_readlane(nx_c, __builtin_ctz(hm_));                               \
    }                                                                                                   \
    const unsigned oa_ = (unsigned)max(cA_, 0) * D_P + lane, ob_ = (unsigned)max(cB_, 0) * D_P + lane;  \
    const unsigned oc_ = (unsigned)max(cC_, 0) * D_P + lane;                                            \
    apA = (int)ldg_b(reinterpret_cast<const unsigned*>(a.apos), 4u * oa_); dvA = ldg_b(a.d_pc, 4u * oa_); \
    apB = (int)ldg_b(reinterpret_cast<const unsigned*>(a.apos), 4u * ob_); dvB = ldg_b(a.d_pc, 4u * ob_); \
    apC = (int)ldg_b(reinterpret_cast<const unsigned*>(a.apos), 4u * oc_); dvC = ldg_b(a.d_pc, 4u * oc_); \
  } while (0)
  // bias rows rc[c], rn[n] as the lane's own sixteen 16-byte pieces (lane = row, features 8 g + 4 half .. + 3 and + 32) and its
  // P row (B layout) of the tile whose records are in nx_e / nx_c / nx_nz: all twenty requests in flight together
#ifndef EBW_X
#define EBW_X 0   /* measurement builds (WRONG results, durations only): 1 = every tile gathers the rows of the wave's FIRST tile (cache hits: what a perfect prefetch of the bias / P rows would buy) */
#endif
#define EBW_LOAD_TILE(bx, by, bpa)                                                                      \
  do {                                                                                                  \
    if (EBW_X & 1) { nx_e = first_e; nx_c = first_c; nx_nz = first_nz; }                                \
    const unsigned po_ = (unsigned)nx_e * (D_E * 4u) + 16u * half;                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { const float4 v_ = ldg4_b(a.pw, po_ + 32u * k_); bpa[k_] = f32x4{v_.x, v_.y, v_.z, v_.w}; } \
    const unsigned oc_ = (unsigned)max(nx_c, 0) * (D_P * 4u) + 16u * half, on_ = (unsigned)nx_nz * (D_P * 4u) + 16u * half; \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                  \
      bx[g_] = ldg4_b(a.rc, oc_ + 32u * g_);                                                            \
      by[g_] = ldg4_b(a.rn, on_ + 32u * g_);                                                            \
    }                                                                                                   \
  } while (0)
  // Kernel front: the first tile's chain of dependent requests -- list entries -> row records -> (bias rows, P rows,
  // column records): three round trips -- runs BESIDE the staging of the weights (16-byte loads into registers, then
  // LDS), not after it, and the first tile's h1 is formed right behind the barrier: with one tile per wave or fewer (a
  // single image) the front used to be 14 of the kernel's 36 us.  Later tiles form theirs at the top of the loop body.
  const bool have_tiles = t0 < t1;
  if (have_tiles) EBW_LOAD_LIST(t0);
  if (have_tiles) {
    EBW_LOAD_ROWS(t0);
    if (t0 + 1 < t1) EBW_LOAD_LIST(t0 + 1);
  }
  f32x4 wst[6];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int i = tid + 256 * j; wst[j] = *reinterpret_cast<const f32x4*>(a.w1t + (i >> 3) * (D_E + 2 * D_R) + 4 * (i & 7)); }   // W1[pf][f] as [f][pf]
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; wst[2 + j] = *reinterpret_cast<const f32x4*>(a.w2 + 4 * i); }
  __builtin_amdgcn_sched_barrier(0);
  float4 fx[8], fy[8];
  f32x4 fpa[4];
  const int first_e = nx_e;
  const int first_c = nx_c, first_nz = nx_nz;      // (measurement builds only: EBW_X)
  if (have_tiles) { EBW_LOAD_TILE(fx, fy, fpa); EBW_LOAD_BIAS(); }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int i = tid + 256 * j; *reinterpret_cast<f32x4*>(sWpT + (i >> 3) * LD32 + 4 * (i & 7)) = wst[j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j;
    if (EBW_G1_BF16) {
      // W2[f][4 g .. 4 g + 3] -> slots 4 ((g & 3) >> 1) .. + 3 of (half g & 1, k-step g >> 2): two words per term
      const int f = i >> 4, g = i & 15;
      unsigned ph0, pm0, pl0, ph1, pm1, pl1;
      split3_pk(wst[2 + j].x, wst[2 + j].y, ph0, pm0, pl0);
      split3_pk(wst[2 + j].z, wst[2 + j].w, ph1, pm1, pl1);
      unsigned* dst = reinterpret_cast<unsigned*>(sW2) + f * EBW_LDW + (g & 1) * 16 + (g >> 2) * 4 + 2 * ((g & 3) >> 1);
      *reinterpret_cast<uint2*>(dst) = make_uint2(ph0, ph1);
      *reinterpret_cast<uint2*>(dst + D_P * EBW_LDW) = make_uint2(pm0, pm1);
      *reinterpret_cast<uint2*>(dst + 2 * D_P * EBW_LDW) = make_uint2(pl0, pl1);
    } else {
      *reinterpret_cast<f32x4*>(sW2 + (i >> 4) * LD64 + 4 * (i & 15)) = wst[2 + j];
    }
  }
  __syncthreads();
  GSTAMP(a, 1);
  if (have_tiles) ebw_stage_h1(sH, sE, sWpT, fx, fy, fpa, first_e, lane);
  GSTAMP(a, 2);
  const bool young = blockIdx.x >= gridDim.x / 2;      // the second workgroup dispatched to its CU (wave priorities below)
  for (int t = t0; t < t1; ++t) {
    const int p0 = t * 32;
    const int nrows = min(32, W - p0);
    const int my_e = nx_e, my_c = nx_c;
    const int tapA = apA, tapB = apB, tapC = apC;
    const float tdvA = dvA, tdvB = dvB, tdvC = dvC;
    // ---- stage: edge ids, bias tile, h1 (the first tile's were formed in the kernel front)
    if (t != t0) {
      float4 x[8], y[8];
      f32x4 pa[4];
      EBW_LOAD_TILE(x, y, pa);
      ebw_stage_h1(sH, sE, sWpT, x, y, pa, my_e, lane);
    }
    // the next tile's row records and P rows (its list entries arrived a tile ago); the list entries after those
    if (t + 1 < t1) EBW_LOAD_ROWS(t + 1);
    if (t + 2 < t1) EBW_LOAD_LIST(t + 2);
    wave_lds_sync();
    if (t == t0) GSTAMP(a, 3);
    // ---- segments of the tile (rows sorted by centre): wave-uniform loop
    unsigned heads;
    {
      const int prev = __shfl_up(my_c, 1);
      heads = (unsigned)__ballot(half == 0 && col < nrows && (col == 0 || my_c != prev));
    }
    float dA[32];                                        // d h2[row = col][j = 4 half + 8 s + t] at index 4 s + t
#pragma unroll
    for (int i = 0; i < 32; ++i) dA[i] = 0.f;
    unsigned hleft = heads;
    int seg0 = 0;                                        // index of the chunk's first detection within the tile
    // segment index of this lane's row (lane = row): number of heads at or before it, minus one
    const int kseg = __popc(heads & (0xffffffffu >> (31 - col))) - 1;
    while (hleft) {
      // EBW_SLOTS detections per chunk (one chunk for nearly every tile), branch-free: a missing detection is an
      // all-inactive slot.  Pass 1 = column records -> slots + the d W2 gathers (all requested back to back),
      // pass 2 = d h2 rows from the slots.
      int lo[EBW_SLOTS], hi[EBW_SLOTS], tfk[EBW_SLOTS], growk[EBW_SLOTS];
      float dvk[EBW_SLOTS], dvm[EBW_SLOTS];
      int tf_any = 0;
      const int nk = min(EBW_SLOTS, __popc(hleft));      // detections in this chunk (wave-uniform)
#pragma unroll
      for (int k = 0; k < EBW_SLOTS; ++k) {
        const bool have = hleft != 0u;                   // wave-uniform
        lo[k] = have ? __builtin_ctz(hleft) : 0;
        hleft &= hleft - 1;
        hi[k] = have ? (hleft ? __builtin_ctz(hleft) : nrows) : 0;
        // lane j: column j of this detection (the first detections of a tile were requested a tile ago)
        int rec; float dv;
        if (seg0 == 0) { rec = k == 0 ? tapA : k == 1 ? tapB : tapC; dv = k == 0 ? tdvA : k == 1 ? tdvB : tdvC; }
        else {
          const int cseg = max(__builtin_amdgcn_readlane(my_c, lo[k]), 0);
          rec = a.apos[(size_t)cseg * D_P + lane]; dv = a.d_pc[(size_t)cseg * D_P + lane];
        }
        const int ap = (rec & 0xffffff) - 1;                                   // list position, -1 = no gradient
        const int tf = __builtin_amdgcn_readfirstlane(rec) >> 30;              // the detection's tie flag (same in all 64 records)
        const int rowj = ap - p0;
        const bool inj = have && ap >= 0 && rowj >= lo[k] && rowj < hi[k];     // the column's winner is a row of this tile
        sRJ[k * 64 + lane] = inj ? rowj : -1;
        sDV[k * 64 + lane] = dv;
        growk[k] = inj ? rowj : 0; dvm[k] = inj ? dv : 0.f; dvk[k] = dv;
        tfk[k] = have ? tf : 0; tf_any |= tfk[k];
        gb2 += dvm[k];
      }
      // d W2[:, j] += d_pc[c][j] * h1[row of the column's recorded winner].  One copy of the gather (not unrolled
      // over the slots: the register allocator cannot keep three copies' loads apart); inactive lanes add 0 * h1[0]
      static_assert(EBW_SLOTS == 3, "slot selects below");
#pragma unroll 1
      for (int k = 0; k < nk; ++k) {
        const int gr = k == 0 ? growk[0] : k == 1 ? growk[1] : growk[2];
        const float dm = k == 0 ? dvm[0] : k == 1 ? dvm[1] : dvm[2];
        const float* hr = sH + gr * LD64;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * q);
          w2acc[4 * q + 0] = fmaf(dm, hv.x, w2acc[4 * q + 0]);
          w2acc[4 * q + 1] = fmaf(dm, hv.y, w2acc[4 * q + 1]);
          w2acc[4 * q + 2] = fmaf(dm, hv.z, w2acc[4 * q + 2]);
          w2acc[4 * q + 3] = fmaf(dm, hv.w, w2acc[4 * q + 3]);
          if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight quads in flight (registers)
        }
      }
      wave_lds_sync();
      // pass 2: d h2 rows, A-operand layout (lane = row): the row reads the slot of its own detection
      {
        const int ks = kseg - seg0;
        const bool mine = col < nrows && ks >= 0 && ks < EBW_SLOTS;
        const int slot = mine ? ks : 0;
        const int mrow = mine ? col : -2;                // (slot entries are tile rows or -1)
        const int* rjp = sRJ + slot * 64 + 4 * half;
        const float* dvp = sDV + slot * 64 + 4 * half;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int4 rj = *reinterpret_cast<const int4*>(rjp + 8 * s8);
          const float4 vv = *reinterpret_cast<const float4*>(dvp + 8 * s8);
          dA[4 * s8 + 0] = rj.x == mrow ? vv.x : dA[4 * s8 + 0];
          dA[4 * s8 + 1] = rj.y == mrow ? vv.y : dA[4 * s8 + 1];
          dA[4 * s8 + 2] = rj.z == mrow ? vv.z : dA[4 * s8 + 2];
          dA[4 * s8 + 3] = rj.w == mrow ? vv.w : dA[4 * s8 + 3];
        }
      }
      if (tf_any) {
        // tied maxima (rare): rows that are EXTRA winners of some columns receive the same d_pc[c][j] (already
        // divided by the count).  One (not unrolled) copy of the code: slot values picked by selects.
        static_assert(EBW_SLOTS == 3, "slot selects below");
#pragma unroll 1
        for (int k = 0; k < EBW_SLOTS; ++k) {
          const int tfs = k == 0 ? tfk[0] : k == 1 ? tfk[1] : tfk[2];
          if (!tfs) continue;
          const int los = k == 0 ? lo[0] : k == 1 ? lo[1] : lo[2];
          const int his = k == 0 ? hi[0] : k == 1 ? hi[1] : hi[2];
          const float dvs = k == 0 ? dvk[0] : k == 1 ? dvk[1] : dvk[2];
          const bool mine = col >= los && col < his;
          const unsigned long long xs = mine ? a.xmask[my_e] : 0ull;     // lane = row: its extra columns
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            const float4 vv = *reinterpret_cast<const float4*>(sDV + k * 64 + 4 * half + 8 * s8);
            const unsigned bits = (unsigned)(xs >> (4 * half + 8 * s8)) & 15u;
            dA[4 * s8 + 0] = (bits & 1u) ? vv.x : dA[4 * s8 + 0];
            dA[4 * s8 + 1] = (bits & 2u) ? vv.y : dA[4 * s8 + 1];
            dA[4 * s8 + 2] = (bits & 4u) ? vv.z : dA[4 * s8 + 2];
            dA[4 * s8 + 3] = (bits & 8u) ? vv.w : dA[4 * s8 + 3];
          }
#pragma unroll 1
          for (int r = los; r < his; ++r) {
            const unsigned long long xr = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(xs >> 32), r) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)xs, r);
            if (xr == 0ull) continue;
            const float dx = ((xr >> lane) & 1ull) ? dvs : 0.f;      // lane j: column j has an extra winner in row r
            gb2 += dx;
            const float* hr = sH + r * LD64;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * q);
              w2acc[4 * q + 0] = fmaf(dx, hv.x, w2acc[4 * q + 0]);
              w2acc[4 * q + 1] = fmaf(dx, hv.y, w2acc[4 * q + 1]);
              w2acc[4 * q + 2] = fmaf(dx, hv.z, w2acc[4 * q + 2]);
              w2acc[4 * q + 3] = fmaf(dx, hv.w, w2acc[4 * q + 3]);
              if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      seg0 += EBW_SLOTS;
      if (hleft) wave_lds_sync();                        // the slots are rewritten by the next chunk
    }
    if (t == t0) GSTAMP(a, 4);
    // ---- g1 = (h1 > 0) * (d h2 . W2^T), computed transposed (g1^T = W2 . d h2^T: the operands swapped) so that its
    // accumulators share h1^T's layout -- lane = row, register r = feature crow(r, half) [+ 32]: the mask is the lane's own
    // eight 16-byte pieces of its h1 row (requested in front of the MFMAs), the masked values ARE the A operand of the d P
    // product below (k pairing of mma_abt), and the rows go to LDS as 16-byte pieces (over h1) for the d Wp product and
    // the row store
    f32x16 g1a = zero16(), g1b = zero16();
    if (young) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);     // (the MFMA section: see the note at the atomics below)
    {
      float* hp = sH + col * LD64 + 4 * half;
      float4 hm[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) { hm[g] = *reinterpret_cast<const float4*>(hp + 8 * g); hm[4 + g] = *reinterpret_cast<const float4*>(hp + 32 + 8 * g); }
      if (EBW_G1_BF16) {
        // six bf16 products of exact three-term splits per fp32 product (common.hpp: mma6): 48 MFMAs of 32 cycles for the 64 of 64 cycles
        // below; the lane's d h2 registers are split here (176 vector instructions on the wave's own path, lesson 76: the balance is
        // +1700 pipe cycles per tile)
        const unsigned* a0 = reinterpret_cast<const unsigned*>(sW2) + col * EBW_LDW + half * 16;
        const unsigned* a1 = a0 + 32 * EBW_LDW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const Bf3 db = split3_8(f32x4{dA[8 * q], dA[8 * q + 1], dA[8 * q + 2], dA[8 * q + 3]}, f32x4{dA[8 * q + 4], dA[8 * q + 5], dA[8 * q + 6], dA[8 * q + 7]});
          Bf3 w0, w1;
          w0.h = *reinterpret_cast<const u32x4*>(a0 + 4 * q); w0.m = *reinterpret_cast<const u32x4*>(a0 + 4 * q + D_P * EBW_LDW); w0.l = *reinterpret_cast<const u32x4*>(a0 + 4 * q + 2 * D_P * EBW_LDW);
          w1.h = *reinterpret_cast<const u32x4*>(a1 + 4 * q); w1.m = *reinterpret_cast<const u32x4*>(a1 + 4 * q + D_P * EBW_LDW); w1.l = *reinterpret_cast<const u32x4*>(a1 + 4 * q + 2 * D_P * EBW_LDW);
          g1a = mma6(g1a, w0, db);
          g1b = mma6(g1b, w1, db);
        }
      }
      const float* b0 = sW2 + col * LD64 + 4 * half;
      const float* b1 = b0 + 32 * LD64;
#pragma unroll
      for (int s8 = 0; s8 < (EBW_G1_BF16 ? 0 : 8); ++s8) {
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(b0 + 8 * s8);
        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(b1 + 8 * s8);
        g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0.x, dA[4 * s8 + 0], g1a, 0, 0, 0);
        g1b = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1.x, dA[4 * s8 + 0], g1b, 0, 0, 0);
        g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0.y, dA[4 * s8 + 1], g1a, 0, 0, 0);
        g1b = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1.y, dA[4 * s8 + 1], g1b, 0, 0, 0);
        g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0.z, dA[4 * s8 + 2], g1a, 0, 0, 0);
        g1b = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1.z, dA[4 * s8 + 2], g1b, 0, 0, 0);
        g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0.w, dA[4 * s8 + 3], g1a, 0, 0, 0);
        g1b = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1.w, dA[4 * s8 + 3], g1b, 0, 0, 0);
      }
      const bool live = col < nrows;                     // rows past the list (last tile): zero
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        g1a[4 * g + 0] = (live && hm[g].x > 0.f) ? g1a[4 * g + 0] : 0.f; g1a[4 * g + 1] = (live && hm[g].y > 0.f) ? g1a[4 * g + 1] : 0.f;
        g1a[4 * g + 2] = (live && hm[g].z > 0.f) ? g1a[4 * g + 2] : 0.f; g1a[4 * g + 3] = (live && hm[g].w > 0.f) ? g1a[4 * g + 3] : 0.f;
        g1b[4 * g + 0] = (live && hm[4 + g].x > 0.f) ? g1b[4 * g + 0] : 0.f; g1b[4 * g + 1] = (live && hm[4 + g].y > 0.f) ? g1b[4 * g + 1] : 0.f;
        g1b[4 * g + 2] = (live && hm[4 + g].z > 0.f) ? g1b[4 * g + 2] : 0.f; g1b[4 * g + 3] = (live && hm[4 + g].w > 0.f) ? g1b[4 * g + 3] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(hp + 8 * g) = make_float4(g1a[4 * g], g1a[4 * g + 1], g1a[4 * g + 2], g1a[4 * g + 3]);
        *reinterpret_cast<float4*>(hp + 32 + 8 * g) = make_float4(g1b[4 * g], g1b[4 * g + 1], g1b[4 * g + 2], g1b[4 * g + 3]);
      }
    }
    wave_lds_sync();
    // the next tile's bias rows and first column records: requested BEFORE this tile's stores, in flight during the
    // d Wp / d P MFMAs
    if (t + 1 < t1) EBW_LOAD_BIAS();
    if (t == t0) GSTAMP(a, 5);
    // X = P^T for the d Wp product below: P[row][pf = col] gathered again (the rows were read for the h1 MFMAs: L1 / L2
    // hits), requested here so that the round trip runs under the d P MFMAs
    float xs[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) xs[kk] = ldg_b(a.pw, (unsigned)sE[2 * kk + half] * (D_E * 4u) + 4u * col);
    if (t == t0) GSTAMP(a, 6);
    // ---- d P = g1 . Wp^T;  d_pw[e] += d P
    {
      f32x16 acc = zero16();
      const float* bp = sWpT + (4 * half) * LD32 + col;             // B[k = f][n = pf] = Wp[pf = col][f] = sWpT[f][pf]
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                  // k = f = 8 g + 4 half + q: the g1^T registers, in k order
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1a[4 * g + 0], bp[(8 * g + 0) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1a[4 * g + 1], bp[(8 * g + 1) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1a[4 * g + 2], bp[(8 * g + 2) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1a[4 * g + 3], bp[(8 * g + 3) * LD32], acc, 0, 0, 0);
        if ((g & 1) != 0) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1b[4 * g + 0], bp[(32 + 8 * g + 0) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1b[4 * g + 1], bp[(32 + 8 * g + 1) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1b[4 * g + 2], bp[(32 + 8 * g + 2) * LD32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1b[4 * g + 3], bp[(32 + 8 * g + 3) * LD32], acc, 0, 0, 0);
        if ((g & 1) != 0) __builtin_amdgcn_sched_barrier(0);
      }
      // Everything outside the g1 / d P MFMA section runs at a raised wave priority: a wave in its bookkeeping (atomics, row
      // stores, the next tile's slot logic and gathers) gets its issue slots ahead of the other workgroup's MFMA stream and is
      // back in its own MFMA section sooner (-1 %; raising the MFMA section instead: +-0).  The CU's second-dispatched workgroup
      // (the loser of every age-based arbitration) runs one level above the first in both sections: another -2.7 %.
      if (young) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
      // rows past the list (last tile) go to the slack row E of d_pw: unconditional, no divergent branches
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // (edge of accumulator row r through two v_readlane of the lane = row records: an LDS read per atomic would
        // put an LDS latency in front of each of the 16)
        const unsigned er = crow(r, half) < nrows ? (unsigned)row_bcast(my_e, r, half) : (unsigned)a.n_edge;
        __hip_atomic_fetch_add(reinterpret_cast<float*>(reinterpret_cast<char*>(a.d_pw) + (er * (D_E * 4u) + 4u * col)), acc[r],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (t == t0) GSTAMP(a, 7);
    // ---- d Wp += P^T . g1 (issued behind the atomics: its operands were requested in front of them, so the in-order
    // memory counter does not wait for the atomics' round trip)
    {
      const float* Y = sH + col;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int row = 2 * kk + half;
        aWp0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[kk], Y[row * LD64], aWp0, 0, 0, 0);
        aWp1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[kk], Y[row * LD64 + 32], aWp1, 0, 0, 0);
      }
    }
    // ---- g1 rows -> compact list order (whole 256-byte rows; the buffer has slack rows past W)
    {
      // (uniform base + 32-bit byte offset: a per-lane 64-bit pointer kept across the tile loop was spilled, and its reload
      // sat behind the sixteen atomics above -- the in-order memory counter made every tile wait for their round trip)
      const unsigned off0 = (unsigned)(p0 + q4) * (D_P * 4u) + 16u * f4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + q4;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.g1c) + (off0 + (unsigned)i * (4u * D_P * 4u))) =
            *reinterpret_cast<const float4*>(sH + row * LD64 + 4 * f4);
      }
    }
    wave_lds_sync();
    GSTAMP(a, (10 + (t - t0)) < 14 ? 10 + (t - t0) : 13);
  }
#undef EBW_LOAD_LIST
#undef EBW_LOAD_ROWS
#undef EBW_LOAD_BIAS
#undef EBW_LOAD_TILE
  // ---- partial weight gradients of this workgroup: the four waves' accumulators are added in wave order
  __syncthreads();
  GSTAMP(a, 14);
  // the whole LDS allocation is free now (17 408 floats): two copies of the gradient block [W2 4096 | Wp 2048 | b2 64].
  // Waves 0 / 1 store their accumulators into copy 0 / 1, waves 2 / 3 add theirs on top (every lane touches its own
  // elements only), then all threads fold the two copies on the way to the arena: (w0 + w2) + (w1 + w3), a fixed order.
  constexpr int RED = D_P * D_P + 2 * 16 * 64 + 64;
  static_assert(2 * RED <= D_P * LD32 + D_P * LD64 + EBW_WAVES * EBW_WAVE_FLOATS, "two reduction copies fit");
  float* red = smem + (wave & 1) * RED;
  for (int ph = 0; ph < 2; ++ph) {
    if ((wave >> 1) == ph) {
#pragma unroll
      for (int f = 0; f < D_P; ++f) red[f * D_P + lane] = (ph ? red[f * D_P + lane] : 0.f) + w2acc[f];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[D_P * D_P + r * 64 + lane] = (ph ? red[D_P * D_P + r * 64 + lane] : 0.f) + aWp0[r];
        red[D_P * D_P + (16 + r) * 64 + lane] = (ph ? red[D_P * D_P + (16 + r) * 64 + lane] : 0.f) + aWp1[r];
      }
      red[D_P * D_P + 2048 + lane] = (ph ? red[D_P * D_P + 2048 + lane] : 0.f) + gb2;
    }
    __syncthreads();
  }
  const float* r0 = smem; const float* r1 = smem + RED;
  float* ar = a.arena + (size_t)blockIdx.x * a.stride;
  for (int i = tid; i < D_P * D_P / 4; i += 64 * EBW_WAVES) {                 // d W2 [f][j], 16-byte stores
    const float4 u = *reinterpret_cast<const float4*>(r0 + 4 * i), v = *reinterpret_cast<const float4*>(r1 + 4 * i);
    *reinterpret_cast<float4*>(ar + a.o_w2 + 4 * i) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
  if (wave < 2) {                              // rows 0-31 of pw_fc1 (pairwise features), f tile = wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = D_P * D_P + (16 * wave + r) * 64 + lane;
      if (crow(r, half) < a.w1_rows) ar[a.o_w1 + (size_t)crow(r, half) * D_P + 32 * wave + col] = r0[o] + r1[o];
    }
  }
  if (tid < D_P) ar[a.o_b2 + tid] = r0[D_P * D_P + 2048 + tid] + r1[D_P * D_P + 2048 + tid];
  GSTAMP(a, 15);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// the zeroing the winner maps need: independent of the forward pass (may run long before edge_stage_prepare)
int edge_stage_clear(const gnet_config* cfg, const gnet_shape* shape, gnet_buffers* buf, hipStream_t s) {
  const int B = cfg->num_blocks, N = shape->n_det, E = (int)shape->n_edge;
  const EdgeGeom G = edge_geom(E, N);
  HIP_CHECK_RET(hipMemsetAsync(buf->ewin, 0, (size_t)(B + 1) * G.bm_stride * sizeof(unsigned long long), s));
  HIP_CHECK_RET(hipMemsetAsync(buf->rl_scratch + (size_t)(B + 1) * (2 * G.n_wg + 1), 0, (size_t)GNET_MAX_BLOCKS * sizeof(int), s));
  // tpos = -1 everywhere: winner_tpos scatters the positions of the winners' reversed pairs over it
  if (E > 0) HIP_CHECK_RET(hipMemsetAsync(buf->tpos, 0xff, (size_t)B * G.wl_stride * sizeof(int), s));
  if (E > 0 && buf->spos) HIP_CHECK_RET(hipMemsetAsync(buf->spos, 0xff, (size_t)B * G.tf_stride * sizeof(int), s));
  return GNET_OK;
}

// part 0 = everything; 1 = what the edge kernels read (maps, lists, arg-max positions); 2 = what only gather_winners reads (the
// reversed pairs' list positions): the second part is off the chain in front of the first edge_bwd_w (80 of 275 us at the bench batch)
int edge_stage_prepare(const gnet_config* cfg, const gnet_shape* shape, const ParamLayout& L, const float* params,
                       gnet_buffers* buf, hipStream_t s, int part) {
  const int B = cfg->num_blocks, N = shape->n_det, E = (int)shape->n_edge;
  const EdgeGeom G = edge_geom(E, N);
  void* prof = buf->profiler;
  const float* pt = buf->packed_t;
  WinArgs w;
  w.n_det = N; w.bm_stride = (long long)G.bm_stride; w.xm_stride = (long long)G.xm_stride; w.tf_stride = (long long)G.tf_stride;
  w.ewin = (unsigned long long*)buf->ewin; w.xmask = (unsigned long long*)buf->xmask; w.tflag = (unsigned char*)buf->tflag;
  // list of the detections with tied maxima: the arg-max position array apos is written after winners_ties, its
  // first N ints per block serve as the list until then; the counters live in the slack of the scan scratch
  w.tlist = buf->apos; w.tl_stride = (long long)G.ap_stride;
  w.tcount = buf->rl_scratch + (size_t)(B + 1) * (2 * G.n_wg + 1);     // (zeroed by edge_stage_clear)
  w.row_ptr = buf->row_ptr; w.edge_nz = buf->edge_nz; w.pw = buf->pw_feats;
  for (int b = 1; b <= B; ++b) {
    w.pm[b - 1] = (const unsigned long long*)buf->blk_pm[b]; w.parg[b - 1] = (const unsigned long long*)buf->blk_parg[b];
    w.rc[b - 1] = buf->blk_rc[b]; w.rn[b - 1] = buf->blk_rn[b];
    w.w1t[b - 1] = pt + packed_w1_off(L, b); w.w2t[b - 1] = pt + L.blk[b].w2; w.b2[b - 1] = params + L.blk[b].b2;
  }
  if (part != 2) {
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, winners_mark<<<dim3(min((N + 3) / 4, 1024), B), 256, 0, s>>>(w));
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, winners_ties<<<dim3(16, B), 256, 0, s>>>(w));
  }
  ListArgs l;
  l.n_words = (int)G.n_words; l.n_edge = E; l.n_wg = (int)G.n_wg; l.n_lists = B + 1;
  l.bm_stride = (long long)G.bm_stride; l.wl_stride = (long long)G.wl_stride;
  l.bits = (const unsigned long long*)buf->ewin;
  l.wg_count = buf->rl_scratch; l.wg_off = buf->rl_scratch + (size_t)(B + 1) * G.n_wg;
  l.rows = buf->wlist; l.rows_any = buf->pw_rows; l.wprefix = buf->wprefix;
  if (part != 2) {
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, ewin_or<<<(int)((G.n_words + 255) / 256), 256, 0, s>>>((unsigned long long*)buf->ewin, (long long)G.bm_stride, B, (int)G.n_words));
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, list_count<<<dim3((unsigned)G.n_wg, B + 1), 256, 0, s>>>(l));
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, list_scan<<<B + 1, 1024, 0, s>>>(l));
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, list_fill<<<dim3((unsigned)G.n_wg, B + 1), 256, 0, s>>>(l));
  PosArgs p;
  p.n_det = N; p.bm_stride = (long long)G.bm_stride; p.ap_stride = (long long)G.ap_stride;
  p.ewin = (const unsigned long long*)buf->ewin; p.wprefix = buf->wprefix; p.apos = buf->apos;
  p.tflag = (const unsigned char*)buf->tflag; p.tf_stride = (long long)G.tf_stride;
  for (int b = 1; b <= B; ++b) p.parg[b - 1] = (const unsigned long long*)buf->blk_parg[b];
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, winner_positions<<<dim3(min((N * D_P + 255) / 256, 1024), B), 256, 0, s>>>(p));
  }
  if (part == 1) return GNET_OK;
  TposArgs t;
  t.n_det = N; t.n_edge = E; t.bm_stride = (long long)G.bm_stride; t.wl_stride = (long long)G.wl_stride; t.tf_stride = (long long)G.tf_stride;
  t.ewin = (const unsigned long long*)buf->ewin; t.wprefix = buf->wprefix;
  t.row_ptr = buf->row_ptr; t.edge_c = buf->edge_c; t.edge_n = buf->edge_n; t.edge_t = buf->edge_t;   // (edge_t: gnet_graph_transpose, earlier on this stream)
  t.tpos = buf->tpos; t.wrow = buf->wrow; t.n_wg = (int)G.n_wg; t.wlist = buf->wlist; t.wg_off = l.wg_off;
  t.spos = L.raw ? buf->spos : nullptr; t.sp_stride = (long long)G.tf_stride;
  GNET_LAUNCH(prof, GNET_K_WINNERS, s, winner_tpos<<<dim3(min((E + 255) / 256, 1024), B), 256, 0, s>>>(t));
  return GNET_OK;
}

int edge_stage_set_attributes() {
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_bwd_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEdgeBwdWSmem));
  return GNET_OK;
}

int edge_stage_block(const gnet_config* cfg, const gnet_shape* shape, const ParamLayout& L, const float* params, int b,
                     gnet_buffers* buf, int n_partials, hipStream_t s) {
  const int B = cfg->num_blocks, N = shape->n_det, E = (int)shape->n_edge;
  const EdgeGeom G = edge_geom(E, N);
  const BlockLayout& K = L.blk[b];
  void* prof = buf->profiler;
  const float* pt = buf->packed_t;
  const int* wg_off = buf->rl_scratch + (size_t)(B + 1) * G.n_wg;
  EdgeBwdWArgs e;
  e.n_edge = E; e.n_det = N;
  e.wcount = wg_off + (size_t)(b - 1) * (G.n_wg + 1) + G.n_wg;
  e.wlist = buf->wlist + (size_t)(b - 1) * G.wl_stride;
  e.edge_c = buf->edge_c; e.edge_nz = buf->edge_nz;
  e.apos = buf->apos + (size_t)(b - 1) * G.ap_stride;
  e.xmask = (const unsigned long long*)buf->xmask + (size_t)(b - 1) * G.xm_stride;
  e.pw = buf->pw_feats; e.rc = buf->blk_rc[b]; e.rn = buf->blk_rn[b]; e.d_pc = buf->d_pc;
  e.w1t = pt + packed_w1_off(L, b); e.w2 = params + K.w2;
  e.d_pw = buf->d_pw; e.g1c = buf->d_g1;
  e.arena = buf->arena; e.stride = arena_stride(L.total); e.o_w2 = K.w2; e.o_b2 = K.b2;
  e.o_w1 = K.w1 + (L.raw ? (int64_t)2 * L.cprime * D_P : 0); e.w1_rows = L.raw ? 7 : D_E;
  GNET_TRACE_SET(e, "EDGE_BWD", b == B / 2);
  GNET_LAUNCH(prof, GNET_K_EDGE_BWD, s, edge_bwd_w<<<n_partials, 64 * EBW_WAVES, kEdgeBwdWSmem, s>>>(e));
  return GNET_OK;
}
