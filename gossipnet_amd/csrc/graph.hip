// Neighbour-graph construction: N x N box IoU sweep -> ordered CSR edge list.
// Replaces network.py:170-176 (_xyxy_to_boxdata, _iou of dets with dets) and
// network.py:192-195 (tf.where(det_det_iou >= neighbor_thresh), row-major order).
//
// Bit-exactness contract: iou = inter / ((a_area + b_area) - inter), inter = w*h with
// w = max(0, min(ax2,bx2) - max(ax1,bx1)); no FMA contraction (-ffp-contract=off) and
// IEEE-correct division (hipcc default), compared against the fp32 threshold.
//
// Two passes (count -> exclusive scan -> fill) so that edges come out in row-major order
// without materialising the N x N matrix.  HBM-bound in principle (16 B per detection read,
// 12 B per edge written) but at these sizes the sweep is ALU/latency bound: column boxes are
// staged through LDS as SoA tiles and every wave tests 4 rows against each 64-column group.
#include "common.hpp"

namespace {

constexpr int kRowsPerWave = 4;
constexpr int kWaves = 4;
constexpr int kRowsPerBlock = kRowsPerWave * kWaves;
constexpr int kColTile = 1024;

__device__ __forceinline__ int image_of(const int* __restrict__ det_off, int n_img, int row) {
  int lo = 0, hi = n_img;  // det_off[lo] <= row < det_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (det_off[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

template <bool FILL>
__global__ void __launch_bounds__(256) graph_sweep(const float4* __restrict__ dets, int n,
                                                   const int* __restrict__ det_off, int n_img, float thr,
                                                   int* __restrict__ deg, const int* __restrict__ row_ptr,
                                                   int* __restrict__ edge_c, int* __restrict__ edge_n,
                                                   float* __restrict__ edge_iou) {
  __shared__ float sx1[kColTile], sy1[kColTile], sx2[kColTile], sy2[kColTile], sar[kColTile];
  // FILL: the hits of a row (a few per 64-column group) collect in a 128-entry LDS ring and leave as whole
  // 64-entry groups -- one full-wave store per array instead of one store instruction per group with hits
  __shared__ int rng_n[FILL ? kRowsPerBlock * 128 : 1];
  __shared__ float rng_v[FILL ? kRowsPerBlock * 128 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int brow0 = blockIdx.x * kRowsPerBlock;
  const int brow1 = min(n, brow0 + kRowsPerBlock) - 1;
  // column range of the block = images touched by its rows
  const int cmin = det_off[image_of(det_off, n_img, brow0)];
  const int cmax = det_off[image_of(det_off, n_img, brow1) + 1];

  float ax1[kRowsPerWave], ay1[kRowsPerWave], ax2[kRowsPerWave], ay2[kRowsPerWave], aar[kRowsPerWave];
  int lo[kRowsPerWave], hi[kRowsPerWave], pos[kRowsPerWave], base[kRowsPerWave], flushed[kRowsPerWave];
#pragma unroll
  for (int q = 0; q < kRowsPerWave; ++q) {
    const int row = brow0 + wave * kRowsPerWave + q;
    if (row < n) {
      const float4 b = dets[row];
      ax1[q] = b.x; ay1[q] = b.y; ax2[q] = b.z; ay2[q] = b.w;
      aar[q] = (b.z - b.x) * (b.w - b.y);           // network.py:468-471
      const int img = image_of(det_off, n_img, row);
      lo[q] = det_off[img]; hi[q] = det_off[img + 1];
      base[q] = FILL ? row_ptr[row] : 0; pos[q] = 0; flushed[q] = 0;
    } else {
      ax1[q] = ay1[q] = ax2[q] = ay2[q] = aar[q] = 0.f;
      lo[q] = hi[q] = 0; pos[q] = 0; base[q] = 0; flushed[q] = 0;
    }
  }

  for (int c0 = cmin; c0 < cmax; c0 += kColTile) {
    __syncthreads();
    const int tile_n = min(kColTile, cmax - c0);
    for (int i = threadIdx.x; i < tile_n; i += 256) {
      const float4 b = dets[c0 + i];
      sx1[i] = b.x; sy1[i] = b.y; sx2[i] = b.z; sy2[i] = b.w;
      sar[i] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    for (int i0 = 0; i0 < tile_n; i0 += 64) {
      const int i = i0 + lane;
      const bool valid = i < tile_n;
      const int j = c0 + i;
      const int ii = valid ? i : 0;
      const float bx1 = sx1[ii], by1 = sy1[ii], bx2 = sx2[ii], by2 = sy2[ii], bar = sar[ii];
#pragma unroll
      for (int q = 0; q < kRowsPerWave; ++q) {
        // network.py:504-510
        const float w = fmaxf(0.0f, fminf(ax2[q], bx2) - fmaxf(ax1[q], bx1));
        const float h = fmaxf(0.0f, fminf(ay2[q], by2) - fmaxf(ay1[q], by1));
        const float inter = w * h;
        const bool inrange = valid && j >= lo[q] && j < hi[q];
        // a pair with zero intersection has iou 0 (or NaN): it can only pass a threshold <= 0
        const bool cand = inrange && (inter > 0.0f || !(thr > 0.0f));
        if (__ballot(cand) == 0ull) continue;
        const float uni = (aar[q] + bar) - inter;    // network.py:480
        const float iou = inter / uni;               // network.py:481
        const bool pred = inrange && (iou >= thr);   // network.py:192-193
        const unsigned long long mask = __ballot(pred);
        if (FILL) {
          const int slot = (wave * kRowsPerWave + q) * 128;
          if (pred) {
            const int p = pos[q] + __popcll(mask & ((1ull << lane) - 1ull));
            rng_n[slot + (p & 127)] = j;
            rng_v[slot + (p & 127)] = iou;
          }
          pos[q] += __popcll(mask);
          if (pos[q] - flushed[q] >= 64) {                 // wave-uniform
            wave_lds_sync();
            const int k = flushed[q] + lane;
            edge_c[base[q] + k] = brow0 + wave * kRowsPerWave + q;
            edge_n[base[q] + k] = rng_n[slot + (k & 127)];
            edge_iou[base[q] + k] = rng_v[slot + (k & 127)];
            flushed[q] += 64;
            wave_lds_sync();                               // the ring entries are rewritten by later groups
          }
        } else {
          pos[q] += __popcll(mask);
        }
      }
    }
  }
  if (FILL) {
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const int slot = (wave * kRowsPerWave + q) * 128;
      const int k = flushed[q] + lane;
      if (k < pos[q]) {
        edge_c[base[q] + k] = brow0 + wave * kRowsPerWave + q;
        edge_n[base[q] + k] = rng_n[slot + (k & 127)];
        edge_iou[base[q] + k] = rng_v[slot + (k & 127)];
      }
    }
  }
  if (!FILL && lane == 0) {
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const int row = brow0 + wave * kRowsPerWave + q;
      if (row < n) deg[row] = pos[q];
    }
  }
}

// Exclusive scan of deg[0..n) into out[0..n], out[n] = total.  One workgroup.
// exclusive scan of deg[0..n) into out[0..n], out[n] = total.  One workgroup of 16 waves; 4096 elements per pass:
// every thread takes four consecutive elements (coalesced 16-byte load), waves scan with shuffles, the 16 wave totals
// go through LDS, a running carry links the passes.  (The first version -- 16 strided elements per thread and a
// 10-step 1024-thread Hillis-Steele scan with 20 barriers -- took 40 us for 16 000 rows.)
__global__ void __launch_bounds__(1024) exclusive_scan(const int* __restrict__ deg, int n, int* __restrict__ out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int carry = 0;
  for (int base = 0; base < n; base += 4096) {
    const int i0 = base + 4 * t;
    int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q < n ? deg[i0 + q] : 0;
    const int mine = v[0] + v[1] + v[2] + v[3];
    int incl = mine;                                    // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += w < wave ? wsum[w] : 0;
    int run = carry + woff + incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (i0 + q < n) out[i0 + q] = run; run += v[q]; }
    if (t == 1023) carry_s = run;                       // total so far (thread 1023 holds the last elements of the pass)
    __syncthreads();
    carry = carry_s;
  }
  if (t == 0) out[n] = carry;
}

// edge_t[e] = position of the reversed pair: for e = (c, n) find c in row n (columns ascending).
__global__ void __launch_bounds__(256) graph_transpose(const int* __restrict__ row_ptr, const int* __restrict__ edge_c,
                                                       const int* __restrict__ edge_n, long long n_edge,
                                                       int* __restrict__ edge_t) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edge) return;
  const int c = edge_c[e], n = edge_n[e];
  int lo = row_ptr[n], hi = row_ptr[n + 1] - 1, pos = -1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int v = edge_n[mid];
    if (v == c) { pos = mid; break; }
    if (v < c) lo = mid + 1; else hi = mid - 1;
  }
  edge_t[e] = pos;
}

}  // namespace

extern "C" int gnet_graph_count(const float* dets, int32_t n_det, const int32_t* det_off, int32_t n_img,
                                float thresh, int32_t* row_ptr, int32_t* scratch, gnet_stream_t stream) {
  clear_hip_error();
  if (n_det < 0 || n_img < 1 || !row_ptr || !det_off) return GNET_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n_det == 0) {
    HIP_CHECK_RET(hipMemsetAsync(row_ptr, 0, sizeof(int32_t), s));
    return GNET_OK;
  }
  if (!dets || !scratch) return GNET_ERR_INVALID;
  const int grid = (n_det + kRowsPerBlock - 1) / kRowsPerBlock;
  graph_sweep<false><<<grid, 256, 0, s>>>((const float4*)dets, n_det, det_off, n_img, thresh, scratch,
                                          nullptr, nullptr, nullptr, nullptr);
  exclusive_scan<<<1, 1024, 0, s>>>(scratch, n_det, row_ptr);
  return launch_status();
}

extern "C" int gnet_graph_fill(const float* dets, int32_t n_det, const int32_t* det_off, int32_t n_img,
                               float thresh, const int32_t* row_ptr, int32_t* edge_c, int32_t* edge_n,
                               float* edge_iou, gnet_stream_t stream) {
  clear_hip_error();
  if (n_det < 0 || n_img < 1 || !row_ptr || !det_off) return GNET_ERR_INVALID;
  if (n_det == 0) return GNET_OK;
  if (!dets || !edge_c || !edge_n || !edge_iou) return GNET_ERR_INVALID;
  const int grid = (n_det + kRowsPerBlock - 1) / kRowsPerBlock;
  graph_sweep<true><<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)dets, n_det, det_off, n_img, thresh,
                                                          nullptr, row_ptr, edge_c, edge_n, edge_iou);
  return launch_status();
}

extern "C" int gnet_graph_transpose(const int32_t* row_ptr, const int32_t* edge_c, const int32_t* edge_n, int64_t n_edge,
                                    int32_t* edge_t, gnet_stream_t stream) {
  clear_hip_error();
  if (n_edge < 0) return GNET_ERR_INVALID;
  if (n_edge == 0) return GNET_OK;
  if (!row_ptr || !edge_c || !edge_n || !edge_t) return GNET_ERR_INVALID;
  graph_transpose<<<(int)((n_edge + 255) / 256), 256, 0, (hipStream_t)stream>>>(row_ptr, edge_c, edge_n, n_edge, edge_t);
  return launch_status();
}
