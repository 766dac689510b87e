// Shared device helpers + host-side parameter layout for libgossipnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gossipnet_hip.h"

// ---- compiled dimensions (the two shipped experiment configs; see gnet_config) ----------
constexpr int D_S = 128;   // shortcut_dim
constexpr int D_R = 32;    // reduced_dim
constexpr int D_P = 64;    // pairfeat_dim
constexpr int D_H = 256;   // pwfeat_dim
constexpr int D_E = 32;    // pwfeat_narrow_dim
constexpr int D_HEAD = 128;
// partial weight-gradient copies per parameter (upper bound on writer workgroups per kernel)
constexpr int GNET_ARENA_PARTIALS = 512;
// row stride of the partial-gradient arena: the parameter count rounded up to 64 floats (16-byte loads in reduce_partials)
__host__ __device__ inline long long arena_stride(long long total) { return (total + 63) & ~63ll; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HIP_CHECK_RET(expr)                    \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return GNET_ERR_HIP; \
  } while (0)

// ---- optional event profiler (see gnet_profiler_* in gossipnet_hip.h) ------------------------
struct GnetProfiler {
  uint32_t mask;
  int cap, n;
  hipEvent_t* ev0;
  hipEvent_t* ev1;
  int* cls;
  int stride;                       // bracket every stride-th launch of a selected class (1 = every launch)
  int seen[GNET_KCLASS_COUNT];      // launches of each class since the last read
};
struct ProfScope {
  GnetProfiler* p; int idx; hipStream_t s;
  ProfScope(void* prof, int cls, hipStream_t stream) : p((GnetProfiler*)prof), idx(-1), s(stream) {
    if (p && ((p->mask >> cls) & 1u) && (p->seen[cls]++ % p->stride) == 0 && p->n < p->cap) {
      idx = p->n++;
      p->cls[idx] = cls;
      (void)hipEventRecord(p->ev0[idx], s);
    }
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(p->ev1[idx], s); }
};
#define GNET_LAUNCH(prof, cls, stream, ...) \
  do { ProfScope _ps((prof), (cls), (stream)); __VA_ARGS__; } while (0)

// ---- optional workgroup time stamps (measurement builds only: -DGNET_TRACE, see tools/wg_trace.py) ----------------
// The shipped library carries none of this.  A trace build adds a pointer to a kernel's argument struct, taken from the
// environment variable GNET_TRACE_<NAME> at the launch site; wave 0 of every workgroup writes wall_clock64() (the
// 100 MHz constant clock, common to all CUs) into trace[blockIdx.x * 16 + slot].
#ifdef GNET_TRACE
#include <stdlib.h>
#define GNET_TRACE_FIELD unsigned long long* trace;
static inline unsigned long long* gnet_trace_ptr(const char* name) {
  const char* v = getenv(name);
  return v ? (unsigned long long*)strtoull(v, nullptr, 10) : nullptr;
}
#define GNET_TRACE_SET(args, name, cond) (args).trace = (cond) ? gnet_trace_ptr("GNET_TRACE_" name) : nullptr
#define GSTAMP(args, slot)                                                                      \
  do {                                                                                          \
    if ((args).trace && threadIdx.x == 0 && (slot) < 16)                                        \
      (args).trace[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64();                          \
  } while (0)
// the same from the first lane of another wave (how far apart are the waves of a workgroup?)
#define GSTAMP_W(args, slot, thread)                                                            \
  do {                                                                                          \
    if ((args).trace && threadIdx.x == (thread) && (slot) < 16)                                 \
      (args).trace[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64();                          \
  } while (0)
#else
#define GSTAMP_W(args, slot, thread) ((void)0)
#define GNET_TRACE_FIELD
#define GNET_TRACE_SET(args, name, cond) ((void)0)
#define GSTAMP(args, slot) ((void)0)
#endif

// HIP keeps the last error of ANY runtime call of the thread (torch's included): clear it on entry.
static inline void clear_hip_error() { (void)hipGetLastError(); }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? GNET_OK : GNET_ERR_HIP; }

// ---- parameter layout (offsets in floats into the flat buffer; see gossipnet_hip.h) ------
struct BlockLayout {
  int64_t wr, br;    // reduce_dim [128,32], [32]
  int64_t w1, b1;    // pw_fc1 [kp+64,64] rows 0..kp-1 pairwise (kp = 32, or 2C'+7 with num_pwfeat_fc = 0), then 32 centre, 32 neighbour; [64]
  int64_t w2, b2;    // pw_fc2 [64,64]
  int64_t w3, b3;    // fc1 [64,64]
  int64_t w4, b4;    // fc2 [64,128]
  int64_t wrn, brn;  // reduce_dim_neighbor [128,32], [32] (neighbor_feats only; -1 otherwise)
};
struct ParamLayout {
  int dpw;           // 2*C' + 7
  int cprime;        // C if multiclass else 1
  int64_t pw1, pb1, pw2, pb2, pw3, pb3;
  BlockLayout blk[GNET_MAX_BLOCKS + 1];  // 1-based
  int64_t hw1, hb1, hw2, hb2, hwl, hbl;
  int64_t total;
  int raw;           // num_pwfeat_fc == 0: no pw-MLP, the blocks' pw_fc1 reads the raw 2C'+7 geometry columns (network.py:217-221)
  int kp;            // rows of the pairwise part of a block's pw_fc1: 32 (pwfeat_narrow_dim), or dpw when raw
};

static inline int config_supported(const gnet_config* c) {
  if (!c) return 0;
  return c->num_classes >= 1 && c->num_blocks >= 1 && c->num_blocks <= GNET_MAX_BLOCKS &&
         c->shortcut_dim == D_S && c->reduced_dim == D_R && c->pairfeat_dim == D_P &&
         ((c->num_pwfeat_fc == 3 && c->pwfeat_dim == D_H && c->pwfeat_narrow_dim == D_E) || c->num_pwfeat_fc == 0) &&
         c->predict_fc_dim == D_HEAD && c->num_predict_fc == 3 && c->num_block_pw_fc == 2 &&
         c->num_block_fc == 2;
}

static inline ParamLayout make_layout(const gnet_config* c) {
  ParamLayout L;
  L.cprime = c->num_classes > 1 ? c->num_classes : 1;
  L.dpw = 2 * L.cprime + 7;
  L.raw = c->num_pwfeat_fc == 0;
  L.kp = L.raw ? L.dpw : D_E;
  int64_t o = 0;
  if (!L.raw) {
    L.pw1 = o; o += (int64_t)L.dpw * D_H;
    L.pb1 = o; o += D_H;
    L.pw2 = o; o += D_H * D_H;
    L.pb2 = o; o += D_H;
    L.pw3 = o; o += D_H * D_E;
    L.pb3 = o; o += D_E;
  } else { L.pw1 = L.pb1 = L.pw2 = L.pb2 = L.pw3 = L.pb3 = 0; }
  for (int b = 1; b <= c->num_blocks; ++b) {
    BlockLayout& B = L.blk[b];
    B.wr = o; o += D_S * D_R;
    B.br = o; o += D_R;
    B.w1 = o; o += (int64_t)(L.kp + 2 * D_R) * D_P;
    B.b1 = o; o += D_P;
    B.w2 = o; o += D_P * D_P;
    B.b2 = o; o += D_P;
    B.w3 = o; o += D_P * D_P;
    B.b3 = o; o += D_P;
    B.w4 = o; o += D_P * D_S;
    B.b4 = o; o += D_S;
    if (c->neighbor_feats) { B.wrn = o; o += D_S * D_R; B.brn = o; o += D_R; } else { B.wrn = B.brn = -1; }
  }
  L.hw1 = o; o += D_S * D_HEAD;
  L.hb1 = o; o += D_HEAD;
  L.hw2 = o; o += D_HEAD * D_HEAD;
  L.hb2 = o; o += D_HEAD;
  L.hwl = o; o += D_HEAD;
  L.hbl = o; o += 1;
  L.total = o;
  return L;
}

// Where the TRANSPOSED copy of block b's pw_fc1 lives in packed_t.  Every consumer reads it as [64][96] (pairwise | centre |
// neighbour columns).  With a pw-MLP that is the transpose of the [96, 64] variable, at the variable's own offset; with
// num_pwfeat_fc = 0 the variable is [2C'+7+64, 64] -- smaller than 96 rows for a single class -- and the [64][96] copy (geometry
// rows | zeros | centre | neighbour, forward.hip pack_transpose) lives behind the parameters' copies.
constexpr int64_t W1T_FLOATS = (int64_t)D_P * (D_E + 2 * D_R);
static inline int64_t packed_w1_off(const ParamLayout& L, int b) { return L.raw ? L.total + (int64_t)(b - 1) * W1T_FLOATS : L.blk[b].w1; }
// The pw-MLP's fc2 / fc3 weights as bf16 three-term MFMA operand fragments (forward.hip pack_pw_bf16; 32-bit words, every array
// [term hi | mid | lo][wave 8][k-step][lane 64][4 words]: a lane's eight k-slots of a k-step of 16 = one 16-byte load), behind the
// transposed copies, 16-byte aligned.  Only with a pw-MLP.
//   W2A  pw_fwd3's fc2 (transposed: A = W2^T), 16 k-steps   W3B  pw_fwd3's fc3 (B = W3), 2 k-steps per wave
//   W2D  pw_bwd_bf's d h1 = d2 . W2^T (B = W2^T), 16 k-steps   W3D  pw_bwd_bf's d2 = d3 . W3^T (B = W3^T), 2 k-steps
constexpr int64_t PWBF_W2A = 0, PWBF_W2A_WORDS = 3ll * 8 * 16 * 64 * 4;
constexpr int64_t PWBF_W3B = PWBF_W2A + PWBF_W2A_WORDS, PWBF_W3B_WORDS = 3ll * 8 * 2 * 64 * 4;
constexpr int64_t PWBF_W2D = PWBF_W3B + PWBF_W3B_WORDS, PWBF_W2D_WORDS = PWBF_W2A_WORDS;
constexpr int64_t PWBF_W3D = PWBF_W2D + PWBF_W2D_WORDS, PWBF_W3D_WORDS = PWBF_W3B_WORDS;
constexpr int64_t PWBF_WORDS = PWBF_W3D + PWBF_W3D_WORDS;
static inline int64_t packed_pwbf_off(const ParamLayout& L) { return (L.total + 3) & ~3ll; }
static inline int64_t packed_floats(const ParamLayout& L, int nblocks) {
  return L.raw ? L.total + (int64_t)nblocks * W1T_FLOATS : packed_pwbf_off(L) + PWBF_WORDS;
}
// feature (within a wave's 32) that k-slot t of the lane's eight holds in k-step q (0 / 1) of the wave's pair: the register order of
// a 32x32 MFMA accumulator column -- crow(8 q + t, half) -- so that accumulators ARE operands (no shuffle between chained layers)
__host__ __device__ __forceinline__ int frag_feat(int q, int half, int t) { return (t & 3) + 8 * (2 * q + (t >> 2)) + 4 * half; }

// ---- geometry of the winner maps / lists of the backward edge stage (plan.hip, backward*.hip) ----------
struct EdgeGeom {
  size_t n_words;     // 64-bit words of a 1-bit-per-edge map
  size_t n_wg;        // 256-word scan chunks (covers word index n_words too: position of the end sentinel)
  size_t bm_stride;   // words between the blocks' maps / prefix arrays (= n_wg * 256)
  size_t wl_stride;   // ints between the blocks' winner lists
  size_t xm_stride;   // u64 between the blocks' extra-winner masks
  size_t tf_stride;   // bytes between the blocks' per-detection tie flags
  size_t ap_stride;   // ints between the blocks' [N,64] arg-max list positions
};
static inline EdgeGeom edge_geom(int64_t E, int64_t N) {
  EdgeGeom g;
  g.n_words = (size_t)(E + 63) / 64;
  g.n_wg = (g.n_words + 1 + 255) / 256;
  g.bm_stride = g.n_wg * 256;
  g.wl_stride = ((size_t)E + 64 + 63) & ~(size_t)63;
  g.xm_stride = ((size_t)E + 64 + 63) & ~(size_t)63;
  g.tf_stride = ((size_t)N + 32 + 63) & ~(size_t)63;
  g.ap_stride = ((size_t)N + 32) * D_P;
  return g;
}

// ---- balanced contiguous ranges: n_items dealt to n_parts so that every part gets floor or ceil of the mean --------
// part g owns [range_begin(g), range_begin(g + 1)); range_owner() is its inverse.  (Equal ceil-sized ranges leave the last
// parts empty -- with the XCD-aware part numbering all of them on one XCD: edge_bwd_w ran on seven XCDs.)
// floor(g n / parts) = g q + floor(g r / parts) with n = q parts + r: the same value in 32-bit arithmetic (0 <= g <= parts < 2^16 at every
// call site).  A 64-bit division is a subroutine of ~150 scalar instructions here -- edge_fwd_w's front ran ~900 of them (2.2 us of every
// launch) before its first request.
__host__ __device__ __forceinline__ int range_begin(int g, int n_items, int n_parts) {
  const unsigned q = (unsigned)n_items / (unsigned)n_parts, r = (unsigned)n_items - q * (unsigned)n_parts;
  return (int)((unsigned)g * q + ((unsigned)g * r) / (unsigned)n_parts);
}
__host__ __device__ __forceinline__ int range_owner(int item, int n_items, int n_parts) {
  return (int)((((long long)item + 1) * n_parts - 1) / n_items);
}

// ---- edge_fwd_w's tile ranges ------------------------------------------------------------------------------------------
// With three workgroups per CU the SIMD's issue arbiter favours the OLDEST wave: measured (tools/wg_trace.py, 8 images), the
// first workgroup dispatched to a CU walks a tile in 8.5 us, the second in 10.4, the third in 13, and with equal ranges they
// left at 125 / 143 / 160 us of a 170 us launch -- the last 35 us ran at one or two waves per SIMD.  The workgroups of an XCD
// are dispatched in blockIdx order, a third of them per "layer" of the CUs, so the ranges are sized by layer: 1.18 : 1.00 :
// 0.77 of the mean (1.215 : 1.02 : 0.765).  (If the dispatcher ever orders them differently only the balance is lost, not the result: the ranges
// still tile the list, and efw_owner() -- used for the per-detection straddle flags -- is the inverse of efw_begin().)
__host__ __device__ __forceinline__ bool efw_layered(int n_tiles, int n_waves) {
  return n_waves % (8 * 3 * 4) == 0 && n_tiles >= 6 * n_waves;
}
__host__ __device__ __forceinline__ int efw_layer_cut(int k) { return k <= 0 ? 0 : k == 1 ? 405 : k == 2 ? 745 : 1000; }   // per mille, cumulative
__host__ __device__ __forceinline__ int efw_begin(int gw, int n_tiles, int n_waves) {
  if (!efw_layered(n_tiles, n_waves)) return range_begin(gw, n_tiles, n_waves);
  if (gw >= n_waves) return n_tiles;
  const int wpx = n_waves / 8, tw = wpx / 3;                 // waves per XCD, per layer of an XCD
  const int x = (int)((unsigned)gw / (unsigned)wpx), jx = gw - x * wpx, k = (int)((unsigned)jx / (unsigned)tw), j = jx - k * tw;
  const int x0 = range_begin(x, n_tiles, 8), tx = range_begin(x + 1, n_tiles, 8) - x0;
  const int b0 = x0 + (int)((unsigned)tx * (unsigned)efw_layer_cut(k) / 1000u), b1 = x0 + (int)((unsigned)tx * (unsigned)efw_layer_cut(k + 1) / 1000u);   // (tx < 2^17 at the edge limit)
  return b0 + range_begin(j, b1 - b0, tw);
}
__host__ __device__ __forceinline__ int efw_owner(int tile, int n_tiles, int n_waves) {
  if (!efw_layered(n_tiles, n_waves)) return range_owner(tile, n_tiles, n_waves);
  const int wpx = n_waves / 8, tw = wpx / 3;
  const int x = range_owner(tile, n_tiles, 8);
  const int x0 = range_begin(x, n_tiles, 8), tx = range_begin(x + 1, n_tiles, 8) - x0;
  const int c1 = x0 + (int)((unsigned)tx * (unsigned)efw_layer_cut(1) / 1000u), c2 = x0 + (int)((unsigned)tx * (unsigned)efw_layer_cut(2) / 1000u);
  const int k = tile >= c2 ? 2 : tile >= c1 ? 1 : 0;
  const int b0 = k == 0 ? x0 : k == 1 ? c1 : c2, b1 = k == 0 ? c1 : k == 1 ? c2 : x0 + tx;
  return x * wpx + k * tw + range_owner(tile - b0, b1 - b0, tw);
}

// ---- fp32 MFMA tile primitives ---------------------------------------------------------
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// the 16 accumulator registers hold D[row][col] with col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Exact f32 fma chain.
__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Value held by the lane that owns tile row crow(r, half) (lanes 0..31 hold rows 0..31).  The row of an
// accumulator register is uniform per half-wave, so two v_readlane (scalar results, no LDS round trip --
// __shfl would be a ds_bpermute) and a select replace a cross-lane shuffle.  r must be a constant.
__device__ __forceinline__ int row_bcast(int v, int r, int half) {
  const int lo = __builtin_amdgcn_readlane(v, crow(r, 0));
  const int hi = __builtin_amdgcn_readlane(v, crow(r, 1));
  return half ? hi : lo;
}

// acc += A[32 x K] * Bt[32 x K]^T ; A, Bt row-major with leading dims lda/ldb (floats),
// both 16-byte aligned at (row*ld + 4*half).  Each lane reads 4 consecutive k per 16-B load:
// the half-waves pair k = kb+t (half 0) with k = kb+4+t (half 1), a permutation of the k sum.
template <int K>
__device__ __forceinline__ void mma_abt(f32x16& acc, const float* A, int lda, const float* Bt, int ldb,
                                        int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = A + r * lda + 4 * h;
  const float* bp = Bt + r * ldb + 4 * h;
#pragma unroll
  for (int k = 0; k < K; k += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bp + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

// The same product with the B operand already in registers: kernels whose stages are separated by barriers
// request every weight slice they will need at their top (load_bt), so that no stage waits for its own loads.
template <int K> struct BtRegs { f32x4 v[K / 8]; };
template <int K>
__device__ __forceinline__ void load_bt(BtRegs<K>& b, const float* __restrict__ Bt, int ldb, int lane) {
  const float* bp = Bt + (size_t)(lane & 31) * ldb + 4 * (lane >> 5);
#pragma unroll
  for (int k = 0; k < K; k += 8) b.v[k / 8] = *reinterpret_cast<const f32x4*>(bp + k);
}
template <int K>
__device__ __forceinline__ void mma_abt_r(f32x16& acc, const float* A, int lda, const BtRegs<K>& b, int lane) {
  const float* ap = A + (lane & 31) * lda + 4 * (lane >> 5);
#pragma unroll
  for (int k = 0; k < K; k += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.v[k / 8].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.v[k / 8].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.v[k / 8].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.v[k / 8].w, acc, 0, 0, 0);
  }
}

// Global accesses as (uniform base pointer) + (32-bit byte offset): selects the "saddr + voffset" form of
// global_load / global_store, one 32-bit VALU op per address instead of the 64-bit pointer arithmetic of
// base[index] (3-4 VALU ops).  Vector instructions are paid in MFMA time (the fp32 MFMA shares the FP32 lanes).
__device__ __forceinline__ float ldg_b(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned ldg_b(const unsigned* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(base) + byte_off);
}
// Fold a value over the two half-waves (lanes l and l ^ 32) with v_permlane32_swap (gfx950): one VALU
// instruction yields "lower half everywhere" and "upper half everywhere" -- no LDS round trip as with
// __shfl_xor(x, 32) (ds_bpermute), which sat on the critical path of the per-segment reductions.
__device__ __forceinline__ void half_bcast(unsigned x, unsigned& lo, unsigned& hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  lo = r[0]; hi = r[1];
}
__device__ __forceinline__ float half_fmax(float x) {
  unsigned lo, hi; half_bcast(__float_as_uint(x), lo, hi);
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__uint_as_float(lo)), "v"(__uint_as_float(hi)));   // no canonicalising pre-ops
  return r;
}
__device__ __forceinline__ unsigned half_add(unsigned x) { unsigned lo, hi; half_bcast(x, lo, hi); return lo + hi; }
__device__ __forceinline__ int half_min(int x) { unsigned lo, hi; half_bcast((unsigned)x, lo, hi); return min((int)lo, (int)hi); }

// ---- fp32 products on the bf16 pipe (edge_fwd_w and the kernels that must reproduce its bits) -------------------------------
// x = hi + mid + lo EXACTLY, each term a bf16: hi and mid by truncation to the upper 16 bits, lo the <= 8 significant bits that are
// left (normal numbers; the activations and weights here are finite).  A product of two fp32 operands is then the six bf16 products
// hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid (what is dropped is below 2^-24 of |a||b|), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16: 6 x 32 cycles for K = 16 against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same K -- measured
// (tools/bf16x3_probe.hip, profiles/r05_bf16x3_probe.txt): the same error against fp64 as the fp32 MFMA (1.7e-7 of sum |a||b|
// against 2.1e-7), and an edge_fwd_w-shaped tile loop 1.7x faster with the split's vector instructions included.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Bf3 { u32x4 h, m, l; };       // eight values as three bf16 terms, packed in k-slot order (slot 0 in the low half of word 0)

// the three terms of two values, packed (first value in the low halves)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SPLIT3_PACKED
#define SPLIT3_PACKED 0      /* the two remainders of a pair as ONE v_pk_add_f32 each (the same arithmetic: 9 instead of 11 instructions per pair) */
#endif
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
  const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
#if SPLIT3_PACKED
  const f32x2 r = f32x2{x0, x1} - f32x2{__uint_as_float(b0 & 0xffff0000u), __uint_as_float(b1 & 0xffff0000u)};
  const unsigned m0 = __float_as_uint(r.x), m1 = __float_as_uint(r.y);
  const f32x2 q = r - f32x2{__uint_as_float(m0 & 0xffff0000u), __uint_as_float(m1 & 0xffff0000u)};
  const float s0 = q.x, s1 = q.y;
#else
  const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);
  const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(m0 & 0xffff0000u), s1 = r1 - __uint_as_float(m1 & 0xffff0000u);
#endif
  ph = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  pm = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  pl = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ Bf3 split3_8(const f32x4 a, const f32x4 b) {       // slots 0-3 = a, 4-7 = b
  unsigned h[4], m[4], l[4];
  split3_pk(a.x, a.y, h[0], m[0], l[0]);
  split3_pk(a.z, a.w, h[1], m[1], l[1]);
  split3_pk(b.x, b.y, h[2], m[2], l[2]);
  split3_pk(b.z, b.w, h[3], m[3], l[3]);
  Bf3 t;
  t.h = u32x4{h[0], h[1], h[2], h[3]}; t.m = u32x4{m[0], m[1], m[2], m[3]}; t.l = u32x4{l[0], l[1], l[2], l[3]};
  return t;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A . B over 16 k-slots, the six products from the smallest to the largest.  ONE sequence, used by every kernel that has to
// reproduce another's bits (the tie semantics compare recomputed values for equality).
__device__ __forceinline__ f32x16 mma6(f32x16 acc, const Bf3& a, const Bf3& b) {
  acc = mfma_bf16(a.l, b.h, acc);
  acc = mfma_bf16(a.h, b.l, acc);
  acc = mfma_bf16(a.m, b.m, acc);
  acc = mfma_bf16(a.m, b.h, acc);
  acc = mfma_bf16(a.h, b.m, acc);
  acc = mfma_bf16(a.h, b.h, acc);
  return acc;
}

// relu as a signed-integer max on the bit pattern: one VALU op.  fmaxf() of an MFMA result costs two (the
// compiler canonicalises a possibly-signalling NaN first); negative floats are negative integers, -0 -> +0.
__device__ __forceinline__ float relu_bits(float v) {
  return __int_as_float(max(__float_as_int(v), 0));
}
__device__ __forceinline__ float4 ldg4_b(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg_b(float* __restrict__ base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// vmcnt is ONE in-order counter for loads and stores, and the compiler's wait insertion merges the
// pending-load state of a loop's preheader and latch conservatively: if the first tile's prefetch is
// still "pending" when the loop is entered (followed by few memory operations), every iteration waits
// as if equally few operations followed the prefetch -- i.e. for the previous tile's STORES to be
// acknowledged by L2 (measured: 9 k of 27 k cycles per edge_bwd tile).  Draining the counter once before
// the loop leaves only the latch state, and the in-loop wait becomes vmcnt(#operations issued after the
// prefetch), which never covers a store.
__device__ __forceinline__ void drain_vmem_before_loop() {
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt / lgkmcnt untouched
}

// Variants whose B operand streams from global memory (weights that do not fit in LDS): the 16-byte
// B loads run PF k-steps (8 k each) ahead of the MFMAs that consume them, in a register ring.
template <int K, int PF>
__device__ __forceinline__ void mma_abt_gB(f32x16& acc, const float* A, int lda, const float* __restrict__ Bt, int ldb,
                                           int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = A + r * lda + 4 * h;
  const float* bp = Bt + (size_t)r * ldb + 4 * h;
  constexpr int NS = K / 8;
  f32x4 ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) ring[i] = *reinterpret_cast<const f32x4*>(bp + 8 * (i < NS ? i : NS - 1));
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4 b = ring[s % PF];
    if (s + PF < NS) ring[s % PF] = *reinterpret_cast<const f32x4*>(bp + 8 * (s + PF));
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * s);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

// acc[m][n] += X[32 rows x (32*MI)]^T * Y[32 rows x (32*NJ)] (weight-gradient shape: the
// contraction runs over the 32 tile rows).  X, Y row-major; rows that do not exist must be 0 in Y.
// The same product for one row tile of X with the operands of step kk + 1 requested before the NJ MFMAs of step kk
// (pinned: otherwise every pair of MFMAs sits behind its own LDS read and wait).
template <int NJ>
__device__ __forceinline__ void mma_xty_pipelined(f32x16 (&acc)[1][NJ], const float* X, int ldx, const float* Y, int ldy, int lane) {
  const int r = lane & 31, h = lane >> 5;
  float a = X[h * ldx + r], b[NJ];
#pragma unroll
  for (int n = 0; n < NJ; ++n) b[n] = Y[h * ldy + 32 * n + r];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int row = 2 * (kk + 1 < 16 ? kk + 1 : kk) + h;
    float an = X[row * ldx + r], bn[NJ];
#pragma unroll
    for (int n = 0; n < NJ; ++n) bn[n] = Y[row * ldy + 32 * n + r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NJ; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[0][n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
#pragma unroll
    for (int n = 0; n < NJ; ++n) b[n] = bn[n];
  }
}

template <int MI, int NJ>
__device__ __forceinline__ void mma_xty(f32x16 (&acc)[MI][NJ], const float* X, int ldx, const float* Y,
                                        int ldy, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int row = 2 * kk + h;
    float a[MI], b[NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m) a[m] = X[row * ldx + 32 * m + r];
#pragma unroll
    for (int n = 0; n < NJ; ++n) b[n] = Y[row * ldy + 32 * n + r];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int n = 0; n < NJ; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Wave-level LDS hand-off: all earlier LDS accesses of this wave complete before later ones.
// (LDS operations of one wave execute in order; this pins the compiler and drains lgkmcnt.)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (segment max, tie count) packed as (float bits << 32) | count; values are >= 0 (post-ReLU)
// so the float order equals the unsigned order of the bits.  combine() is associative and
// commutative, so any reduction order gives bit-identical results.
__device__ __forceinline__ void pm_flush(unsigned long long* addr, float m, unsigned cnt) {
  const unsigned mb = __float_as_uint(m);
  unsigned long long old = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    const unsigned ob = (unsigned)(old >> 32);
    if (ob > mb) return;
    const unsigned long long neu =
        (ob == mb) ? (old + cnt) : (((unsigned long long)mb << 32) | (unsigned long long)cnt);
    if (__hip_atomic_compare_exchange_strong(addr, &old, neu, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT))
      return;
  }
}

__device__ __forceinline__ void atomic_add_f32(float* addr, float v) {
  __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
