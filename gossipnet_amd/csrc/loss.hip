// Loss side of Gnet (network.py:174-187, 275-313) and the DetectionMatching op
// (matching_module/det_matching.cc:72-160), all on device (the reference op is CPU-only and
// forces a D->H->D round trip per step).
//
//   anno_rows     one wave per detection: det_anno_iou = _iou(dets, gt, crowd) with IoA on crowd columns, class-masked,
//                 and in the same pass the detection's best two non-crowd candidates (iou >= 0.5) and first crowd hit
//   match_rank    score order: rank = #{j : s_j > s_i or (s_j == s_i and j > i)}
//                 (std::sort ascending + reverse with ties resolved "higher index first")
//   match_greedy  one wave per image walks the detections in score order, 64 at a time, resolving each batch in
//                 parallel rounds that reproduce the sequential decisions
//   loss_kernel   class weighting, sigmoid cross-entropy, per-image sums, d loss / d logit
#include "common.hpp"

namespace {

__device__ __forceinline__ int image_of(const int* __restrict__ off, int n_img, int row) {
  int lo = 0, hi = n_img;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// wave-wide maximum of a 64-bit key (butterfly over the 64 lanes; every lane ends with the maximum)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned hi = __shfl_xor((unsigned)(k >> 32), o), lo = __shfl_xor((unsigned)k, o);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    k = other > k ? other : k;
  }
  return k;
}

// One WAVE per detection, lanes over the ground-truth boxes of its image (64 per pass): the box IoU row
// (network.py:475-488, IoA on crowd columns, class mask :182-187) leaves as one coalesced store per pass, and -- with
// CAND -- the matching's per-detection candidate records come out of the same pass: the two best non-crowd candidates
// as keys (iou bits << 32 | gt index; the reference's inner loop keeps the maximum iou >= 0.5 and lets a later equal iou
// win, det_matching.cc:142-148, i.e. the maximum key), their number, and the first crowd GT with iou >= 0.5
// (det_matching.cc:138).  IOU = false reads a given IoU matrix instead of computing it (the standalone op).
// (A thread per detection walking its row -- 64 lanes on 64 different cache lines, one element each per step -- took
// 1.9 ms beside the forward pass for a 5 MB matrix.)
template <bool IOU, bool CAND>
__global__ void __launch_bounds__(256) anno_rows(const float4* __restrict__ dets, const int* __restrict__ det_classes,
                                                 const int* __restrict__ det_off, const float4* __restrict__ gts,
                                                 const unsigned char* __restrict__ gt_crowd,
                                                 const int* __restrict__ gt_classes, const int* __restrict__ gt_off,
                                                 const long long* __restrict__ anno_off, int n_det, int n_img,
                                                 int multiclass, float* __restrict__ out,
                                                 unsigned long long* __restrict__ k1, unsigned long long* __restrict__ k2,
                                                 int* __restrict__ ncand, int* __restrict__ cfirst) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (d >= n_det) return;
  const int img = image_of(det_off, n_img, d);
  const int g0 = gt_off[img], m = gt_off[img + 1] - g0;
  float* row = out + anno_off[img] + (long long)(d - det_off[img]) * m;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f); float a_area = 0.f; int dc = 0;
  if (IOU) { a = dets[d]; a_area = (a.z - a.x) * (a.w - a.y); dc = multiclass ? det_classes[d] : 0; }
  unsigned long long b1 = 0, b2 = 0;
  int nc = 0, cf = -1;
  for (int gb = 0; gb < m; gb += 64) {
    const int g = gb + lane;
    const bool in = g < m;
    float v = 0.f;
    const bool crowd = in && gt_crowd && gt_crowd[g0 + g];
    if (IOU) {
      if (in) {
        const float4 b = gts[g0 + g];
        const float b_area = (b.z - b.x) * (b.w - b.y);
        const float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
        const float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
        const float inter = w * h;
        if (crowd) v = inter / a_area;                           // network.py:485-488
        else v = inter / ((a_area + b_area) - inter);            // network.py:480-481
        if (multiclass && dc != gt_classes[g0 + g]) v = 0.0f;    // network.py:182-187
        row[g] = v;
      }
    } else if (in) {
      v = row[g];
    }
    if (CAND) {
      const bool hit = in && v >= 0.5f;
      const unsigned long long crowd_hits = __ballot(hit && crowd);
      if (cf < 0 && crowd_hits) cf = gb + (int)__builtin_ctzll(crowd_hits);
      const bool cand = hit && !crowd;
      nc += (int)__popcll(__ballot(cand));
      const unsigned long long key = cand ? (((unsigned long long)__float_as_uint(v) << 32) | (unsigned)g) : 0ull;
      const unsigned long long p1 = wave_max_u64(key);           // keys are distinct (the index is part of them)
      const unsigned long long p2 = wave_max_u64(key == p1 ? 0ull : key);
      // merge the pass's best two into the running best two
      if (p1 > b1) { b2 = p2 > b1 ? p2 : b1; b1 = p1; } else if (p1 > b2) { b2 = p1; }
    }
  }
  if (CAND && lane == 0) { k1[d] = b1; k2[d] = b2; ncand[d] = nc; cfirst[d] = cf; }
}

// order[lo + rank] = d with rank = number of detections of the image that sort before d (score descending,
// ties by index descending -- the stable argsort of det_matching.cc:95-100 read backwards).  Sixteen adjacent lanes
// share a detection and take interleaved columns of the LDS score tile; 16 detections per workgroup (four lanes and 64
// detections per workgroup left 32 workgroups for a 2 000-detection image: a 125-compare chain per lane on an eighth of the chip).
constexpr int RANK_LANES = 16, RANK_DETS = 256 / RANK_LANES;
__global__ void __launch_bounds__(256) match_rank(const float* __restrict__ score, const int* __restrict__ det_off,
                                                  int n_det, int n_img, int* __restrict__ order) {
  __shared__ float ss[1024];
  const int q = threadIdx.x & (RANK_LANES - 1);
  const int d = blockIdx.x * RANK_DETS + (threadIdx.x / RANK_LANES);
  const int dd = min(d, n_det - 1);
  const int b0 = blockIdx.x * RANK_DETS, b1 = min(n_det, b0 + RANK_DETS) - 1;
  const int cmin = det_off[image_of(det_off, n_img, b0)];
  const int cmax = det_off[image_of(det_off, n_img, b1) + 1];
  const int img = image_of(det_off, n_img, dd);
  const int lo = det_off[img], hi = det_off[img + 1];
  // (a NaN score -- a diverged run -- ranks as -inf: `>` and `==` are false for it, the ranks would collide and leave
  //  entries of `order` unwritten, i.e. stale indices of an earlier batch for match_greedy to follow out of bounds)
  float s = score[dd];
  s = s == s ? s : -INFINITY;
  int rank = 0;
  for (int c0 = cmin; c0 < cmax; c0 += 1024) {
    __syncthreads();
    const int tn = min(1024, cmax - c0);
    for (int i = threadIdx.x; i < tn; i += 256) { const float v = score[c0 + i]; ss[i] = v == v ? v : -INFINITY; }
    __syncthreads();
    const int jlo = max(lo, c0), jhi = min(hi, c0 + tn);
    for (int j = jlo + q; j < jhi; j += RANK_LANES) {
      const float t = ss[j - c0];
      rank += (t > s || (t == s && j > dd)) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 1; o < RANK_LANES; o <<= 1) rank += __shfl_xor(rank, o);
  if (q == 0 && d < n_det) order[lo + rank] = d;
}

// One wave per image walks the detections in score order, 64 at a time.  The reference loop (det_matching.cc:118-157)
// is sequential, but within a batch most decisions do not depend on each other, so the batch is resolved in rounds:
//   - a detection's wish is its best candidate that is still free (c1, else c2);
//   - every unresolved detection registers as "interested" in each of its free candidates (LDS atomicMin of the
//     lane id per GT); a detection whose wish is owned by itself cannot lose it to an earlier one -> it commits;
//   - a detection with no free candidate among <= 2 falls to its crowd hit (state-independent);
//   - a detection with more than two candidates may want GTs the two keys do not name: while it is unresolved
//     and not committing, no later detection commits; when it is the first unresolved one and both keys are
//     taken, its IoU row is scanned by the whole wave.
// The first unresolved detection always resolves, so the rounds terminate; every commit is the decision the
// sequential loop takes at that detection's turn (its wish is free then, and nobody before it asked for it).
// taken[] / owner[] of the image's ground-truth boxes live in (dynamic) LDS sized for the TOTAL number of boxes of the
// call -- the one count the host knows (the per-image counts are device data), and an upper bound of every image's.
constexpr int kMaxGt = 24576;      // 5 bytes per box: 120 KB of the 160 KB LDS
__global__ void __launch_bounds__(64) match_greedy(const float* __restrict__ iou, const long long* __restrict__ anno_off,
                                                   const int* __restrict__ det_off, const int* __restrict__ gt_off,
                                                   const unsigned char* __restrict__ ignore,
                                                   const int* __restrict__ order, const unsigned long long* __restrict__ k1,
                                                   const unsigned long long* __restrict__ k2,
                                                   const int* __restrict__ ncand, const int* __restrict__ cfirst,
                                                   float* __restrict__ labels, float* __restrict__ weights,
                                                   int* __restrict__ assign, int cap) {
  extern __shared__ __attribute__((aligned(16))) int match_lds[];
  int* owner = match_lds;                                                    // [cap]
  unsigned char* taken = reinterpret_cast<unsigned char*>(match_lds + cap);   // [cap]
  const int img = blockIdx.x, lane = threadIdx.x;
  const int d0 = det_off[img], d1 = det_off[img + 1];
  const int g0 = gt_off[img], m = gt_off[img + 1] - g0;
  const float* ibase = iou + anno_off[img];
  for (int g = lane; g < m; g += 64) { taken[g] = 0; owner[g] = 64; }
  // the records of the next 64 detections (two dependent loads: order -> k1/k2/ncand/cfirst) are requested
  // before the current ones are resolved
  int n_det_ = d0 + lane < d1 ? order[d0 + lane] : -1;
  unsigned long long n_c1 = n_det_ >= 0 ? k1[n_det_] : 0ull, n_c2 = n_det_ >= 0 ? k2[n_det_] : 0ull;
  int n_nc = n_det_ >= 0 ? ncand[n_det_] : 0, n_cf = n_det_ >= 0 ? cfirst[n_det_] : -1;
  __syncthreads();
  for (int p0 = d0; p0 < d1; p0 += 64) {
    const int p = p0 + lane;
    const bool act = p < d1;
    const int det = n_det_;
    const int ga = (int)(unsigned)n_c1, gb = (int)(unsigned)n_c2;
    const int nc = n_nc, cf = n_cf;
    {
      const int pn = p0 + 64 + lane;
      n_det_ = pn < d1 ? order[pn] : -1;
      n_c1 = n_det_ >= 0 ? k1[n_det_] : 0ull; n_c2 = n_det_ >= 0 ? k2[n_det_] : 0ull;
      n_nc = n_det_ >= 0 ? ncand[n_det_] : 0; n_cf = n_det_ >= 0 ? cfirst[n_det_] : -1;
    }
    int res = (act && nc == 0) ? cf : -1;     // no regular candidate: state-independent
    bool open = act && nc > 0;
    while (__ballot(open)) {
      const bool f1 = open && !taken[ga];
      const bool f2 = open && nc >= 2 && !taken[gb];
      const int wish = f1 ? ga : (f2 ? gb : -1);
      if (open && wish < 0 && nc <= 2) { res = cf; open = false; }   // det_matching.cc:134-148 fall-through
      if (f1) atomicMin(&owner[ga], lane);
      if (f2) atomicMin(&owner[gb], lane);
      __syncthreads();
      const bool mine = open && wish >= 0 && owner[wish] == lane;
      __syncthreads();
      if (f1) owner[ga] = 64;
      if (f2) owner[gb] = 64;
      const unsigned long long hold = __ballot(open && nc > 2 && !mine);
      const int fence = hold ? __builtin_ctzll(hold) : 64;
      if (mine && lane < fence) { taken[wish] = 1; res = wish; open = false; }
      __syncthreads();
      const unsigned long long rest = __ballot(open);
      if (rest && __builtin_ctzll(rest) == fence && __builtin_amdgcn_readlane((int)(wish < 0), fence)) {
        // more than two candidates and the best two are taken: scan the row of that detection
        const int dl = __builtin_amdgcn_readlane(det, fence);
        const float* row = ibase + (long long)(dl - d0) * m;
        unsigned long long best = 0;
        for (int g = lane; g < m; g += 64) {
          const float v = row[g];
          if (v >= 0.5f && !ignore[g0 + g] && !taken[g]) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)g;
            best = key > best ? key : best;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long other = __shfl_xor(best, o);
          best = other > best ? other : best;
        }
        const int mt = best ? (int)(unsigned)best : -1;
        if (lane == fence) {
          if (mt >= 0) taken[mt] = 1;
          res = mt >= 0 ? mt : cf;
          open = false;
        }
        __syncthreads();
      }
    }
    if (act) {
      assign[det] = res;
      labels[det] = res >= 0 ? 1.0f : 0.0f;
      weights[det] = (res >= 0 && ignore[g0 + res]) ? 0.0f : 1.0f;
    }
  }
}

// network.py:282-313.  One workgroup per image.
constexpr int LOSS_THREADS = 1024;     // one workgroup per image: 2 detections per thread at N = 2000 (it was a chain of 8)
__global__ void __launch_bounds__(LOSS_THREADS) loss_kernel(const float* __restrict__ pred, const float* __restrict__ labels,
                                                   float* __restrict__ weights, const int* __restrict__ assign,
                                                   const int* __restrict__ det_off, const int* __restrict__ gt_off,
                                                   const unsigned char* __restrict__ gt_crowd,
                                                   const int* __restrict__ gt_classes,
                                                   const float* __restrict__ class_weights, int num_classes,
                                                   int normalize, float loss_mult, float grad_scale,
                                                   float* __restrict__ loss, float* __restrict__ d_logits) {
  __shared__ float red[LOSS_THREADS];
  const int img = blockIdx.x;
  const int d0 = det_off[img], d1 = det_off[img + 1], g0 = gt_off[img];
  const int n = d1 - d0;
  const float gscale = grad_scale * loss_mult * (normalize ? (n > 0 ? 1.0f / (float)n : 0.f) : 1.0f);
  float acc = 0.f;
  for (int d = d0 + threadIdx.x; d < d1; d += LOSS_THREADS) {
    const int a = assign[d];
    int cls = 0;
    if (a >= 0 && !gt_crowd[g0 + a]) cls = gt_classes[g0 + a];        // :286-297
    float cw = 1.0f;
    if (class_weights) cw = (cls >= 0 && cls <= num_classes) ? class_weights[cls] : 0.f;
    const float w = weights[d] * cw;                                    // :298-299
    weights[d] = w;
    const float x = pred[d], z = labels[d];
    // sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|))  (:301-302)
    const float l = fmaxf(x, 0.f) - x * z + logf(1.0f + expf(-fabsf(x)));
    acc += w * l;
    const float sig = 1.0f / (1.0f + expf(-x));
    d_logits[d] = gscale * w * (sig - z);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = LOSS_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[2 * img] = red[0];                                  // cls_loss_unnormed (:304-305)
    loss[2 * img + 1] = n > 0 ? red[0] / (float)n : 0.f;     // cls_loss_normed   (:307-308)
  }
}

struct MatchWs {
  int* order; unsigned long long* k1; unsigned long long* k2; int* ncand; int* cfirst;
};
size_t match_ws_bytes(int n_det) {
  const size_t n = (size_t)n_det + 64;
  return ((n * 4 + 255) & ~(size_t)255) * 3 + ((n * 8 + 255) & ~(size_t)255) * 2;
}
MatchWs carve_match(void* ws, int n_det) {
  const size_t n = (size_t)n_det + 64;
  const size_t s4 = (n * 4 + 255) & ~(size_t)255, s8 = (n * 8 + 255) & ~(size_t)255;
  char* p = (char*)ws;
  MatchWs w;
  w.k1 = (unsigned long long*)p; p += s8;
  w.k2 = (unsigned long long*)p; p += s8;
  w.order = (int*)p; p += s4;
  w.ncand = (int*)p; p += s4;
  w.cfirst = (int*)p;
  return w;
}

int run_matching(const float* iou, const long long* anno_off, const int* det_off, const int* gt_off,
                 const unsigned char* ignore, const float* score, int n_det, int n_gt, int n_img, void* ws, float* labels,
                 float* weights, int* assign, bool have_cand, hipStream_t s, void* prof = nullptr) {
  if (n_gt > kMaxGt) return GNET_ERR_UNSUPPORTED;
  const int cap = (max(n_gt, 1) + 63) & ~63;
  const size_t lds = (size_t)cap * 5;
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)match_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kMaxGt * 5)));
  const MatchWs w = carve_match(ws, n_det);
  if (!have_cand)
    anno_rows<false, true><<<(n_det + 3) / 4, 256, 0, s>>>(nullptr, nullptr, det_off, nullptr, ignore, nullptr, gt_off, anno_off, n_det, n_img, 0,
                                                            const_cast<float*>(iou), w.k1, w.k2, w.ncand, w.cfirst);
  GNET_LAUNCH(prof, GNET_K_LOSS, s, match_rank<<<(n_det + RANK_DETS - 1) / RANK_DETS, 256, 0, s>>>(score, det_off, n_det, n_img, w.order));
  GNET_LAUNCH(prof, GNET_K_LOSS, s, match_greedy<<<n_img, 64, lds, s>>>(iou, anno_off, det_off, gt_off, ignore, w.order, w.k1, w.k2, w.ncand, w.cfirst,
                                                                        labels, weights, assign, cap));
  return launch_status();
}

}  // namespace

extern "C" size_t det_matching_workspace_bytes(int32_t n_det, int32_t n_gt) {
  (void)n_gt;
  if (n_det < 0) return 0;
  return match_ws_bytes(n_det) + 1024;   // + the single-image offset tables
}

extern "C" int det_matching_f32(const float* iou, const float* score, const uint8_t* ignore, int32_t n_det,
                                int32_t n_gt, float* labels, float* weights, int32_t* assignment, void* workspace,
                                size_t workspace_bytes, gnet_stream_t stream) {
  clear_hip_error();
  // shape checks of det_matching.cc:76-93 are the caller's tensor ranks; sizes must be consistent
  if (n_det < 0 || n_gt < 0) return GNET_ERR_INVALID;
  if (n_det == 0) return GNET_OK;
  if (!score || !labels || !weights || !assignment || !workspace) return GNET_ERR_INVALID;
  if (n_gt > 0 && (!iou || !ignore)) return GNET_ERR_INVALID;
  if (n_gt > kMaxGt) return GNET_ERR_UNSUPPORTED;
  if (workspace_bytes < det_matching_workspace_bytes(n_det, n_gt)) return GNET_ERR_WORKSPACE;
  if (((uintptr_t)workspace & 255) != 0) return GNET_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  // single image: offset tables live at the head of the workspace
  struct { long long anno[2]; int det[2]; int gt[2]; } h;
  h.anno[0] = 0; h.anno[1] = (long long)n_det * n_gt; h.det[0] = 0; h.det[1] = n_det; h.gt[0] = 0; h.gt[1] = n_gt;
  HIP_CHECK_RET(hipMemcpyAsync(workspace, &h, sizeof(h), hipMemcpyHostToDevice, s));
  HIP_CHECK_RET(hipStreamSynchronize(s));   // h is a stack object
  const long long* anno_off = (const long long*)workspace;
  const int* det_off = (const int*)((char*)workspace + 16);
  const int* gt_off = (const int*)((char*)workspace + 24);
  return run_matching(iou, anno_off, det_off, gt_off, ignore, score, n_det, n_gt, 1, (char*)workspace + 1024, labels, weights,
                      assignment, false, s);
}

namespace {
int check_loss_args(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in, const gnet_buffers* buf) {
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !in || !buf || !buf->labels || !buf->match_ws) return GNET_ERR_INVALID;
  if (shape->n_det == 0) return GNET_OK;
  if (!in->gt_off || !in->anno_off || !in->det_off) return GNET_ERR_INVALID;
  if (shape->n_gt > 0 && (!in->gt_boxes || !in->gt_crowd || !in->gt_classes)) return GNET_ERR_INVALID;
  if (shape->n_gt > kMaxGt) return GNET_ERR_UNSUPPORTED;      // the matching keeps per-box state in LDS (match_greedy)
  return GNET_OK;
}
}  // namespace

extern "C" int gnet_match_prepare(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                                  gnet_buffers* buf, gnet_stream_t stream) {
  clear_hip_error();
  const int st = check_loss_args(cfg, shape, in, buf);
  if (st != GNET_OK || shape->n_det == 0) return st;
  hipStream_t s = (hipStream_t)stream;
  const int N = shape->n_det;
  const MatchWs w = carve_match((char*)buf->match_ws + 1024, N);
  // det_anno_iou and the candidate records in ONE pass (one wave per detection; an image without ground truth has
  // empty rows and gets the "no candidate" records)
  GNET_LAUNCH(buf->profiler, GNET_K_LOSS, s,
              (anno_rows<true, true><<<(N + 3) / 4, 256, 0, s>>>((const float4*)in->dets, in->det_classes, in->det_off, (const float4*)in->gt_boxes,
                                                                 in->gt_crowd, in->gt_classes, in->gt_off, (const long long*)in->anno_off, N,
                                                                 shape->n_img, cfg->num_classes > 1, buf->det_anno_iou, w.k1, w.k2, w.ncand, w.cfirst)));
  return launch_status();
}

extern "C" int gnet_loss(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                         const float* class_weights, float grad_scale, gnet_buffers* buf, int32_t prepared,
                         gnet_stream_t stream) {
  clear_hip_error();
  int st = check_loss_args(cfg, shape, in, buf);
  if (st != GNET_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  if (shape->n_det == 0) {                 // no detection: the per-image losses are zero (not what an earlier step left there)
    if (buf->loss && shape->n_img > 0) HIP_CHECK_RET(hipMemsetAsync(buf->loss, 0, (size_t)2 * shape->n_img * sizeof(float), s));
    return GNET_OK;
  }
  const int N = shape->n_det;
  if (!prepared) {
    st = gnet_match_prepare(cfg, shape, in, buf, stream);
    if (st != GNET_OK) return st;
  }
  st = run_matching(buf->det_anno_iou, (const long long*)in->anno_off, in->det_off, in->gt_off, in->gt_crowd,
                    buf->prediction, N, shape->n_gt, shape->n_img, (char*)buf->match_ws + 1024, buf->labels, buf->weights,
                    buf->det_gt_matching, true, s, buf->profiler);
  if (st != GNET_OK) return st;
  GNET_LAUNCH(buf->profiler, GNET_K_LOSS, s,
              loss_kernel<<<shape->n_img, LOSS_THREADS, 0, s>>>(buf->prediction, buf->labels, buf->weights, buf->det_gt_matching,
                                                                in->det_off, in->gt_off, in->gt_crowd, in->gt_classes, class_weights,
                                                                cfg->num_classes, cfg->normalize_loss, cfg->loss_multiplyer, grad_scale,
                                                                buf->loss, buf->d_logits));
  return launch_status();
}

// Plain box IoU [n_a_i, n_b_i] per image (network.py:475-481, no crowd columns, no class mask): Gnet.det_det_iou on demand.
extern "C" int gnet_box_iou(const float* a_boxes, int32_t n_a, const float* b_boxes, int32_t n_b, const int32_t* a_off,
                            const int32_t* b_off, const int64_t* out_off, int32_t n_img, float* out, gnet_stream_t stream) {
  clear_hip_error();
  if (n_a < 0 || n_b < 0 || n_img < 1) return GNET_ERR_INVALID;
  if (n_a == 0 || n_b == 0) return GNET_OK;
  if (!a_boxes || !b_boxes || !a_off || !b_off || !out_off || !out) return GNET_ERR_INVALID;
  anno_rows<true, false><<<(n_a + 3) / 4, 256, 0, (hipStream_t)stream>>>((const float4*)a_boxes, nullptr, a_off, (const float4*)b_boxes, nullptr,
                                                                          nullptr, b_off, (const long long*)out_off, n_a, n_img, 0, out,
                                                                          nullptr, nullptr, nullptr, nullptr);
  return launch_status();
}
