// Workspace carving (host code): gnet_param_count / gnet_workspace_bytes / gnet_plan.
#include <string.h>
#include "common.hpp"

namespace {

struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Upper bounds on the number of workgroups that write partial weight gradients (see backward.hip).
size_t arena_floats(const gnet_config* cfg, const gnet_shape* sh) {
  const ParamLayout L = make_layout(cfg);
  (void)sh;
  // every backward kernel writes at most kMaxPartials partial copies of the parameters it owns
  return (size_t)GNET_ARENA_PARTIALS * (size_t)arena_stride(L.total);
}

size_t carve(const gnet_config* cfg, const gnet_shape* sh, int training, void* ws, gnet_buffers* out) {
  const ParamLayout L = make_layout(cfg);
  const size_t N = (size_t)sh->n_det, E = (size_t)sh->n_edge, B = (size_t)cfg->num_blocks;
  const size_t Np = N + 32, Ep = E + 64;   // slack for tile tails
  gnet_buffers b;
  memset(&b, 0, sizeof(b));
  Carver c(ws);
  b.row_ptr = c.take<int32_t>(N + 1);
  b.edge_c = c.take<int32_t>(Ep);
  b.edge_n = c.take<int32_t>(Ep);
  b.edge_iou = c.take<float>(Ep);
  b.edge_t = c.take<int32_t>(Ep);
  b.edge_nz = c.take<int32_t>(Ep);
  b.pw_feats = c.take<float>(Ep * D_E);
  b.packed_t = c.take<float>((size_t)packed_floats(L, cfg->num_blocks));
  b.prediction = c.take<float>(Np);
  b.scratch_i = c.take<int32_t>(N + 1024);
  b.geo = c.take<float>(Ep * 8);
  // (num_pwfeat_fc = 0: no pw-MLP -- its tables, activations and gradients are not carved; pw_feats holds the geometry columns
  // padded to 32, the blocks' rn tables carry one more row per detection: its neighbour score term alone, read by its self pair)
  const size_t rn_rows = L.raw ? 2 * N + 34 : Np;
  if (!L.raw) {
    b.pw_tc = c.take<float>(Np * D_H);
    b.pw_tn = c.take<float>(Np * D_H);
  }
  if (training) {
    if (!L.raw) {
      b.pw_h1 = c.take<float>(Ep * D_H);
      b.pw_h2 = c.take<float>(Ep * D_H);
    }
    b.block_feats[0] = nullptr;
    for (size_t k = 1; k <= B; ++k) {
      b.block_feats[k] = c.take<float>(Np * D_S);
      b.blk_r[k] = c.take<float>(Np * D_R);
      b.blk_rc[k] = c.take<float>(Np * D_P);
      b.blk_rn[k] = c.take<float>(rn_rows * D_P);
      b.blk_pm[k] = c.take<uint64_t>(2 * Np * D_P);      // pm, then parg (one memset clears both)
      b.blk_parg[k] = b.blk_pm[k] + Np * D_P;
      b.blk_q[k] = c.take<float>(Np * D_P);
      if (cfg->neighbor_feats) b.blk_rnb[k] = c.take<float>(Np * D_R);
      if (training == 2) { b.blk_h1[k] = c.take<float>(Ep * D_P); b.blk_h2[k] = c.take<float>(Ep * D_P); }   // tests / debugging only
    }
    b.head1 = c.take<float>(Np * D_HEAD);
    b.head2 = c.take<float>(Np * D_HEAD);
    b.det_anno_iou = c.take<float>((size_t)sh->n_anno + 64);
    b.labels = c.take<float>(Np);
    b.weights = c.take<float>(Np);
    b.det_gt_matching = c.take<int32_t>(Np);
    b.loss = c.take<float>((size_t)sh->n_img * 2 + 2);
    b.match_ws_bytes = det_matching_workspace_bytes(sh->n_det, sh->n_gt);
    b.match_ws = c.take<char>(b.match_ws_bytes);
    b.d_logits = c.take<float>(Np);
    b.d_x = c.take<float>(Np * D_S);
    b.d_pc = c.take<float>(Np * D_P);
    b.d_rc = c.take<float>(Np * D_P);
    b.d_rn = c.take<float>(Np * D_P);
    b.d_pw = c.take<float>(Ep * D_E);
    if (!L.raw) b.d_h1 = c.take<float>(Ep * D_H);
    b.d_g1 = c.take<float>(Ep * D_P);
    {
      const EdgeGeom G = edge_geom((int64_t)E, (int64_t)N);
      b.ewin = c.take<uint64_t>((B + 1) * G.bm_stride);
      b.wprefix = c.take<int32_t>((B + 1) * G.bm_stride);
      b.wlist = c.take<int32_t>(B * G.wl_stride);
      b.xmask = c.take<uint64_t>(B * G.xm_stride);
      b.tflag = c.take<uint8_t>(B * G.tf_stride);
      b.apos = c.take<int32_t>(B * G.ap_stride);
      b.tpos = c.take<int32_t>(B * G.wl_stride);
      b.wrow = c.take<int32_t>(B * G.tf_stride);
      if (L.raw) b.spos = c.take<int32_t>(B * G.tf_stride);
      b.rl_scratch = c.take<int32_t>((B + 1) * (2 * G.n_wg + 1) + GNET_MAX_BLOCKS + 64);   // + per-block tie-list counters
    }
    b.pw_rows = c.take<int32_t>(Ep);
    if (!L.raw) {
      b.w1_s = c.take<float>(Np * D_H);
      b.w1_t = c.take<float>(Np * D_H);
    }
    b.arena_floats = arena_floats(cfg, sh);
    b.arena = c.take<float>(b.arena_floats);
  } else {
    // inference: per-block tensors are transient -> two alternating sets
    float* xf[2] = {c.take<float>(Np * D_S), c.take<float>(Np * D_S)};
    float* rc = c.take<float>(Np * D_P);
    float* rn = c.take<float>(rn_rows * D_P);
    uint64_t* pm = c.take<uint64_t>(Np * D_P);
    for (size_t k = 1; k <= B; ++k) {
      b.block_feats[k] = xf[k & 1];
      b.blk_rc[k] = rc; b.blk_rn[k] = rn; b.blk_pm[k] = pm;
    }
  }
  if (out) *out = b;
  return (c.off + 255) & ~(size_t)255;
}

}  // namespace

extern "C" int64_t gnet_param_count(const gnet_config* cfg) {
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  return make_layout(cfg).total;
}

extern "C" size_t gnet_workspace_bytes(const gnet_config* cfg, const gnet_shape* shape, int training) {
  if (!config_supported(cfg) || !shape || shape->n_det < 0 || shape->n_edge < 0 || shape->n_img < 1) return 0;
  return carve(cfg, shape, training, nullptr, nullptr);
}

extern "C" int gnet_plan(const gnet_config* cfg, const gnet_shape* shape, int training, void* workspace,
                         size_t workspace_bytes, gnet_buffers* out) {
  if (!config_supported(cfg)) return GNET_ERR_UNSUPPORTED;
  if (!shape || !workspace || !out || shape->n_det < 0 || shape->n_edge < 0 || shape->n_img < 1) return GNET_ERR_INVALID;
  if (((uintptr_t)workspace & 255) != 0) return GNET_ERR_INVALID;
  const size_t need = carve(cfg, shape, training, nullptr, nullptr);
  if (workspace_bytes < need) return GNET_ERR_WORKSPACE;
  carve(cfg, shape, training, workspace, out);
  return GNET_OK;
}

extern "C" int gnet_profiler_create(int32_t capacity, uint32_t class_mask, void** out) {
  if (capacity <= 0 || !out) return GNET_ERR_INVALID;
  GnetProfiler* p = new GnetProfiler();
  p->mask = class_mask; p->cap = capacity; p->n = 0; p->stride = 1;
  memset(p->seen, 0, sizeof(p->seen));
  p->ev0 = new hipEvent_t[capacity]; p->ev1 = new hipEvent_t[capacity]; p->cls = new int[capacity];
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&p->ev0[i]) != hipSuccess || hipEventCreate(&p->ev1[i]) != hipSuccess) return GNET_ERR_HIP;
  }
  *out = p;
  return GNET_OK;
}

extern "C" int gnet_profiler_read(void* profiler, double* ms_sum, int32_t* count) {
  GnetProfiler* p = (GnetProfiler*)profiler;
  if (!p || !ms_sum || !count) return GNET_ERR_INVALID;
  for (int i = 0; i < p->n; ++i) {
    if (hipEventSynchronize(p->ev1[i]) != hipSuccess) return GNET_ERR_HIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p->ev0[i], p->ev1[i]) != hipSuccess) return GNET_ERR_HIP;
    ms_sum[p->cls[i]] += ms;
    count[p->cls[i]] += 1;
  }
  p->n = 0;
  memset(p->seen, 0, sizeof(p->seen));
  return GNET_OK;
}

// Brackets for launches whose entry points carry no gnet_buffers (the graph build: gnet_graph_count / _fill / _transpose take
// plain pointers): the caller opens a scope in front of the call and closes it behind it, on the stream it passes to the call.
extern "C" int gnet_profiler_begin(void* profiler, int32_t cls, gnet_stream_t stream) {
  GnetProfiler* p = (GnetProfiler*)profiler;
  if (!p || cls < 0 || cls >= GNET_KCLASS_COUNT) return -1;
  if (!((p->mask >> cls) & 1u) || (p->seen[cls]++ % p->stride) != 0 || p->n >= p->cap) return -1;
  const int idx = p->n++;
  p->cls[idx] = cls;
  (void)hipEventRecord(p->ev0[idx], (hipStream_t)stream);
  return idx;
}

extern "C" int gnet_profiler_end(void* profiler, int32_t idx, gnet_stream_t stream) {
  GnetProfiler* p = (GnetProfiler*)profiler;
  if (!p || idx < 0 || idx >= p->n) return GNET_OK;
  return hipEventRecord(p->ev1[idx], (hipStream_t)stream) == hipSuccess ? GNET_OK : GNET_ERR_HIP;
}

extern "C" int gnet_profiler_set_stride(void* profiler, int32_t stride) {
  GnetProfiler* p = (GnetProfiler*)profiler;
  if (!p || stride < 1) return GNET_ERR_INVALID;
  p->stride = stride;
  return GNET_OK;
}

extern "C" int gnet_profiler_destroy(void* profiler) {
  GnetProfiler* p = (GnetProfiler*)profiler;
  if (!p) return GNET_OK;
  for (int i = 0; i < p->cap; ++i) { (void)hipEventDestroy(p->ev0[i]); (void)hipEventDestroy(p->ev1[i]); }
  delete[] p->ev0; delete[] p->ev1; delete[] p->cls;
  delete p;
  return GNET_OK;
}

extern "C" const char* gnet_version(void) { return "gossipnet_hip 0.5 (gfx950, fp32 MFMA)"; }

extern "C" int gnet_abi_version(void) { return GNET_ABI_VERSION; }

extern "C" int gnet_abi_sizes(size_t out[8]) {
  if (!out) return GNET_ERR_INVALID;
  out[0] = sizeof(gnet_config);
  out[1] = sizeof(gnet_shape);
  out[2] = sizeof(gnet_inputs);
  out[3] = sizeof(gnet_buffers);
  out[4] = offsetof(gnet_buffers, head1);
  out[5] = offsetof(gnet_buffers, d_g1);
  out[6] = offsetof(gnet_buffers, match_ws_bytes);
  out[7] = offsetof(gnet_buffers, start_feat);
  return GNET_KCLASS_COUNT;
}
