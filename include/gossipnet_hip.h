/* gossipnet_hip.h -- C ABI of libgossipnet_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the GossipNet hot path.  Every entry point replaces a
 * reference interface (paths relative to the reference repo root):
 *
 *   gnet_*                 nms_net/network.py:121-322  class Gnet (graph build,
 *                          pairwise features, pw-MLP, _block stack, head, loss,
 *                          and TF autodiff of those = gnet_backward)
 *   det_matching_f32       nms_net/matching_module/det_matching.cc:16-33,72-165
 *                          (op "DetectionMatching", loaded at matching_module/__init__.py:9-13)
 *   roi_pool_fwd_f32       nms_net/roi_pooling_layer/roi_pooling_op.cc:35-43,79-195
 *                          (op "RoiPool";  GPU twin roi_pooling_op_gpu.cu:19-110)
 *   roi_pool_bwd_f32       nms_net/roi_pooling_layer/roi_pooling_op.cc:45-54,324-457
 *                          (op "RoiPoolGrad"; GPU twin roi_pooling_op_gpu.cu:113-215)
 *
 * Conventions: plain C, raw DEVICE pointers + explicit sizes, caller-owned
 * buffers and workspace (query the size first), the HIP stream is passed in as
 * an opaque pointer (hipStream_t), every call returns an int status (0 = ok)
 * and never exits the process (unlike roi_pooling_op_gpu.cu:102-107), no global
 * mutable state, re-entrant per stream.  All tensors are dense row-major fp32
 * unless stated.  Indices are int32.
 */
#ifndef GOSSIPNET_HIP_H
#define GOSSIPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNET_OK 0
#define GNET_ERR_INVALID (-1)     /* bad argument (the TF ops raise InvalidArgument)      */
#define GNET_ERR_UNSUPPORTED (-2) /* a hyper-parameter outside the compiled configuration */
#define GNET_ERR_WORKSPACE (-3)   /* workspace too small                                  */
#define GNET_ERR_HIP (-4)         /* a HIP runtime call or launch failed                  */

typedef void* gnet_stream_t; /* hipStream_t */

/* Hyper-parameters: the cfg.gnet.* / cfg.train.* values Gnet reads at construction
 * (nms_net/config.py:46-79, experiments/<exp>/conf.yaml).  Compiled layer widths: shortcut 128, reduced 32, pairfeat 64,
 * predict_fc 128, num_predict_fc 3, num_block_pw_fc 2, num_block_fc 2, and the pairwise-feature MLP either as in the two
 * shipped experiments (num_pwfeat_fc 3, pwfeat 256, pwfeat_narrow 32) or ABSENT (num_pwfeat_fc 0, the reference's default,
 * config.py:73 / network.py:217-221: the blocks' pw_fc1 then reads the 2C'+7 raw _geometry_feats columns; pwfeat_dim and
 * pwfeat_narrow_dim are ignored).  Anything else: GNET_ERR_UNSUPPORTED.  num_classes, num_blocks, neighbor_thresh,
 * neighbor_feats, pw_feat_multiplyer and the loss flags are run-time values; the image-feature variant (cfg.gnet.imfeats) is
 * selected per call through gnet_buffers.start_feat. */
typedef struct gnet_config {
  int32_t num_classes;     /* C; multiclass = C > 1 (network.py:151)            */
  int32_t num_blocks;      /* cfg.gnet.num_blocks                                */
  float neighbor_thresh;   /* cfg.gnet.neighbor_thresh (0.2)                     */
  int32_t normalize_loss;  /* cfg.train.normalize_loss                           */
  float loss_multiplyer;   /* cfg.train.loss_multiplyer                          */
  int32_t shortcut_dim, reduced_dim, pairfeat_dim, pwfeat_dim, pwfeat_narrow_dim;
  int32_t num_pwfeat_fc, predict_fc_dim, num_predict_fc, num_block_pw_fc, num_block_fc;
  float pw_feat_multiplyer; /* cfg.gnet.pw_feat_multiplyer: factor on every _geometry_feats column (network.py:199-200) */
  int32_t neighbor_feats;   /* cfg.gnet.neighbor_feats: a second reduce FC `reduce_dim_neighbor` feeds the neighbour half of
                               build_context (network.py:356-365); its variables follow fc2 in every block's group */
} gnet_config;

/* Sizes of one batch: n_img images concatenated (the reference runs n_img = 1,
 * train.py:115; a batch is a block-diagonal graph, images never interact).
 * n_anno = sum_i n_det_i * n_gt_i (ragged det_anno_iou). */
typedef struct gnet_shape {
  int32_t n_img, n_det, n_gt;
  int64_t n_edge;
  int64_t n_anno;
} gnet_shape;

/* Inputs = Gnet.get_batch_spec (network.py:131-146), concatenated over images.
 * Precondition (the reference's data path checks the same: imdb/tools.py:99-110 validate_boxes): finite coordinates and scores,
 * x2 > x1 and y2 > y1.  A zero-width or zero-height detection makes log(n_w / c_w) infinite; the FC kernels form their fp32
 * products from exact three-term bf16 splits, and the split of +-inf is inf - inf = NaN -- the outputs of that image are
 * then NaN (no fault). */
typedef struct gnet_inputs {
  const float* dets;          /* [n_det,4] xyxy                        */
  const float* det_scores;    /* [n_det]                               */
  const int32_t* det_classes; /* [n_det] 1-based                       */
  const int32_t* det_off;     /* [n_img+1] first detection of image i  */
  const float* gt_boxes;      /* [n_gt,4]   (training; may be NULL)    */
  const uint8_t* gt_crowd;    /* [n_gt] bool                           */
  const int32_t* gt_classes;  /* [n_gt]                                */
  const int32_t* gt_off;      /* [n_img+1]                             */
  const int64_t* anno_off;    /* [n_img+1] offset of image i in det_anno_iou */
} gnet_inputs;

#define GNET_MAX_BLOCKS 64

/* Device pointers carved out of the caller's workspace by gnet_plan().  These are
 * the tensors the reference exposes as Gnet attributes (network.py:170-313) plus
 * what the backward pass re-reads. */
typedef struct gnet_buffers {
  /* graph (network.py:192-195): CSR of neighbor_pair_idxs, row-major order */
  int32_t* row_ptr;   /* [n_det+1]                     */
  int32_t* edge_c;    /* [n_edge] pair_c_idxs          */
  int32_t* edge_n;    /* [n_edge] pair_n_idxs          */
  float* edge_iou;    /* [n_edge] det_det_iou at pairs */
  int32_t* edge_t;    /* [n_edge] index of the reversed pair (n,c); the graph is symmetric */
  int32_t* edge_nz;   /* [n_edge+64] neighbour index for the rn gathers: n_det (a zero row) for self pairs and the tail */
  float* geo;         /* [n_edge,8] 7 geometry columns of _geometry_feats (+pad) */
  float* pw_tc;       /* [n_det,256] per detection: score x (pw_feats/fc1 row of its class's CENTRE score column) + bias -- the one-hot x score columns of _geometry_feats (network.py:413-419) contribute one row of fc1 per detection, not per edge */
  float* pw_tn;       /* [n_det,256] per detection: score x (fc1 row of its class's NEIGHBOUR score column) */
  float* pw_h1;       /* [n_edge,256]  pw_feats/fc1 output (training)            */
  float* pw_h2;       /* [n_edge,256]  pw_feats/fc2 output (training)            */
  float* pw_feats;    /* [n_edge,32]   Gnet.pw_feats                             */
  float* block_feats[GNET_MAX_BLOCKS + 1]; /* [n_det,128] each; [0] = zeros      */
  float* blk_r[GNET_MAX_BLOCKS + 1];       /* [n_det,32]  relu(reduce_dim)        */
  float* blk_rc[GNET_MAX_BLOCKS + 1];      /* [n_det,64]  r.W1[32:64] + b1        */
  float* blk_rn[GNET_MAX_BLOCKS + 1];      /* [n_det+1,64] r.W1[64:96], row n_det = zeros (self pairs); with num_pwfeat_fc = 0: [2 n_det+2,64], row n_det+1+i = detection i's neighbour score term alone (its self pair) */
  uint64_t* blk_pm[GNET_MAX_BLOCKS + 1];   /* [n_det,64]  (segment max bits<<32)|tie count (the count is formed by a training forward only; a forward-only pass leaves a meaningless low word) */
  float* blk_q[GNET_MAX_BLOCKS + 1];       /* [n_det,64]  relu(fc1)               */
  float* blk_rnb[GNET_MAX_BLOCKS + 1];     /* [n_det,32]  relu(reduce_dim_neighbor) (neighbor_feats, training)   */
  float* blk_h1[GNET_MAX_BLOCKS + 1];      /* [n_edge+64,64] relu(pw_fc1) per edge -- only when planned with training == 2 (tests / debugging: the backward pass recomputes the rows it needs) */
  float* blk_h2[GNET_MAX_BLOCKS + 1];      /* [n_edge+64,64] pw_fc2 pre-activation (h1.W2 + b2) per edge, the bits the segment maximum was taken on -- only when planned with training == 2 (tests: the winner records are checked against it exactly) */
  uint64_t* blk_parg[GNET_MAX_BLOCKS + 1]; /* [n_det,64] (segment-max bits << 32) | index of the first edge that attains it (training) */
  float* head1;       /* [n_det,128] predict/fc1 */
  float* head2;       /* [n_det,128] predict/fc2 */
  float* prediction;  /* [n_det] logits (Gnet.prediction) */
  /* loss (network.py:275-313) */
  float* det_anno_iou;      /* [n_anno] ragged [n_det_i, n_gt_i] per image */
  float* labels;            /* [n_det] */
  float* weights;           /* [n_det] after class weighting */
  int32_t* det_gt_matching; /* [n_det] */
  float* loss;              /* [n_img,2]: (loss_unnormed, loss_normed) per image */
  /* backward scratch */
  float* d_logits;    /* [n_det] */
  float* d_x;         /* [n_det,128] */
  float* d_pc;        /* [n_det,64]  dp / tie count */
  float* d_rc;        /* [n_det,64] */
  float* d_rn;        /* [n_det,64] */
  float* d_pw;        /* [n_edge,32] grad wrt pw_feats */
  float* d_h1;        /* [n_edge,256] grad wrt pre-activation of pw_feats/fc1 */
  float* d_g1;        /* [n_edge+64,64] grad wrt pre-activation of block pw_fc1 on the block's winner edges, list order (per block, transient) */
  /* sparse SegmentMax backward (csrc/backward_edge.hip): per block b = 1..num_blocks at index b-1, strides = edge_geom() */
  uint64_t* ewin;     /* [num_blocks+1][bm_stride] 1 bit per edge: the edge attains a positive segment maximum of the block ("winner": gradient flows through it); last map = OR over the blocks */
  int32_t* wprefix;   /* [num_blocks+1][bm_stride] winners before 64-edge word w */
  int32_t* wlist;     /* [num_blocks][wl_stride] ascending winner edges of the block */
  uint64_t* xmask;    /* [num_blocks][xm_stride] per edge: columns whose (tied) maximum it attains besides the recorded arg-max edge; valid on rows of flagged detections only */
  uint8_t* tflag;     /* [num_blocks][tf_stride] the detection has a tied positive maximum in this block */
  int32_t* apos;      /* [num_blocks][n_det+32,64] per (detection, column): bits 0-23 = list position of the arg-max edge + 1 (0 = no gradient), bit 30 = the detection's tie flag */
  int32_t* tpos;      /* [num_blocks][n_edge+64] list position of every edge's REVERSED pair in the block's winner list; -1 = not a winner (or a self pair) */
  int32_t* wrow;      /* [num_blocks][n_det+32] list position of the first winner of every detection's edge range (CSR row pointers of the winner lists) */
  int32_t* spos;      /* [num_blocks][n_det+32] num_pwfeat_fc = 0 only: list position of every detection's SELF pair in the block's winner list, -1 = not a winner (its neighbour SCORE column receives gradient although its neighbour features are zeroed, network.py:371-374) */
  int32_t* rl_scratch;/* scan scratch of the list construction; also holds the list lengths */
  int32_t* pw_rows;   /* [n_edge+64] ascending indices of the edges with a non-zero d_pw row (rows of the pw-MLP backward) */
  float* w1_s;        /* [n_det,256] sum of d_h1 over the detection's own pairs (centre role)      */
  float* w1_t;        /* [n_det,256] sum of d_h1 over the reversed pairs (neighbour role)          */
  float* packed_t;    /* [param_count (+ num_blocks x 64 x 96 with num_pwfeat_fc = 0)] transposed copies of the weight matrices */
  float* arena;       /* per-workgroup partial weight gradients */
  int32_t* scratch_i; /* [n_det + 1024] per detection: 1 = its segment-max records start from zero in every block (no edge, or its edges are split between two waves' ranges of the forward edge kernel), written once per step by gnet_forward (no backward kernel touches scratch_i).  Two gnet_forward calls in flight at the same time must not share one planned workspace. */
  void* match_ws;     /* det_matching_workspace_bytes(n_det, n_gt) */
  size_t match_ws_bytes;
  size_t arena_floats;
  void* profiler;     /* optional gnet_profiler (NULL = off); set by the caller after gnet_plan */
  const float* start_feat; /* optional [n_det,128] block_feats[0] (image-feature variant, network.py:223-240); NULL = zeros
                              (network.py:241-246).  Set by the caller after gnet_plan; gnet_backward then leaves the
                              gradient wrt it in d_x. */
} gnet_buffers;

/* ---- parameters ------------------------------------------------------------
 * One flat fp32 buffer in TF-variable order, weights [in,out]:
 *   gnet/pw_feats/fc{1,2,3}/{weights,biases}                  (absent with num_pwfeat_fc = 0; pw_fc1 is then [2C'+7+64, 64])
 *   gnet/block{k}/{reduce_dim,pw_fc1,pw_fc2,fc1,fc2[,reduce_dim_neighbor]}/{weights,biases}   k = 1..B
 *   gnet/predict/fc{1,2}/fully_connected/{weights,biases}
 *   gnet/predict/logits/fully_connected/{weights,biases}
 * gnet_param_count returns <0 on an unsupported config. */
int64_t gnet_param_count(const gnet_config* cfg);

/* ---- graph build (network.py:170-176,192-195) --------------------------------
 * Pass 1: per-row neighbour counts + exclusive scan -> row_ptr[n_det+1]
 * (row_ptr[n_det] = n_edge, read it back to size the edge buffers).
 * scratch: at least (n_det + 1024) int32. */
int gnet_graph_count(const float* dets, int32_t n_det, const int32_t* det_off, int32_t n_img,
                     float thresh, int32_t* row_ptr, int32_t* scratch, gnet_stream_t stream);
/* Pass 2: ordered fill of edge_c / edge_n / edge_iou (row-major = tf.where order). */
int gnet_graph_fill(const float* dets, int32_t n_det, const int32_t* det_off, int32_t n_img,
                    float thresh, const int32_t* row_ptr, int32_t* edge_c, int32_t* edge_n,
                    float* edge_iou, gnet_stream_t stream);

/* Reverse-edge permutation: edge_t[e] = position of (n,c) for e = (c,n) (binary search in row n). */
int gnet_graph_transpose(const int32_t* row_ptr, const int32_t* edge_c, const int32_t* edge_n, int64_t n_edge,
                         int32_t* edge_t, gnet_stream_t stream);

/* Dense box IoU per image (network.py:475-481; no crowd columns, no class mask): out + out_off[i] = [n_a_i, n_b_i] row-major.
 * Gnet.det_det_iou (network.py:176) on demand -- the hot path never materialises N x N. */
int gnet_box_iou(const float* a_boxes, int32_t n_a, const float* b_boxes, int32_t n_b, const int32_t* a_off,
                 const int32_t* b_off, const int64_t* out_off, int32_t n_img, float* out, gnet_stream_t stream);

/* ---- workspace -----------------------------------------------------------------
 * training: 0 = inference (two alternating sets of per-block tensors), 1 = training (everything the backward
 * pass re-reads), 2 = training + keep the per-block pw_fc1 activations blk_h1 (tests / debugging only). */
size_t gnet_workspace_bytes(const gnet_config* cfg, const gnet_shape* shape, int training);
int gnet_plan(const gnet_config* cfg, const gnet_shape* shape, int training, void* workspace,
              size_t workspace_bytes, gnet_buffers* out);

/* ---- forward: features + pw-MLP + blocks + head -> buf->prediction
 * (network.py:197-273).  buf->row_ptr/edge_* must have been filled. */
int gnet_forward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                 const float* params, gnet_buffers* buf, int training, gnet_stream_t stream);

/* ---- loss: det_anno_iou, detection_matching, class weighting, sigmoid x-ent
 * (network.py:174-187,275-313); also seeds d_logits = grad_scale * dloss/dlogit.
 * class_weights: [num_classes+1] device, or NULL for ones (network.py:282-284). */
int gnet_loss(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
              const float* class_weights, float grad_scale, gnet_buffers* buf, int32_t prepared,
              gnet_stream_t stream);
/* Limit: shape->n_gt (all images of the step together) <= 24576 -- the greedy matching keeps 5 bytes of state per
 * ground-truth box in LDS, sized by the one count the host knows; GNET_ERR_UNSUPPORTED beyond (det_matching_f32 alike). */

/* The score-independent half of gnet_loss -- det_anno_iou (network.py:174-187) and the per-detection
 * candidate keys of the matching (det_matching.cc:128-148) -- depends on the inputs only: a caller may
 * run it on another stream while gnet_forward runs and then pass prepared = 1 to gnet_loss (after
 * ordering the two streams).  With prepared = 0 gnet_loss does this work itself. */
int gnet_match_prepare(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                       gnet_buffers* buf, gnet_stream_t stream);

/* ---- backward: d loss / d params -> grads[param_count] (overwritten).
 * Reproducible (bitwise identical on repeated runs): weight gradients are per-workgroup partials summed in index
 * order; the one float atomic in the path -- edge_bwd_w adds a tile's d P rows into d_pw with a returnless
 * atomic add -- touches every address at most once per launch and the 16 block launches are stream-ordered, so
 * each element still receives its additions in a fixed order.  (Integer atomics build the winner bitmaps; the
 * order of the tied-detection list `tlist` is not fixed, and nothing depends on it: winners_ties only sets bits.)
 * The gradient of the SegmentMax (network.py:383-386) reaches one edge per (detection, column); the edge
 * stage and the pw-MLP backward therefore run on the edges that carry gradient only -- the same sums as the
 * dense algorithm minus exact zeros.
 * Limits: n_edge <= 2^24 - 128 (32-bit byte offsets into [E,64] fp32 arrays; GNET_ERR_UNSUPPORTED beyond). */
int gnet_backward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                  const float* params, gnet_buffers* buf, float* grads, int32_t prepared, void* prepared_event,
                  void* positions_event, gnet_stream_t stream);
/* prepared_event (hipEvent_t or NULL, only with prepared = 1): recorded by the caller behind gnet_backward_prepare on
 * its stream.  gnet_backward makes `stream` wait for it where the prepared data is first needed -- after the head's
 * and the last block's node kernels, which need none of it -- instead of the caller waiting before the call.
 * positions_event (hipEvent_t or NULL, only with prepared = 1): for a caller that ran the preparation in two pieces -- phase 2,
 * prepared_event recorded behind it, then phase 3, positions_event behind that.  gnet_backward waits for it in front of the
 * first gather_winners, the first reader of phase 3's output: phase 3 then runs beside the last block's edge kernel instead of
 * in front of it.  NULL: prepared_event covers everything (phases 0 or 2 + 3 in front of it). */

/* The part of gnet_backward that depends on the forward pass only, not on the loss: the SegmentMax winner maps and
 * row lists of every block, and the zeroed d_pw accumulator.  A caller may run it on another stream once
 * gnet_forward has finished, beside gnet_loss, and pass prepared = 1 to gnet_backward (after ordering the two
 * streams).  With prepared = 0 gnet_backward does this work itself. */
int gnet_backward_prepare(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                          const float* params, gnet_buffers* buf, int32_t phase, gnet_stream_t stream);
/* phase 0 = everything (after gnet_forward); 1 = only the fills (d_pw and the winner maps zeroed, tpos set to -1:
 * independent of the forward pass, may run beside it); 2 = the forward-dependent part the edge kernels read (winner maps, row
 * lists, arg-max list positions; after gnet_forward and after phase 1); 3 = the rest of the forward-dependent part (the reversed
 * pairs' list positions, read by gather_winners only; after phase 2).  0 = 1 + 2 + 3. */

/* ---- training step around the path (train.py:64-77: slim create_train_op with Adam / Momentum) --------
 * All buffers are flat fp32 of n = gnet_param_count elements (device).  grad_scale multiplies the
 * gradient on the fly (e.g. 1/world after a sum all-reduce).  t = 1-based step count (Adam bias terms).
 * gnet_clip_by_norm = tf.clip_by_norm of every gradient tensor (clip_gradient_norm > 0), tensors given by
 * n_tensors+1 device offsets into the flat buffer. */
int gnet_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t t, float grad_scale, gnet_stream_t stream);
int gnet_momentum_step(float* params, const float* grads, float* accum, int64_t n, float lr, float momentum,
                       float grad_scale, gnet_stream_t stream);
int gnet_clip_by_norm(float* grads, const int64_t* tensor_offsets, int32_t n_tensors, float clip_norm,
                      gnet_stream_t stream);

/* ---- dense FC layers of the image-feature start features ("reduce_imfeats", network.py:223-240):
 * y = act(x . w + b), x [M,K], w [K,N] ([in,out] as tf.contrib.layers.fully_connected), y [M,N]; K and N multiples of 4.
 * gnet_fc_backward: dy = gradient wrt y (post-activation; the ReLU mask is y > 0); dw [K,N], db [N] overwritten,
 * dx [M,K] optional (NULL: not computed).  fp32 MFMA GEMMs, split-K partials added in a fixed order (reproducible).
 * workspace: gnet_fc_workspace_bytes(M, K, N) for either call. */
size_t gnet_fc_workspace_bytes(int64_t M, int64_t K, int64_t N);
int gnet_fc_forward(const float* x, const float* w, const float* b, int64_t M, int64_t K, int64_t N, int relu,
                    float* y, void* workspace, size_t workspace_bytes, gnet_stream_t stream);
int gnet_fc_backward(const float* x, const float* w, const float* y, const float* dy, int64_t M, int64_t K, int64_t N,
                     int relu, float* dw, float* db, float* dx, void* workspace, size_t workspace_bytes,
                     gnet_stream_t stream);

/* ---- DetectionMatching (det_matching.cc:72-160).  iou [n_det,n_gt], score [n_det],
 * ignore [n_gt] (bool as u8) -> labels, weights (f32 [n_det]), assignment (i32 [n_det]).
 * Ties in score: higher index first; equal ignore flags: lower index first.
 * workspace: det_matching_workspace_bytes(n_det, n_gt). */
size_t det_matching_workspace_bytes(int32_t n_det, int32_t n_gt);
int det_matching_f32(const float* iou, const float* score, const uint8_t* ignore, int32_t n_det,
                     int32_t n_gt, float* labels, float* weights, int32_t* assignment,
                     void* workspace, size_t workspace_bytes, gnet_stream_t stream);

/* ---- RoiPool / RoiPoolGrad, NHWC, argmax = index within the image
 * (roi_pooling_op.cc:128-187, 374-449).  rois [R,5] = (batch, x1, y1, x2, y2). */
int roi_pool_fwd_f32(const float* bottom_data, int32_t B, int32_t H, int32_t W, int32_t C,
                     const float* bottom_rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                     float spatial_scale, float* top_data, int32_t* argmax, gnet_stream_t stream);
int roi_pool_bwd_f32(const float* top_diff, const int32_t* argmax, const float* bottom_rois,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t R, int32_t pooled_h,
                     int32_t pooled_w, float spatial_scale, float* bottom_diff, gnet_stream_t stream);
/* roi_pool_bwd_f32 sums in the CPU kernel's order (ROI index, then bin): bit-exact and reproducible.  The _atomic
 * variant scatters every pooled element's gradient to its arg-max with a float atomic: less traffic (but, since round 3,
 * 3-4x SLOWER than the ordered kernel at the contract shape: 2.3 ms against 0.62 ms), summation order not fixed, and NOT
 * the reference's result where its in-ROI / feasible-bin tests (roi_pooling_op.cc:405-431)
 * drop an element (arg-max pixel one past the rounded ROI end). */
int roi_pool_bwd_atomic_f32(const float* top_diff, const int32_t* argmax, const float* bottom_rois,
                            int32_t B, int32_t H, int32_t W, int32_t C, int32_t R, int32_t pooled_h,
                            int32_t pooled_w, float spatial_scale, float* bottom_diff, gnet_stream_t stream);

/* ---- optional per-kernel timing (measurement only; no reference counterpart) -------------
 * A caller-owned pool of HIP event pairs recorded on the launch stream around every kernel of the
 * selected classes.  gnet_profiler_read synchronises the recorded events, adds the elapsed
 * milliseconds and launch counts per class into ms_sum[GNET_KCLASS_COUNT] / count[...], and resets. */
enum {
  GNET_K_GRAPH = 0, GNET_K_PACK, GNET_K_PW_FWD, GNET_K_NODE_FWD, GNET_K_EDGE_FWD, GNET_K_LOSS,
  GNET_K_HEAD_BWD, GNET_K_WINNERS, GNET_K_EDGE_BWD, GNET_K_GATHER, GNET_K_NODE_BWD, GNET_K_PW_BWD,
  GNET_K_W1_SUMS, GNET_K_W1_CLASS, GNET_K_REDUCE, GNET_K_GEOMETRY, GNET_KCLASS_COUNT
};
int gnet_profiler_create(int32_t capacity, uint32_t class_mask, void** out);
int gnet_profiler_read(void* profiler, double* ms_sum, int32_t* count);
/* Bracket only every stride-th launch of each selected class (default 1): an event pair costs a few microseconds of the
 * stream's time, so a timed region that wants a kernel's average duration samples its launches instead of bracketing all. */
int gnet_profiler_set_stride(void* profiler, int32_t stride);
int gnet_profiler_destroy(void* profiler);
/* Scopes for the entry points that take no gnet_buffers (gnet_graph_count / _fill / _transpose): gnet_profiler_begin records the
 * opening event of a launch of class `cls` on `stream` and returns a scope index (-1: class not selected / not sampled / pool
 * full), gnet_profiler_end(index) records the closing event on the same stream (index -1: no-op). */
int gnet_profiler_begin(void* profiler, int32_t cls, gnet_stream_t stream);
int gnet_profiler_end(void* profiler, int32_t idx, gnet_stream_t stream);

/* ---- numerics probe of the FC kernels' product arithmetic (tests only; csrc/debug.hip) -------------------
 * c[M,N] = a[M,K] . b[K,N], row-major fp32, M and N multiples of 32, K a multiple of 16, a 16-byte aligned.
 * mode 0: every fp32 product as the six bf16 products of exact three-term splits, fp32 accumulation -- the library's own
 * split3 / mma6 (csrc/common.hpp), i.e. the arithmetic of edge_fwd_w, pw_fwd2, pw_bwd_main; mode 1: v_mfma_f32_32x32x2_f32.
 * a_terms (optional, mode 0): [3][M,K] fp32 = the hi / mid / lo bf16 terms of a, widened (hi + mid + lo == a exactly).
 * Operands must be finite (+/-inf splits into inf - inf = NaN). */
int gnet_debug_gemm(const float* a, const float* b, int64_t M, int64_t K, int64_t N, int mode, float* c,
                    float* a_terms, gnet_stream_t stream);

/* Version / build info string (static storage). */
const char* gnet_version(void);

/* ---- ABI guard ---------------------------------------------------------------------------------
 * The structs above cross the boundary by value layout (a foreign-function binding mirrors them field by field:
 * gossipnet_amd/_lib.py).  GNET_ABI_VERSION changes whenever a struct field, an enum value or an
 * entry point's signature changes; gnet_abi_version() returns the value the LIBRARY was compiled with.
 * gnet_abi_sizes fills out[0..7] = sizeof(gnet_config), sizeof(gnet_shape), sizeof(gnet_inputs),
 * sizeof(gnet_buffers), offsetof(gnet_buffers, head1), offsetof(gnet_buffers, d_g1),
 * offsetof(gnet_buffers, match_ws_bytes), offsetof(gnet_buffers, start_feat) and returns GNET_KCLASS_COUNT.
 * A binding compares both with its own mirror before the first call and refuses to go on when they differ
 * (a shifted gnet_buffers would hand the kernels wrong device pointers without any error). */
#define GNET_ABI_VERSION 8
int gnet_abi_version(void);
int gnet_abi_sizes(size_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* GOSSIPNET_HIP_H */
