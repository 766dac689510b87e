/* CPU ORACLE (plain C) for the two custom ops -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * PARITY UNPINNED: the reference has no golden vectors for these ops and its
 * sources need TensorFlow/Eigen headers that are absent here (unbuildable);
 * this restatement is pinned by the hand-derived KATs of SURVEY.md 8c only.
 *
 * Follows (paths under /root/reference/nms_net):
 *   matching_module/det_matching.cc:95-159        -> oracle_det_matching
 *   roi_pooling_layer/roi_pooling_op.cc:128-187   -> oracle_roi_pool_fwd
 *   roi_pooling_layer/roi_pooling_op.cc:374-449   -> oracle_roi_pool_bwd
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* det_matching.cc:45-51,95-98 sort with std::sort (unstable).  The tie order is
 * libstdc++-defined; we DEFINE it as a stable ascending sort (then reversed for
 * the detections, det_matching.cc:96), i.e. equal scores -> higher index first,
 * equal ignore flags -> lower index first. */
static void stable_argsort_f(const float *v, int n, int *idx) {
  int *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) idx[i] = i;
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int a = lo, b = mid, o = lo;
      while (a < mid && b < hi) tmp[o++] = (v[idx[b]] < v[idx[a]]) ? idx[b++] : idx[a++];
      while (a < mid) tmp[o++] = idx[a++];
      while (b < hi) tmp[o++] = idx[b++];
    }
    memcpy(idx, tmp, sizeof(int) * (size_t)n);
  }
  free(tmp);
}

int oracle_det_matching(const float *ious, const float *score, const uint8_t *ignore,
                        int n_dets, int n_gt, float *labels, float *weights, int32_t *assignment) {
  const float iou_thresh = 0.5f;                                   /* :73 */
  int *det_order = (int *)malloc(sizeof(int) * (size_t)(n_dets + 1));
  int *gt_order = (int *)malloc(sizeof(int) * (size_t)(n_gt + 1));
  uint8_t *is_matched = (uint8_t *)calloc((size_t)(n_gt + 1), 1);
  float *ign = (float *)malloc(sizeof(float) * (size_t)(n_gt + 1));
  stable_argsort_f(score, n_dets, det_order);                      /* :95 */
  for (int i = 0; i < n_dets / 2; ++i) {                           /* :96 reverse */
    int t = det_order[i]; det_order[i] = det_order[n_dets - 1 - i]; det_order[n_dets - 1 - i] = t;
  }
  for (int g = 0; g < n_gt; ++g) ign[g] = ignore[g] ? 1.f : 0.f;
  stable_argsort_f(ign, n_gt, gt_order);                           /* :98 */
  for (int i = 0; i < n_dets; ++i) { labels[i] = 0.f; weights[i] = 1.f; assignment[i] = -1; } /* :105-117 */
  for (int di = 0; di < n_dets; ++di) {                            /* :125 */
    const int det = det_order[di];
    float iou = iou_thresh;
    int match = -1;
    for (int gi = 0; gi < n_gt; ++gi) {
      const int gt = gt_order[gi];
      if (is_matched[gt] && !ignore[gt]) continue;                 /* :134 */
      if (match > -1 && ignore[gt]) break;                         /* :138 */
      if (ious[(size_t)det * n_gt + gt] < iou) continue;           /* :142 */
      iou = ious[(size_t)det * n_gt + gt];                         /* :147 */
      match = gt;
    }
    if (match > -1) {                                              /* :151-158 */
      is_matched[match] = 1;
      labels[det] = 1.f;
      assignment[det] = match;
      if (ignore[match]) weights[det] = 0.f;
    }
  }
  free(det_order); free(gt_order); free(is_matched); free(ign);
  return 0;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

int oracle_roi_pool_fwd(const float *data, int B, int H, int W, int C, const float *rois, int R,
                        int PH, int PW, float spatial_scale, float *top, int32_t *argmax) {
  (void)B;
  const int64_t total = (int64_t)R * PH * PW * C;
  for (int64_t b = 0; b < total; ++b) {                            /* :130-186 */
    int64_t n = b;
    int c = (int)(n % C); n /= C;
    int pw = (int)(n % PW); n /= PW;
    int ph = (int)(n % PH); n /= PH;
    const float *roi = rois + n * 5;
    int roi_batch_ind = (int)roi[0];
    int roi_start_w = (int)round(roi[1] * spatial_scale);           /* :145-148: C round() on double */
    int roi_start_h = (int)round(roi[2] * spatial_scale);
    int roi_end_w = (int)round(roi[3] * spatial_scale);
    int roi_end_h = (int)round(roi[4] * spatial_scale);
    int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)PH;
    const float bin_size_w = (float)roi_width / (float)PW;
    int hstart = (int)floor(ph * bin_size_h);
    int wstart = (int)floor(pw * bin_size_w);
    int hend = (int)ceil((ph + 1) * bin_size_h);
    int wend = (int)ceil((pw + 1) * bin_size_w);
    hstart = imin(imax(hstart + roi_start_h, 0), H);
    hend = imin(imax(hend + roi_start_h, 0), H);
    wstart = imin(imax(wstart + roi_start_w, 0), W);
    wend = imin(imax(wend + roi_start_w, 0), W);
    int is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0.f : -FLT_MAX;
    int maxidx = -1;
    const float *bottom = data + (size_t)roi_batch_ind * C * H * W;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        int bi = (h * W + w) * C + c;
        if (bottom[bi] > maxval) { maxval = bottom[bi]; maxidx = bi; }
      }
    top[b] = maxval;
    argmax[b] = maxidx;
  }
  return 0;
}

int oracle_roi_pool_bwd(const float *top_diff, const int32_t *argmax, const float *rois, int B, int H,
                        int W, int C, int R, int PH, int PW, float spatial_scale, float *bottom_diff) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t b = 0; b < total; ++b) {                            /* :376-448 */
    int64_t n = b;
    int c = (int)(n % C); n /= C;
    int w = (int)(n % W); n /= W;
    int h = (int)(n % H); n /= H;
    float gradient = 0.f;
    for (int r = 0; r < R; ++r) {
      const float *roi = rois + (size_t)r * 5;
      int roi_batch_ind = (int)roi[0];
      if (n != roi_batch_ind) continue;
      int roi_start_w = (int)round(roi[1] * spatial_scale);
      int roi_start_h = (int)round(roi[2] * spatial_scale);
      int roi_end_w = (int)round(roi[3] * spatial_scale);
      int roi_end_h = (int)round(roi[4] * spatial_scale);
      if (!(w >= roi_start_w && w <= roi_end_w && h >= roi_start_h && h <= roi_end_h)) continue;
      size_t offset = (size_t)r * PH * PW * C;
      const float *otd = top_diff + offset;
      const int32_t *oam = argmax + offset;
      int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
      int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
      const float bin_size_h = (float)roi_height / (float)PH;
      const float bin_size_w = (float)roi_width / (float)PW;
      int phstart = (int)floor((int)(h - roi_start_h) / bin_size_h);      /* :428-431 */
      int phend = (int)ceil((int)(h - roi_start_h + 1) / bin_size_h);
      int pwstart = (int)floor((int)(w - roi_start_w) / bin_size_w);
      int pwend = (int)ceil((int)(w - roi_start_w + 1) / bin_size_w);
      phstart = imin(imax(phstart, 0), PH);
      phend = imin(imax(phend, 0), PH);
      pwstart = imin(imax(pwstart, 0), PW);
      pwend = imin(imax(pwend, 0), PW);
      for (int ph = phstart; ph < phend; ++ph)
        for (int pw = pwstart; pw < pwend; ++pw)
          if (oam[(ph * PW + pw) * C + c] == (h * W + w) * C + c)
            gradient += otd[(ph * PW + pw) * C + c];
    }
    bottom_diff[b] = gradient;
  }
  return 0;
}
