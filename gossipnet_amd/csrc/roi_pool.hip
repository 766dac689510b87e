// RoiPool / RoiPoolGrad (Fast R-CNN max RoI pooling, NHWC) -- replaces
// nms_net/roi_pooling_layer/roi_pooling_op.cc:128-187 (forward) and :374-449 (backward);
// the CPU kernels define the semantics (the reference CUDA forward reads the wrong image for
// batch index > 0, roi_pooling_op_gpu.cu:75-76 -- not reproduced).
//
// Forward: one thread per output element (r, ph, pw, c), channel fastest -> coalesced reads of
// the NHWC feature map and coalesced writes of top/argmax.  HBM-bound: per ROI the kernel writes
// PH*PW*C*(4+4) bytes and reads ~roi_area*C*4 bytes (mostly from L2: the map is small).
// Backward: the reference scans all R ROIs for every input element (O(H*W*C*R)); here every
// pooled element scatters its gradient to argmax with one float atomic (O(R*PH*PW*C)); the
// summation order is not fixed, so bottom_diff equals the CPU kernel to rounding (<= 1e-5 relative),
// exactly when no input element is the argmax of more than one bin.
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256) roi_pool_fwd(const float* __restrict__ data, int H, int W, int C,
                                                    const float* __restrict__ rois, long long total, int PH, int PW,
                                                    float scale, float* __restrict__ top, int* __restrict__ argmax) {
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < total; b += (long long)gridDim.x * 256) {
    long long n = b;
    const int c = (int)(n % C); n /= C;
    const int pw = (int)(n % PW); n /= PW;
    const int ph = (int)(n % PH); n /= PH;
    const float* roi = rois + n * 5;
    const int roi_batch_ind = (int)roi[0];
    // roi_pooling_op.cc:145-148: round() = half away from zero, evaluated on the float product
    const int roi_start_w = (int)round((double)(roi[1] * scale));
    const int roi_start_h = (int)round((double)(roi[2] * scale));
    const int roi_end_w = (int)round((double)(roi[3] * scale));
    const int roi_end_h = (int)round((double)(roi[4] * scale));
    const int roi_width = max(roi_end_w - roi_start_w + 1, 1);
    const int roi_height = max(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)PH;
    const float bin_size_w = (float)roi_width / (float)PW;
    int hstart = (int)floorf(ph * bin_size_h);
    int wstart = (int)floorf(pw * bin_size_w);
    int hend = (int)ceilf((ph + 1) * bin_size_h);
    int wend = (int)ceilf((pw + 1) * bin_size_w);
    hstart = min(max(hstart + roi_start_h, 0), H);
    hend = min(max(hend + roi_start_h, 0), H);
    wstart = min(max(wstart + roi_start_w, 0), W);
    wend = min(max(wend + roi_start_w, 0), W);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0.f : -3.402823466e+38f;
    int maxidx = -1;
    const float* bottom = data + (size_t)roi_batch_ind * C * H * W;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        const int bi = (h * W + w) * C + c;
        const float v = bottom[bi];
        if (v > maxval) { maxval = v; maxidx = bi; }
      }
    top[b] = maxval;
    argmax[b] = maxidx;
  }
}

__global__ void __launch_bounds__(256) roi_pool_bwd(const float* __restrict__ top_diff, const int* __restrict__ argmax,
                                                    const float* __restrict__ rois, long long total, int per_roi,
                                                    long long image_elems, int B, float* __restrict__ bottom_diff) {
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < total; b += (long long)gridDim.x * 256) {
    const int idx = argmax[b];
    if (idx < 0) continue;
    const long long r = b / per_roi;
    const int bi = (int)rois[r * 5];
    if (bi < 0 || bi >= B) continue;
    atomic_add_f32(bottom_diff + (size_t)bi * image_elems + idx, top_diff[b]);
  }
}

}  // namespace

extern "C" int roi_pool_fwd_f32(const float* bottom_data, int32_t B, int32_t H, int32_t W, int32_t C,
                                const float* bottom_rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                                float spatial_scale, float* top_data, int32_t* argmax, gnet_stream_t stream) {
  clear_hip_error();
  if (pooled_h < 0 || pooled_w < 0 || B < 0 || H < 0 || W < 0 || C < 0 || R < 0) return GNET_ERR_INVALID;  // :59-77
  const long long total = (long long)R * pooled_h * pooled_w * C;
  if (total == 0) return GNET_OK;
  if (!bottom_data || !bottom_rois || !top_data || !argmax) return GNET_ERR_INVALID;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  roi_pool_fwd<<<grid, 256, 0, (hipStream_t)stream>>>(bottom_data, H, W, C, bottom_rois, total, pooled_h, pooled_w,
                                                      spatial_scale, top_data, argmax);
  return launch_status();
}

extern "C" int roi_pool_bwd_f32(const float* top_diff, const int32_t* argmax, const float* bottom_rois, int32_t B,
                                int32_t H, int32_t W, int32_t C, int32_t R, int32_t pooled_h, int32_t pooled_w,
                                float spatial_scale, float* bottom_diff, gnet_stream_t stream) {
  clear_hip_error();
  (void)spatial_scale;
  if (pooled_h < 0 || pooled_w < 0 || B < 0 || H < 0 || W < 0 || C < 0 || R < 0) return GNET_ERR_INVALID;
  const long long image_elems = (long long)H * W * C;
  if (B * image_elems == 0) return GNET_OK;
  if (!bottom_diff) return GNET_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  HIP_CHECK_RET(hipMemsetAsync(bottom_diff, 0, (size_t)B * image_elems * sizeof(float), s));
  const long long total = (long long)R * pooled_h * pooled_w * C;
  if (total == 0) return GNET_OK;
  if (!top_diff || !argmax || !bottom_rois) return GNET_ERR_INVALID;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  roi_pool_bwd<<<grid, 256, 0, s>>>(top_diff, argmax, bottom_rois, total, pooled_h * pooled_w * C, image_elems, B,
                                    bottom_diff);
  return launch_status();
}
