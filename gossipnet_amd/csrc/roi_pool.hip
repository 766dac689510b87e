// RoiPool / RoiPoolGrad (Fast R-CNN max RoI pooling, NHWC) -- replaces
// nms_net/roi_pooling_layer/roi_pooling_op.cc:128-187 (forward) and :374-449 (backward);
// the CPU kernels define the semantics (the reference CUDA forward reads the wrong image for
// batch index > 0, roi_pooling_op_gpu.cu:75-76 -- not reproduced).
//
// HBM-bound op: at the contract shape (R = 2000, 7 x 7, C = 1024) it writes 2 x 401 MB (top, argmax) and reads
// ~roi_area x C x 4 bytes per ROI from a feature map that stays in L2.  Layout decisions for gfx950:
//   forward    one WAVE per (roi, bin): the ROI is decoded once per wave in scalar registers, the lanes run over the
//              channels with 16-byte loads / stores (a wave-instruction moves 1 KB of a pixel's channel vector), the
//              bin's pixels are visited in the reference's order (h, then w; strict >: the first maximum wins), so
//              top and argmax are bit-exact.  (The reference CUDA kernel -- one thread per output element, per-thread
//              ROI decode, 4-byte accesses -- is not the model.)
//   backward   deterministic (default): one wave per input PIXEL, lanes over the channels; the wave walks the ROIs in
//              index order (scalar box test), and for the ROIs that contain the pixel the feasible bins in (ph, pw)
//              order -- the CPU kernel's summation order, so bottom_diff is bit-exact and reproducible.
//              atomic (roi_pool_bwd_atomic_f32): every pooled element adds its gradient to its arg-max with one float
//              atomic: O(R PH PW C) instead of re-reading each pooled element once per pixel of its bin; the order of
//              the additions is not fixed.  NB this is the plain arg-max scatter, which the reference's RoiPoolGrad is
//              not always: its in-ROI / feasible-bin tests (:405-431) drop a pooled element whose arg-max pixel lies one
//              past the rounded ROI end (ceil((pw + 1) * bin) can exceed the ROI width in float) -- the default kernel
//              reproduces the reference bit for bit, the atomic one does not in those cases.
#include "common.hpp"

namespace {

struct RoiBox { int start_w, start_h, end_w, end_h, batch; float bin_h, bin_w; };

// roi_pooling_op.cc:143-156: round() = half away from zero, evaluated on the float product
__device__ __forceinline__ RoiBox roi_decode(const float* __restrict__ roi, float scale, int PH, int PW) {
  RoiBox b;
  b.batch = (int)roi[0];
  b.start_w = (int)round((double)(roi[1] * scale));
  b.start_h = (int)round((double)(roi[2] * scale));
  b.end_w = (int)round((double)(roi[3] * scale));
  b.end_h = (int)round((double)(roi[4] * scale));
  const int roi_width = max(b.end_w - b.start_w + 1, 1);
  const int roi_height = max(b.end_h - b.start_h + 1, 1);
  b.bin_h = (float)roi_height / (float)PH;
  b.bin_w = (float)roi_width / (float)PW;
  return b;
}

template <bool VEC4>
__global__ void __launch_bounds__(256) roi_pool_fwd(const float* __restrict__ data, int H, int W, int C,
                                                    const float* __restrict__ rois, long long nbins, int PH, int PW,
                                                    float scale, float* __restrict__ top, int* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const long long bin = __builtin_amdgcn_readfirstlane((int)(((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) & 0x7fffffff));
  if (bin >= nbins) return;
  const int pw = (int)(bin % PW), ph = (int)((bin / PW) % PH);
  const long long r = bin / ((long long)PW * PH);
  const RoiBox b = roi_decode(rois + r * 5, scale, PH, PW);
  int hstart = (int)floorf(ph * b.bin_h), wstart = (int)floorf(pw * b.bin_w);
  int hend = (int)ceilf((ph + 1) * b.bin_h), wend = (int)ceilf((pw + 1) * b.bin_w);
  hstart = min(max(hstart + b.start_h, 0), H); hend = min(max(hend + b.start_h, 0), H);
  wstart = min(max(wstart + b.start_w, 0), W); wend = min(max(wend + b.start_w, 0), W);
  const bool is_empty = (hend <= hstart) || (wend <= wstart);
  const float init = is_empty ? 0.f : -3.402823466e+38f;
  const float* bottom = data + (size_t)b.batch * C * H * W;
  float* tp = top + (size_t)bin * C; int* ap = argmax + (size_t)bin * C;
  if (VEC4) {
    // 1024 channels per pass = four 16-byte groups per lane, all four requested per pixel before the compares (the op
    // is latency-bound: one load in flight per lane kept it at 1.5 TB/s)
    for (int c0 = 0; c0 < C; c0 += 1024) {
      float4 mv[4]; int4 mi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { mv[i] = make_float4(init, init, init, init); mi[i] = make_int4(-1, -1, -1, -1); }
      for (int h = hstart; h < hend; ++h)
        for (int w = wstart; w < wend; ++w) {
          const int pb = (h * W + w) * C + c0 + 4 * lane;
          float4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            v[i] = (c0 + 256 * i + 4 * lane < C) ? *reinterpret_cast<const float4*>(bottom + pb + 256 * i) : make_float4(init, init, init, init);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int bi = pb + 256 * i;
            if (v[i].x > mv[i].x) { mv[i].x = v[i].x; mi[i].x = bi; }
            if (v[i].y > mv[i].y) { mv[i].y = v[i].y; mi[i].y = bi + 1; }
            if (v[i].z > mv[i].z) { mv[i].z = v[i].z; mi[i].z = bi + 2; }
            if (v[i].w > mv[i].w) { mv[i].w = v[i].w; mi[i].w = bi + 3; }
          }
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + 256 * i + 4 * lane;
        if (c < C) { *reinterpret_cast<float4*>(tp + c) = mv[i]; *reinterpret_cast<int4*>(ap + c) = mi[i]; }
      }
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      float mv = init; int mi = -1;
      for (int h = hstart; h < hend; ++h)
        for (int w = wstart; w < wend; ++w) {
          const int bi = (h * W + w) * C + c;
          const float v = bottom[bi];
          if (v > mv) { mv = v; mi = bi; }
        }
      tp[c] = mv; ap[c] = mi;
    }
  }
}

// deterministic backward: one wave per input pixel (n, h, w)
template <bool VEC4>
__global__ void __launch_bounds__(256) roi_pool_bwd_pixel(const float* __restrict__ top_diff, const int* __restrict__ argmax,
                                                          const float* __restrict__ rois, int B, int H, int W, int C, int R,
                                                          int PH, int PW, float scale, float* __restrict__ bottom_diff) {
  const int lane = threadIdx.x & 63;
  const long long pix = __builtin_amdgcn_readfirstlane((int)((long long)blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (pix >= (long long)B * H * W) return;
  const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  const int here = (h * W + w) * C;
  constexpr int NV = VEC4 ? 4 : 16;         // channel groups a lane keeps: 4 x float4 (C <= 1024) / 16 floats
  float4 acc4[4]; float acc1[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
  (void)NV;
  for (int c0 = 0; c0 < C; c0 += 1024) {     // 1024 channels per outer pass
    for (int r = 0; r < R; ++r) {
      const RoiBox b = roi_decode(rois + (size_t)r * 5, scale, PH, PW);
      if (n != b.batch) continue;
      if (!(w >= b.start_w && w <= b.end_w && h >= b.start_h && h <= b.end_h)) continue;
      int phstart = (int)floorf((float)(h - b.start_h) / b.bin_h), phend = (int)ceilf((float)(h - b.start_h + 1) / b.bin_h);   // :428-431
      int pwstart = (int)floorf((float)(w - b.start_w) / b.bin_w), pwend = (int)ceilf((float)(w - b.start_w + 1) / b.bin_w);
      phstart = min(max(phstart, 0), PH); phend = min(max(phend, 0), PH);
      pwstart = min(max(pwstart, 0), PW); pwend = min(max(pwend, 0), PW);
      for (int ph = phstart; ph < phend; ++ph)
        for (int pw = pwstart; pw < pwend; ++pw) {
          const size_t o = ((size_t)r * PH * PW + (size_t)ph * PW + pw) * C + c0;
          if (VEC4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = 256 * i + 4 * lane;
              if (c0 + c < C) {
                const int4 am = *reinterpret_cast<const int4*>(argmax + o + c);
                const float4 g = *reinterpret_cast<const float4*>(top_diff + o + c);
                const int want = here + c0 + c;
                if (am.x == want) acc4[i].x += g.x;
                if (am.y == want + 1) acc4[i].y += g.y;
                if (am.z == want + 2) acc4[i].z += g.z;
                if (am.w == want + 3) acc4[i].w += g.w;
              }
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int c = 64 * i + lane;
              if (c0 + c < C && argmax[o + c] == here + c0 + c) acc1[i] += top_diff[o + c];
            }
          }
        }
    }
    float* out = bottom_diff + (size_t)pix * C + c0;
    if (VEC4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 256 * i + 4 * lane;
        if (c0 + c < C) *reinterpret_cast<float4*>(out + c) = acc4[i];
        acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 64 * i + lane;
        if (c0 + c < C) out[c] = acc1[i];
        acc1[i] = 0.f;
      }
    }
  }
}

__global__ void __launch_bounds__(256) roi_pool_bwd_atomic(const float* __restrict__ top_diff, const int* __restrict__ argmax,
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
  const long long nbins = (long long)R * pooled_h * pooled_w;
  if (nbins * C == 0) return GNET_OK;
  if (!bottom_data || !bottom_rois || !top_data || !argmax) return GNET_ERR_INVALID;
  if ((long long)H * W * C > 0x7fffffffLL || nbins > 0x7fffffffLL) return GNET_ERR_UNSUPPORTED;   // argmax is an int32 index within the image
  const unsigned grid = (unsigned)((nbins + 3) / 4);
  if ((C & 3) == 0)
    roi_pool_fwd<true><<<grid, 256, 0, (hipStream_t)stream>>>(bottom_data, H, W, C, bottom_rois, nbins, pooled_h, pooled_w, spatial_scale, top_data, argmax);
  else
    roi_pool_fwd<false><<<grid, 256, 0, (hipStream_t)stream>>>(bottom_data, H, W, C, bottom_rois, nbins, pooled_h, pooled_w, spatial_scale, top_data, argmax);
  return launch_status();
}

extern "C" int roi_pool_bwd_f32(const float* top_diff, const int32_t* argmax, const float* bottom_rois, int32_t B,
                                int32_t H, int32_t W, int32_t C, int32_t R, int32_t pooled_h, int32_t pooled_w,
                                float spatial_scale, float* bottom_diff, gnet_stream_t stream) {
  clear_hip_error();
  if (pooled_h < 0 || pooled_w < 0 || B < 0 || H < 0 || W < 0 || C < 0 || R < 0) return GNET_ERR_INVALID;
  const long long image_elems = (long long)H * W * C;
  if (B * image_elems == 0) return GNET_OK;
  if (!bottom_diff) return GNET_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)R * pooled_h * pooled_w * C;
  if (total == 0) { HIP_CHECK_RET(hipMemsetAsync(bottom_diff, 0, (size_t)B * image_elems * sizeof(float), s)); return GNET_OK; }
  if (!top_diff || !argmax || !bottom_rois) return GNET_ERR_INVALID;
  const long long pixels = (long long)B * H * W;
  if (pixels > 0x7fffffffLL) return GNET_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)((pixels + 3) / 4);
  if ((C & 3) == 0)
    roi_pool_bwd_pixel<true><<<grid, 256, 0, s>>>(top_diff, argmax, bottom_rois, B, H, W, C, R, pooled_h, pooled_w, spatial_scale, bottom_diff);
  else
    roi_pool_bwd_pixel<false><<<grid, 256, 0, s>>>(top_diff, argmax, bottom_rois, B, H, W, C, R, pooled_h, pooled_w, spatial_scale, bottom_diff);
  return launch_status();
}

extern "C" int roi_pool_bwd_atomic_f32(const float* top_diff, const int32_t* argmax, const float* bottom_rois, int32_t B,
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
  roi_pool_bwd_atomic<<<grid, 256, 0, s>>>(top_diff, argmax, bottom_rois, total, pooled_h * pooled_w * C, image_elems, B, bottom_diff);
  return launch_status();
}
