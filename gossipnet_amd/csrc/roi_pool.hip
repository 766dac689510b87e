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
//   backward   deterministic (default), C a multiple of 256: one wave per (2 x 2 block of input pixels, 256-channel slice); the
//              entry bookkeeping of 64 ROIs at a time runs across the lanes, a pooled vector is loaded once per block and
//              compared against its four pixels, each (pixel, channel) sum is formed by one lane in the CPU kernel's
//              (roi, ph, pw) order -- bit-exact and reproducible (roi_pool_bwd_block below).  Other C: one wave per input
//              pixel walking the ROIs with a scalar iterator (roi_pool_bwd_pixel).
//              atomic (roi_pool_bwd_atomic_f32): every pooled element adds its gradient to its arg-max with one float
//              atomic: O(R PH PW C) instead of re-reading each pooled element once per pixel of its bin; the order of
//              the additions is not fixed.  NB this is the plain arg-max scatter, which the reference's RoiPoolGrad is
//              not always: its in-ROI / feasible-bin tests (:405-431) drop a pooled element whose arg-max pixel lies one
//              past the rounded ROI end (ceil((pw + 1) * bin) can exceed the ROI width in float) -- the default kernel
//              reproduces the reference bit for bit, the atomic one does not in those cases.
#include <stdlib.h>
#include "common.hpp"

namespace {

struct RoiBox { int start_w, start_h, end_w, end_h, batch; float bin_h, bin_w; };

// roi_pooling_op.cc:143-156: round() = half away from zero, evaluated on the float product
__device__ __forceinline__ RoiBox roi_decode(const float* __restrict__ roi, float scale, int PH, int PW) {
  RoiBox b;
  b.batch = roi[0] == roi[0] ? (int)roi[0] : -1;        // (a NaN index: no image, like any index outside the batch)
  // (roundf on the float product: the product is exactly representable as a double and half-away-from-zero rounding to
  // an integer is the same function in either precision, so this IS round((double)x) -- at a fraction of the instructions)
  b.start_w = (int)roundf(roi[1] * scale);
  b.start_h = (int)roundf(roi[2] * scale);
  b.end_w = (int)roundf(roi[3] * scale);
  b.end_h = (int)roundf(roi[4] * scale);
  const int roi_width = max(b.end_w - b.start_w + 1, 1);
  const int roi_height = max(b.end_h - b.start_h + 1, 1);
  b.bin_h = (float)roi_height / (float)PH;
  b.bin_w = (float)roi_width / (float)PW;
  return b;
}

template <bool VEC4>
__global__ void __launch_bounds__(256) roi_pool_fwd(const float* __restrict__ data, int B, int H, int W, int C,
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
  // (a ROI whose image index is not one of the batch's -- undefined behaviour in the reference, which reads that address --
  //  pools nothing: zeros and arg-max -1)
  const bool no_image = b.batch < 0 || b.batch >= B;
  const bool is_empty = (hend <= hstart) || (wend <= wstart) || no_image;
  const float init = is_empty ? 0.f : -3.402823466e+38f;
  const float* bottom = data + (size_t)(no_image ? 0 : b.batch) * C * H * W;
  float* tp = top + (size_t)bin * C; int* ap = argmax + (size_t)bin * C;
  if (VEC4) {
    // 1024 channels per pass = four 16-byte groups per lane, all four requested per pixel before the compares (the op
    // is latency-bound: one load in flight per lane kept it at 1.5 TB/s)
    for (int c0 = 0; c0 < C; c0 += 1024) {
      float4 mv[4]; int4 mi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { mv[i] = make_float4(init, init, init, init); mi[i] = make_int4(-1, -1, -1, -1); }
      // the bin's pixels in the reference's order (h, then w), TWO per step: eight 16-byte loads per lane in flight
      // (a bin has 4-9 pixels; one pixel per step was a chain of as many L2 round trips)
      const int bw = wend - wstart, npix = is_empty ? 0 : (hend - hstart) * bw;
      for (int p = 0; p < npix; p += 2) {
        const int p1 = min(p + 1, npix - 1);                  // (odd count: the last pixel twice -- a repeated value never wins a strict >)
        const int pb0 = ((hstart + p / bw) * W + wstart + p % bw) * C + c0 + 4 * lane;
        const int pb1 = ((hstart + p1 / bw) * W + wstart + p1 % bw) * C + c0 + 4 * lane;
        float4 v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = c0 + 256 * i + 4 * lane < C;
          v0[i] = ok ? *reinterpret_cast<const float4*>(bottom + pb0 + 256 * i) : make_float4(init, init, init, init);
          v1[i] = ok ? *reinterpret_cast<const float4*>(bottom + pb1 + 256 * i) : make_float4(init, init, init, init);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b0 = pb0 + 256 * i, b1 = pb1 + 256 * i;
          if (v0[i].x > mv[i].x) { mv[i].x = v0[i].x; mi[i].x = b0; }
          if (v0[i].y > mv[i].y) { mv[i].y = v0[i].y; mi[i].y = b0 + 1; }
          if (v0[i].z > mv[i].z) { mv[i].z = v0[i].z; mi[i].z = b0 + 2; }
          if (v0[i].w > mv[i].w) { mv[i].w = v0[i].w; mi[i].w = b0 + 3; }
          if (v1[i].x > mv[i].x) { mv[i].x = v1[i].x; mi[i].x = b1; }
          if (v1[i].y > mv[i].y) { mv[i].y = v1[i].y; mi[i].y = b1 + 1; }
          if (v1[i].z > mv[i].z) { mv[i].z = v1[i].z; mi[i].z = b1 + 2; }
          if (v1[i].w > mv[i].w) { mv[i].w = v1[i].w; mi[i].w = b1 + 3; }
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

// Forward for C = 256 / 512 / 1024 / 2048 (measured at 1024).  Two measured facts shape it (rocprofv3 --pmc on the one-wave-per-bin kernel: 63 % of the wave
// time waiting for an issue slot, ~900 instructions per wave of which 485 scalar -- ROI decode, window arithmetic and integer
// division by the window width -- against 15 loads; the 803 MB of stores alone take 119 us):
//   * the kernel is INSTRUCTION-ISSUE bound, not memory bound: one wave now handles a whole ROW of bins (roi, ph, all pw)
//     of one 256-channel slice, so the ROI is decoded once per seven bins, and the window is walked by nested h / w loops
//     (four pixels of a row in flight; a clamped repeat of the row's last pixel never wins a strict >) -- no divisions;
//   * the channel dimension is dealt to the XCDs (workgroups go round-robin to the 8 XCDs, blockIdx % 8): the feature map of
//     an image (9.8 MB) does not fit an XCD's 4 MB L2, a 256-channel slice (2.4 MB) does; at C = 1024 XCDs 2 s and 2 s + 1 take slice s.
// Pixels of a bin are visited in the reference's order (h, then w; strict >: the first maximum wins): top / argmax bit-exact.
// (C = 256, 512, 1024, 2048 -- the usual trunk widths: 8 / C256 XCDs per 256-channel slice, C a compile-time constant of the
//  address arithmetic)
#ifndef ROI_X
#define ROI_X 0   /* ablation mask of measurement builds (WRONG results, durations only): 1 = no arg-max tracking (values only), 2 = no loads (the window walk and compares on a constant), 4 = no stores, 8 = no compares at all */
#endif
template <int C>
__global__ void __launch_bounds__(256) roi_pool_fwd_rows(const float* __restrict__ data, int B, int H, int W,
                                                         const float* __restrict__ rois, int nrows, int PH, int PW,
                                                         float scale, float* __restrict__ top, int* __restrict__ argmax) {
  constexpr int XPS = 8 / (C / 256);                          // XCDs per slice
  static_assert(C % 256 == 0 && XPS >= 1 && XPS * (C / 256) == 8, "1, 2, 4 or 8 slices");
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, slice = xcd / XPS;
  const int row = ((int)(blockIdx.x >> 3) * XPS + (xcd % XPS)) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (row >= nrows) return;
  const int ph = row % PH, r = row / PH;
  const RoiBox b = roi_decode(rois + (size_t)r * 5, scale, PH, PW);
  int hstart = (int)floorf(ph * b.bin_h), hend = (int)ceilf((ph + 1) * b.bin_h);
  hstart = min(max(hstart + b.start_h, 0), H); hend = min(max(hend + b.start_h, 0), H);
  const bool no_image = b.batch < 0 || b.batch >= B;     // (undefined in the reference; here such a ROI pools nothing)
  if (no_image) hend = hstart;
  // (uniform image base + a lane-constant byte offset: the loads take the scalar-base form, no per-lane address arithmetic;
  //  the arg-max candidates are the UNIFORM pixel offsets -- a select with a scalar operand -- and the lane's channel offset is
  //  added once per bin)
  const float* img = data + (size_t)(no_image ? 0 : b.batch) * C * H * W + 256 * slice;
  const unsigned lane_b = 16u * lane;
  const int cbase = 256 * slice + 4 * lane;
  size_t out = ((size_t)row * PW) * C + cbase;
  for (int pw = 0; pw < PW; ++pw, out += C) {
    int wstart = (int)floorf(pw * b.bin_w), wend = (int)ceilf((pw + 1) * b.bin_w);
    wstart = min(max(wstart + b.start_w, 0), W); wend = min(max(wend + b.start_w, 0), W);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    const float init = is_empty ? 0.f : -3.402823466e+38f;
    float4 mv = make_float4(init, init, init, init); int4 mi = make_int4(-1, -1, -1, -1);
    if (!is_empty) {
      // the window's pixels as one sequence (h, then w), four per step; slot q walks positions q, q + 4, ... with its own
      // (h, w) pair advanced by scalar adds -- no division, no padding at row ends; past the end a slot repeats the last
      // pixel, which never wins a strict >
      const int bw = wend - wstart, npix = (hend - hstart) * bw;
      int hq[4], wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { hq[q] = hstart; wq[q] = wstart + q; while (wq[q] >= wend) { wq[q] -= bw; ++hq[q]; } }
      const int last = ((hend - 1) * W + wend - 1) * C;
      for (int p = 0; p < npix; p += 4) {
        int pb[4]; float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          pb[q] = p + q < npix ? (hq[q] * W + wq[q]) * C : last;
          if (ROI_X & 2) { const float f_ = __int_as_float(pb[q] + lane); v[q] = make_float4(f_, f_, f_, f_); }
          else v[q] = ldg4_b(img + pb[q], lane_b);
          wq[q] += 4;
          while (wq[q] >= wend) { wq[q] -= bw; ++hq[q]; }     // (scalar; measured: a branch-free select chain here is slower -- the kernel is issue-bound)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ROI_X & 8) { mv.x += v[q].x; mv.y += v[q].y; mv.z += v[q].z; mv.w += v[q].w; }
          else if (ROI_X & 1) { mv.x = fmaxf(mv.x, v[q].x); mv.y = fmaxf(mv.y, v[q].y); mv.z = fmaxf(mv.z, v[q].z); mv.w = fmaxf(mv.w, v[q].w); }
          else {
          if (v[q].x > mv.x) { mv.x = v[q].x; mi.x = pb[q]; }
          if (v[q].y > mv.y) { mv.y = v[q].y; mi.y = pb[q]; }
          if (v[q].z > mv.z) { mv.z = v[q].z; mi.z = pb[q]; }
          if (v[q].w > mv.w) { mv.w = v[q].w; mi.w = pb[q]; }
          }
        }
      }
      mi.x = mi.x < 0 ? -1 : mi.x + cbase; mi.y = mi.y < 0 ? -1 : mi.y + cbase + 1;
      mi.z = mi.z < 0 ? -1 : mi.z + cbase + 2; mi.w = mi.w < 0 ? -1 : mi.w + cbase + 3;
    }
    if ((ROI_X & 4) && mv.x != 12345.f) continue;
    *reinterpret_cast<float4*>(top + out) = mv;
    *reinterpret_cast<int4*>(argmax + out) = mi;
  }
}

// roi_pool_fwd_rows2 (round 6): the same decomposition with the SCALAR unit taken out of the pixel loop.  A CU has ONE scalar ALU for
// its four SIMDs, and roi_pool_fwd_rows spent ~15 scalar instructions per window pixel on its four (h, w) walkers, tail tests and address
// arithmetic: measured (tools/roi_ablate.sh), the kernel without its loads AND without its stores still took 250 of 368 us -- 115 k scalar
// instructions per SIMD = 460 k per CU at one per cycle.  Here a bin's window is walked row by row with the row's width a COMPILE-TIME
// constant (a wave-uniform switch per bin, widths 1-8; wider windows in steps of eight with a clamped tail): the loads of a row are buffer
// loads whose per-pixel offsets are the constants k * 4 C in the scalar-offset operand, the row's position is ONE vector add into the lane
// offset and the arg-max candidates are one vector add each -- per row one scalar add and the loop test.  Pixels in the reference's order
// (h, then w; strict >): top / argmax bit-exact.
template <int C, int BW>
__device__ __forceinline__ void roi_row_steps(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int idx, int rows, unsigned row_bytes, int row_idx, float4& mv, int4& mi) {
  for (int h = 0; h < rows; ++h, voff += row_bytes, idx += row_idx) {
    float4 v[BW];
#pragma unroll
    for (int k = 0; k < BW; ++k) {
      if (ROI_X & 2) { const float f_ = __int_as_float((int)voff + k); v[k] = make_float4(f_, f_, f_, f_); continue; }
      const auto q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, k * C * 4, 0);
      v[k] = __builtin_bit_cast(float4, q);
    }
#pragma unroll
    for (int k = 0; k < BW; ++k) {
      const int cand = idx + k * C;
      if (ROI_X & 8) { mv.x += v[k].x; mv.y += v[k].y; mv.z += v[k].z; mv.w += v[k].w; continue; }
      if (ROI_X & 1) { mv.x = fmaxf(mv.x, v[k].x); mv.y = fmaxf(mv.y, v[k].y); mv.z = fmaxf(mv.z, v[k].z); mv.w = fmaxf(mv.w, v[k].w); continue; }
      if (v[k].x > mv.x) { mv.x = v[k].x; mi.x = cand; }
      if (v[k].y > mv.y) { mv.y = v[k].y; mi.y = cand; }
      if (v[k].z > mv.z) { mv.z = v[k].z; mi.z = cand; }
      if (v[k].w > mv.w) { mv.w = v[k].w; mi.w = cand; }
    }
  }
  // (the 16-byte stores behind the loop want four consecutive registers: without this the allocator keeps a SECOND, consecutive copy of
  //  the running maxima up to date inside the loop -- four more selects per pixel)
  asm volatile("" : "+v"(mv.x), "+v"(mv.y), "+v"(mv.z), "+v"(mv.w));
}

template <int C>
__global__ void __launch_bounds__(256) roi_pool_fwd_rows2(const float* __restrict__ data, int B, int H, int W,
                                                          const float* __restrict__ rois, int nrows, int PH, int PW,
                                                          float scale, float* __restrict__ top, int* __restrict__ argmax) {
  constexpr int XPS = 8 / (C / 256);                          // XCDs per slice
  static_assert(C % 256 == 0 && XPS >= 1 && XPS * (C / 256) == 8, "1, 2, 4 or 8 slices");
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, slice = xcd / XPS;
  const int row = ((int)(blockIdx.x >> 3) * XPS + (xcd % XPS)) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (row >= nrows) return;
  const int ph = row % PH, r = row / PH;
  const RoiBox b = roi_decode(rois + (size_t)r * 5, scale, PH, PW);
  int hstart = (int)floorf(ph * b.bin_h), hend = (int)ceilf((ph + 1) * b.bin_h);
  hstart = min(max(hstart + b.start_h, 0), H); hend = min(max(hend + b.start_h, 0), H);
  // (the window bounds come out of float instructions, i.e. vector registers: made provably uniform here, or the row loop's counter and
  //  exit test stay in vector registers and the compiler keeps a second copy of the running maxima for the "divergent" exit)
  hstart = __builtin_amdgcn_readfirstlane(hstart); hend = __builtin_amdgcn_readfirstlane(hend);
  const bool no_image = b.batch < 0 || b.batch >= B;     // (undefined in the reference; here such a ROI pools nothing)
  if (no_image) hend = hstart;
  // the image's channel slice as a buffer: wave-uniform base (two readfirstlanes make that provable: no waterfall loop around the loads)
  const float* img = data + (size_t)(no_image ? 0 : b.batch) * C * H * W + 256 * slice;
  const unsigned long long ib = reinterpret_cast<unsigned long long>(img);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ib), hi = __builtin_amdgcn_readfirstlane((unsigned)(ib >> 32));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                                        (int)(((unsigned)H * W * C - 256u * slice) * 4u), 0x00020000);
  const unsigned lane_b = 16u * lane;
  const int cbase = 256 * slice + 4 * lane;
  const unsigned row_bytes = (unsigned)W * C * 4u;
  const int row_idx = W * C;
  const int rows = hend - hstart;
  size_t out = ((size_t)row * PW) * C + cbase;
  for (int pw = 0; pw < PW; ++pw, out += C) {
    int wstart = (int)floorf(pw * b.bin_w), wend = (int)ceilf((pw + 1) * b.bin_w);
    wstart = min(max(wstart + b.start_w, 0), W); wend = min(max(wend + b.start_w, 0), W);
    wstart = __builtin_amdgcn_readfirstlane(wstart); wend = __builtin_amdgcn_readfirstlane(wend);
    const int bw = wend - wstart;
    const bool is_empty = rows <= 0 || bw <= 0;
    const float init = is_empty ? 0.f : -3.402823466e+38f;
    float4 mv = make_float4(init, init, init, init); int4 mi = make_int4(-1, -1, -1, -1);
    if (!is_empty) {
      const int idx0 = (hstart * W + wstart) * C;                // (float index of the window's first pixel inside the image)
      const unsigned voff = lane_b + (unsigned)idx0 * 4u;
      switch (bw) {                                              // wave-uniform
        case 1: roi_row_steps<C, 1>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 2: roi_row_steps<C, 2>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 3: roi_row_steps<C, 3>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 4: roi_row_steps<C, 4>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 5: roi_row_steps<C, 5>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 6: roi_row_steps<C, 6>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 7: roi_row_steps<C, 7>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        case 8: roi_row_steps<C, 8>(rsrc, voff, idx0, rows, row_bytes, row_idx, mv, mi); break;
        default: {
          // wide windows: the reference's order is h, then w -- so row by row, eight columns per step, the last step of a row
          // shifted left to end at the row's end (its leading pixels were seen already: a repeated value never wins a strict >)
          unsigned vo = voff; int ix = idx0;
          for (int h = 0; h < rows; ++h, vo += row_bytes, ix += row_idx) {
            for (int w0 = 0; w0 < bw; w0 += 8) {
              const int ws = min(w0, bw - 8);
              roi_row_steps<C, 8>(rsrc, vo + (unsigned)ws * (C * 4u), ix + ws * C, 1, 0u, 0, mv, mi);
            }
          }
        }
      }
      mi.x = mi.x < 0 ? -1 : mi.x + cbase; mi.y = mi.y < 0 ? -1 : mi.y + cbase + 1;
      mi.z = mi.z < 0 ? -1 : mi.z + cbase + 2; mi.w = mi.w < 0 ? -1 : mi.w + cbase + 3;
    }
    if ((ROI_X & 4) && mv.x != 12345.f) continue;
    *reinterpret_cast<float4*>(top + out) = mv;
    *reinterpret_cast<int4*>(argmax + out) = mi;
  }
}

// roi_pool_fwd_cols: the same walk TRANSPOSED -- one wave = a COLUMN of bins (roi, all ph, one pw) of one 256-channel slice.  What bounds
// roi_pool_fwd_rows2 is the 8 GB its windows pull through the L1s (64 bytes per clock and CU: ~230 us; tools/roi_ablate.sh), 1.6 times
// the ROIs' area because neighbouring bins overlap by a pixel row / column (floor / ceil of fractional bin edges).  A column of bins has
// ONE width (a single compile-time dispatch per wave instead of one per bin), and its bins follow each other down the image: the row a
// bin shares with the next one is loaded ONCE and compared into both (two running (max, arg-max) sets in registers, rotated at a bin's
// last row) -- a fifth fewer loads.  Needs bin_h >= 1 (then a row belongs to at most two consecutive bins, the earlier bin first: the
// reference's order h, then w inside every bin is kept; strict >) and a window of at most 8 columns; other ROIs take the plain per-bin
// walk below in the same wave.  top / argmax bit-exact.
#ifndef ROI_NT
#define ROI_NT 1      /* the pooled outputs (0.8 GB against a 2.4 MB feature-map slice that should stay in the L2) leave as non-temporal stores: -2..3 %; 0 = plain stores */
#endif
typedef float roi_f4 __attribute__((ext_vector_type(4)));
typedef int roi_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void roi_st4(float* p, float x, float y, float z, float w) {
  const roi_f4 v = {x, y, z, w};
  if (ROI_NT) __builtin_nontemporal_store(v, reinterpret_cast<roi_f4*>(p)); else *reinterpret_cast<roi_f4*>(p) = v;
}
__device__ __forceinline__ void roi_st4(int* p, int x, int y, int z, int w) {
  const roi_i4 v = {x, y, z, w};
  if (ROI_NT) __builtin_nontemporal_store(v, reinterpret_cast<roi_i4*>(p)); else *reinterpret_cast<roi_i4*>(p) = v;
}
#define ROI_ST4(p_, x_, y_, z_, w_) roi_st4((p_), (x_), (y_), (z_), (w_))
template <int C, int BW>
__device__ __forceinline__ void roi_col_shared(const __amdgpu_buffer_rsrc_t rsrc, const RoiBox& b, const int PH, const int PW, const int H, const int W,
                                               const int wstart, const unsigned lane_b, const int cbase, float* __restrict__ top,
                                               int* __restrict__ argmax, size_t out) {
  const float NEG = -3.402823466e+38f;
  int hs0 = __builtin_amdgcn_readfirstlane(min(max((int)floorf(0 * b.bin_h) + b.start_h, 0), H));
  int he0 = __builtin_amdgcn_readfirstlane(min(max((int)ceilf(1 * b.bin_h) + b.start_h, 0), H));
  float4 mv = make_float4(0.f, 0.f, 0.f, 0.f); int4 mi = make_int4(-1, -1, -1, -1);
  { const float i0 = he0 > hs0 ? NEG : 0.f; mv = make_float4(i0, i0, i0, i0); }
  int h = hs0;
  for (int ph = 0; ph < PH; ++ph, out += (size_t)PW * C) {
    int hs1 = 0x7fffffff, he1 = 0x7fffffff;
    if (ph + 1 < PH) {
      hs1 = __builtin_amdgcn_readfirstlane(min(max((int)floorf((ph + 1) * b.bin_h) + b.start_h, 0), H));
      he1 = __builtin_amdgcn_readfirstlane(min(max((int)ceilf((ph + 2) * b.bin_h) + b.start_h, 0), H));
    }
    const float i1 = (ph + 1 < PH && he1 > hs1) ? NEG : 0.f;
    float4 nv = make_float4(i1, i1, i1, i1); int4 ni = make_int4(-1, -1, -1, -1);
    h = max(h, hs0);
    unsigned voff = lane_b + (unsigned)((h * W + wstart) * C) * 4u;
    int idx = (h * W + wstart) * C;
    for (; h < he0; ++h, voff += (unsigned)W * C * 4u, idx += W * C) {
      float4 v[BW];
#pragma unroll
      for (int k = 0; k < BW; ++k) v[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, k * C * 4, 0));
#pragma unroll
      for (int k = 0; k < BW; ++k) {
        const int cand = idx + k * C;
        if (v[k].x > mv.x) { mv.x = v[k].x; mi.x = cand; }
        if (v[k].y > mv.y) { mv.y = v[k].y; mi.y = cand; }
        if (v[k].z > mv.z) { mv.z = v[k].z; mi.z = cand; }
        if (v[k].w > mv.w) { mv.w = v[k].w; mi.w = cand; }
      }
      if (h >= hs1) {                                         // (uniform) the row the next bin shares: the same registers, its own maxima
#pragma unroll
        for (int k = 0; k < BW; ++k) {
          const int cand = idx + k * C;
          if (v[k].x > nv.x) { nv.x = v[k].x; ni.x = cand; }
          if (v[k].y > nv.y) { nv.y = v[k].y; ni.y = cand; }
          if (v[k].z > nv.z) { nv.z = v[k].z; ni.z = cand; }
          if (v[k].w > nv.w) { nv.w = v[k].w; ni.w = cand; }
        }
      }
    }
    asm volatile("" : "+v"(mv.x), "+v"(mv.y), "+v"(mv.z), "+v"(mv.w));     // (see roi_row_steps)
    mi.x = mi.x < 0 ? -1 : mi.x + cbase; mi.y = mi.y < 0 ? -1 : mi.y + cbase + 1;
    mi.z = mi.z < 0 ? -1 : mi.z + cbase + 2; mi.w = mi.w < 0 ? -1 : mi.w + cbase + 3;
    ROI_ST4(top + out, mv.x, mv.y, mv.z, mv.w);
    ROI_ST4(argmax + out, mi.x, mi.y, mi.z, mi.w);
    mv = nv; mi = ni; hs0 = hs1; he0 = he1;
  }
}

template <int C>
__global__ void __launch_bounds__(256) roi_pool_fwd_cols(const float* __restrict__ data, int B, int H, int W,
                                                         const float* __restrict__ rois, int ncols, int PH, int PW,
                                                         float scale, float* __restrict__ top, int* __restrict__ argmax) {
  constexpr int XPS = 8 / (C / 256);                          // XCDs per slice
  static_assert(C % 256 == 0 && XPS >= 1 && XPS * (C / 256) == 8, "1, 2, 4 or 8 slices");
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, slice = xcd / XPS;
  const int colid = ((int)(blockIdx.x >> 3) * XPS + (xcd % XPS)) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (colid >= ncols) return;
  const int pw = colid % PW, r = colid / PW;
  const RoiBox b = roi_decode(rois + (size_t)r * 5, scale, PH, PW);
  const bool no_image = b.batch < 0 || b.batch >= B;     // (undefined in the reference; here such a ROI pools nothing)
  int wstart = (int)floorf(pw * b.bin_w), wend = (int)ceilf((pw + 1) * b.bin_w);
  wstart = min(max(wstart + b.start_w, 0), W); wend = min(max(wend + b.start_w, 0), W);
  wstart = __builtin_amdgcn_readfirstlane(wstart); wend = __builtin_amdgcn_readfirstlane(wend);
  const int bw = no_image ? 0 : wend - wstart;
  const float* img = data + (size_t)(no_image ? 0 : b.batch) * C * H * W + 256 * slice;
  const unsigned long long ib = reinterpret_cast<unsigned long long>(img);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ib), hi = __builtin_amdgcn_readfirstlane((unsigned)(ib >> 32));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                                        (int)(((unsigned)H * W * C - 256u * slice) * 4u), 0x00020000);
  const unsigned lane_b = 16u * lane;
  const int cbase = 256 * slice + 4 * lane;
  size_t out = (((size_t)r * PH) * PW + pw) * C + cbase;
  const bool shared = __builtin_amdgcn_readfirstlane((int)(b.bin_h >= 1.f)) != 0 && bw >= 1 && bw <= 8;
  if (shared) {
    switch (bw) {
      case 1: roi_col_shared<C, 1>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 2: roi_col_shared<C, 2>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 3: roi_col_shared<C, 3>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 4: roi_col_shared<C, 4>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 5: roi_col_shared<C, 5>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 6: roi_col_shared<C, 6>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      case 7: roi_col_shared<C, 7>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
      default: roi_col_shared<C, 8>(rsrc, b, PH, PW, H, W, wstart, lane_b, cbase, top, argmax, out); break;
    }
    return;
  }
  // small bins (a row may belong to three or more of them), wide windows, empty columns: bin by bin, eight columns per step, the last
  // step of a row shifted left to end at the row's end (its leading pixels were seen already: a repeated value never wins a strict >)
  for (int ph = 0; ph < PH; ++ph, out += (size_t)PW * C) {
    int hstart = (int)floorf(ph * b.bin_h), hend = (int)ceilf((ph + 1) * b.bin_h);
    hstart = min(max(hstart + b.start_h, 0), H); hend = min(max(hend + b.start_h, 0), H);
    hstart = __builtin_amdgcn_readfirstlane(hstart); hend = __builtin_amdgcn_readfirstlane(hend);
    const int rows = hend - hstart;
    const bool is_empty = rows <= 0 || bw <= 0;
    const float init = is_empty ? 0.f : -3.402823466e+38f;
    float4 mv = make_float4(init, init, init, init); int4 mi = make_int4(-1, -1, -1, -1);
    if (!is_empty) {
      unsigned vo = lane_b + (unsigned)((hstart * W + wstart) * C) * 4u; int ix = (hstart * W + wstart) * C;
      for (int h = 0; h < rows; ++h, vo += (unsigned)W * C * 4u, ix += W * C) {
        if (bw >= 8) {
          for (int w0 = 0; w0 < bw; w0 += 8) {
            const int ws = min(w0, bw - 8);
            roi_row_steps<C, 8>(rsrc, vo + (unsigned)ws * (C * 4u), ix + ws * C, 1, 0u, 0, mv, mi);
          }
        } else {
          for (int w0 = 0; w0 < bw; ++w0) roi_row_steps<C, 1>(rsrc, vo + (unsigned)w0 * (C * 4u), ix + w0 * C, 1, 0u, 0, mv, mi);
        }
      }
      mi.x = mi.x < 0 ? -1 : mi.x + cbase; mi.y = mi.y < 0 ? -1 : mi.y + cbase + 1;
      mi.z = mi.z < 0 ? -1 : mi.z + cbase + 2; mi.w = mi.w < 0 ? -1 : mi.w + cbase + 3;
    }
    ROI_ST4(top + out, mv.x, mv.y, mv.z, mv.w);
    ROI_ST4(argmax + out, mi.x, mi.y, mi.z, mi.w);
  }
}

// deterministic backward: one wave per input pixel (n, h, w), the four waves of a workgroup = a 2 x 2 block of pixels.
// A pooled element's (argmax, top_diff) vector is re-read by every pixel of its bin's window (2-3 wide after the floor /
// ceil overlap): neighbouring pixels walk the same ROIs and bins in the same order at the same time, so three of the four
// reads of a vector come from the CU's L1 instead of HBM.  The entries (roi, ph, pw) of a pixel are visited in the CPU
// kernel's order with the NEXT entry's eight 16-byte loads in flight while the current one is added.
// The entries (roi, ph, pw) whose bin window contains pixel (n, h, w), in the CPU kernel's order (roi_pooling_op.cc:405-447).
// The ROI list is scanned 64 at a time -- each lane decodes one ROI and tests its box, a ballot keeps the hits -- and only a
// hit is decoded again (uniformly) for its bin range: a scalar scan of all R boxes per wave was most of the kernel's time.
struct PixelEntries {
  const float* rois; int R, PH, PW, n, h, w, lane; float scale;
  int r0 = -64, r = 0, ph = 0, pw = 0, phend = 0, pwstart = 0, pwend = 0;
  unsigned long long hits = 0ull;
  __device__ __forceinline__ PixelEntries(const float* rois_, int R_, int PH_, int PW_, float scale_, int n_, int h_, int w_, int lane_)
      : rois(rois_), R(R_), PH(PH_), PW(PW_), n(n_), h(h_), w(w_), lane(lane_), scale(scale_) {}
  // the next entry's pooled-element index (r * PH + ph) * PW + pw, or false when the list is exhausted (and stays so)
  __device__ __forceinline__ bool next(size_t& bin) {
    for (;;) {
      if (ph < phend) {
        if (pw < pwend) { bin = ((size_t)r * PH + ph) * PW + pw; ++pw; return true; }
        ++ph; pw = pwstart;
        continue;
      }
      while (hits == 0ull) {
        r0 += 64;
        if (r0 >= R) { r0 = R; return false; }
        bool ok = false;
        if (r0 + lane < R) {
          const float* q = rois + (size_t)(r0 + lane) * 5;
          const int sw = (int)roundf(q[1] * scale), sh = (int)roundf(q[2] * scale);
          const int ew = (int)roundf(q[3] * scale), eh = (int)roundf(q[4] * scale);
          ok = ((int)q[0] == n) && w >= sw && w <= ew && h >= sh && h <= eh;
        }
        hits = __ballot(ok);
      }
      r = r0 + __builtin_ctzll(hits);
      hits &= hits - 1ull;
      const RoiBox b = roi_decode(rois + (size_t)r * 5, scale, PH, PW);
      int phs = (int)floorf((float)(h - b.start_h) / b.bin_h), phe = (int)ceilf((float)(h - b.start_h + 1) / b.bin_h);   // :428-431
      int pws = (int)floorf((float)(w - b.start_w) / b.bin_w), pwe = (int)ceilf((float)(w - b.start_w + 1) / b.bin_w);
      ph = min(max(phs, 0), PH); phend = min(max(phe, 0), PH);
      pwstart = min(max(pws, 0), PW); pwend = min(max(pwe, 0), PW);
      pw = pwstart;
    }
  }
};

template <bool VEC4>
__global__ void __launch_bounds__(256) roi_pool_bwd_pixel(const float* __restrict__ top_diff, const int* __restrict__ argmax,
                                                          const float* __restrict__ rois, int B, int H, int W, int C, int R,
                                                          int PH, int PW, float scale, float* __restrict__ bottom_diff) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bw2 = (W + 1) >> 1, bh2 = (H + 1) >> 1;
  const int blk = blockIdx.x;
  const int n = blk / (bw2 * bh2), rem = blk - n * (bw2 * bh2);
  const int h = 2 * (rem / bw2) + (wv >> 1), w = 2 * (rem % bw2) + (wv & 1);
  if (n >= B || h >= H || w >= W) return;
  const long long pix = ((long long)n * H + h) * W + w;
  const int here = (h * W + w) * C;
  for (int c0 = 0; c0 < C; c0 += 1024) {     // 1024 channels per outer pass
    float4 acc4[4]; float acc1[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
    PixelEntries walk(rois, R, PH, PW, scale, n, h, w, lane);
    auto next = [&](size_t& o) -> bool {
      size_t bin;
      if (!walk.next(bin)) return false;
      o = bin * C + c0;
      return true;
    };
    if (VEC4) {
      int4 am[4], am_n[4]; float4 g[4], g_n[4];
      auto load = [&](size_t o, int4 (&a_)[4], float4 (&g_)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 256 * i + 4 * lane;
          const bool ok = c0 + c < C;
          a_[i] = ok ? *reinterpret_cast<const int4*>(argmax + o + c) : make_int4(-1, -1, -1, -1);
          g_[i] = ok ? *reinterpret_cast<const float4*>(top_diff + o + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      size_t o = 0;
      bool have = next(o);
      if (have) load(o, am, g);
      while (have) {
        const bool have_n = next(o);
        if (have_n) load(o, am_n, g_n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int want = here + c0 + 256 * i + 4 * lane;
          if (am[i].x == want) acc4[i].x += g[i].x;
          if (am[i].y == want + 1) acc4[i].y += g[i].y;
          if (am[i].z == want + 2) acc4[i].z += g[i].z;
          if (am[i].w == want + 3) acc4[i].w += g[i].w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { am[i] = am_n[i]; g[i] = g_n[i]; }
        have = have_n;
      }
    } else {
      size_t o = 0;
      while (next(o)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = 64 * i + lane;
          if (c0 + c < C && argmax[o + c] == here + c0 + c) acc1[i] += top_diff[o + c];
        }
      }
    }
    float* out = bottom_diff + (size_t)pix * C + c0;
    if (VEC4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 256 * i + 4 * lane;
        if (c0 + c < C) *reinterpret_cast<float4*>(out + c) = acc4[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 64 * i + lane;
        if (c0 + c < C) out[c] = acc1[i];
      }
    }
  }
}

// Deterministic backward for C a multiple of 256 and PH, PW <= 255: one wave per (2 x 2 block of pixels, 256-channel slice), a
// workgroup = four neighbouring blocks of one slice, consecutive tiles on one XCD.  The per-pixel walk above fetched every
// pooled vector once per pixel of its bin's window (10.9 GB of loads, 3.6 GB from HBM, for 0.8 GB of input) and spent most of
// its instructions in the scalar entry iterator.  Here (a) a vector is loaded once for the four pixels of a block (their entry
// sets mostly coincide) and compared against each, and (b) the bookkeeping runs across the lanes: 64 ROIs per step, lane j
// decodes ROI r0 + j and the bin ranges the reference gives each row and column of the block (:417-431); a prefix sum over the
// ROIs' bounding ranges places the step's bins, in (roi, ph, pw) order, one per lane, each with the four pixels it is valid for;
// the walk is then two readlanes, two loads and the compare-adds per bin.  Bit-exact: each (pixel, channel) sum is formed by one
// lane, in the reference's order, in a register, from exactly the entries the reference's window test admits.
// The NW waves of a workgroup are channel slices of ONE block: their bookkeeping is identical, so they share it -- in a round
// of NW 64-ROI steps wave w does step w's scan and places its bins into the round's list in LDS (bin << 4 | valid pixels, in
// (roi, ph, pw) order), and every wave then walks the list for its own channels as ONE stream: the load pipeline is filled and
// drained once per round, not once per 64 bins.  (A round with more than RPB_CAP bins -- many tiny ROIs on one block -- is
// redone step by step by each wave: correct, just not shared.)
#ifndef ROI_BX
#define ROI_BX 0   /* ablation mask of measurement builds (WRONG results, durations only): 1 = no compare-adds, 2 = no loads of the pooled vectors */
#endif
constexpr int RPB_CAP = 512;        // bins of a round that go through LDS (more: every wave redoes the round's steps itself)
// CPL channels per lane (4: 16-byte loads, 256-channel slices; 2: 8-byte loads, 128-channel slices), NW waves = slices per workgroup
template <int D, int CPL, int NW>
__global__ void __launch_bounds__(64 * NW) roi_pool_bwd_block(const float* __restrict__ top_diff, const int* __restrict__ argmax,
                                                          const float* __restrict__ rois, int B, int H, int W, int C, int R,
                                                          int PH, int PW, float scale, int nwork, float* __restrict__ bottom_diff) {
  constexpr int SLC = 64 * CPL;                               // channels of a slice
  __shared__ int sList[2][RPB_CAP];                           // [round parity][bin of the round]: bin << 4 | valid pixels
  __shared__ int sTotal[2][NW];                               // bins of each step of the round                                // bins of the step, or -1: redo it in full
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int per = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (L >= nwork) return;
  const int nsl = C / SLC, nsg = (nsl + NW - 1) / NW, tw_n = (W + 3) >> 2, th_n = (H + 3) >> 2;
  const int sl = NW * (L % nsg) + wv;                          // this wave's slice (past the last one: bookkeeping only)
  int t = L / nsg;
  const int sub = t & 3;                                      // consecutive work items: the four blocks of a 4 x 4 tile
  t >>= 2;
  const int n = t / (th_n * tw_n);
  t -= n * (th_n * tw_n);
  const int hb = (t / tw_n) * 4 + 2 * (sub >> 1), wb = (t % tw_n) * 4 + 2 * (sub & 1);   // the block's first pixel
  if (hb >= H || wb >= W) return;                             // (the whole workgroup)
  const bool live = sl < nsl;
  const bool row1 = hb + 1 < H, col1 = wb + 1 < W;
  const int c = live ? SLC * sl + CPL * lane : 0;
  float* out = bottom_diff + (((size_t)n * H + hb) * W + wb) * C + c;
  const size_t o_row = (size_t)W * C;
  float acc[4][CPL];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[p][i] = 0.f;
  typedef int ivec __attribute__((ext_vector_type(CPL)));
  typedef float fvec __attribute__((ext_vector_type(CPL)));
  const int* am_lane = argmax + c;
  const float* g_lane = top_diff + c;
  const int pix0 = (hb * W + wb) * C;                         // argmax - c of pixel 0's elements; + C, + W C, + W C + C for the others

  // lane j: ROI r0 + j.  rows / cols: [lo, hi) bin range of each row / column of the block as four bytes (empty = 0, 0);
  // box: the bounding range of the block (ph_lo, pw_lo, width); cnt = its number of bins; pre = exclusive prefix of cnt
  int cnt = 0, pre = 0, total = 0;
  unsigned rows = 0u, cols = 0u, box = 0u;
  unsigned long long hm = 0ull;
  auto scan = [&](int r0) {
    cnt = 0; rows = 0u; cols = 0u; box = 0u; total = 0;
    if (r0 + lane < R) {
      const RoiBox b = roi_decode(rois + (size_t)(r0 + lane) * 5, scale, PH, PW);
      if (b.batch == n) {
        int lo[4], hi[4];                                     // rows 0, 1, columns 0, 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int h = hb + i, w = wb + i;
          int phs = (int)floorf((float)(h - b.start_h) / b.bin_h), phe = (int)ceilf((float)(h - b.start_h + 1) / b.bin_h);
          int pws = (int)floorf((float)(w - b.start_w) / b.bin_w), pwe = (int)ceilf((float)(w - b.start_w + 1) / b.bin_w);
          phs = min(max(phs, 0), PH); phe = min(max(phe, 0), PH);
          pws = min(max(pws, 0), PW); pwe = min(max(pwe, 0), PW);
          const bool rok = h < H && h >= b.start_h && h <= b.end_h && phe > phs;
          const bool cok = w < W && w >= b.start_w && w <= b.end_w && pwe > pws;
          lo[i] = rok ? phs : 0; hi[i] = rok ? phe : 0;
          lo[2 + i] = cok ? pws : 0; hi[2 + i] = cok ? pwe : 0;
        }
        const int ph_lo = hi[0] == 0 ? lo[1] : hi[1] == 0 ? lo[0] : min(lo[0], lo[1]), ph_hi = max(hi[0], hi[1]);
        const int pw_lo = hi[2] == 0 ? lo[3] : hi[3] == 0 ? lo[2] : min(lo[2], lo[3]), pw_hi = max(hi[2], hi[3]);
        cnt = (ph_hi - ph_lo) * (pw_hi - pw_lo);              // (0 when the block has no row or no column in this ROI)
        rows = (unsigned)lo[0] | ((unsigned)hi[0] << 8) | ((unsigned)lo[1] << 16) | ((unsigned)hi[1] << 24);
        cols = (unsigned)lo[2] | ((unsigned)hi[2] << 8) | ((unsigned)lo[3] << 16) | ((unsigned)hi[3] << 24);
        box = (unsigned)ph_lo | ((unsigned)pw_lo << 8) | ((unsigned)max(pw_hi - pw_lo, 1) << 16);
      }
    }
    hm = __ballot(cnt > 0);
    if (hm == 0ull) return;
    pre = cnt;                                                // inclusive prefix sum over the lanes
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const int up = __shfl_up(pre, sft); if (lane >= sft) pre += up; }
    total = __builtin_amdgcn_readlane(pre, 63);
    pre -= cnt;
  };
  // bin base + lane of the scanned step: which ROI, which bin of its bounding range, which of the four pixels take it
  auto place = [&](int r0, int base, int& binv, int& validv) {
    int k = 0, e_r = 0;
    unsigned e_rows = 0u, e_cols = 0u, e_box = 1u << 16;
    for (unsigned long long hh = hm; hh != 0ull; hh &= hh - 1ull) {
      const int j = __builtin_ctzll(hh);
      const int pj = __builtin_amdgcn_readlane(pre, j), cj = __builtin_amdgcn_readlane(cnt, j);
      if (pj + cj <= base) continue;
      if (pj >= base + 64) break;
      const unsigned d = (unsigned)(lane + base - pj);
      if (d < (unsigned)cj) {
        k = (int)d; e_r = r0 + j;
        e_rows = (unsigned)__builtin_amdgcn_readlane((int)rows, j); e_cols = (unsigned)__builtin_amdgcn_readlane((int)cols, j);
        e_box = (unsigned)__builtin_amdgcn_readlane((int)box, j);
      }
    }
    // k / width for small integers: (k + 0.5) / width is at least 0.5 / 255 away from an integer, the rcp's error far below
    const int bww = (int)(e_box >> 16);
    const int qd = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)bww));
    const int ph = (int)(e_box & 255u) + qd, pw = (int)((e_box >> 8) & 255u) + (k - qd * bww);
    binv = (e_r * PH + ph) * PW + pw;
    const bool r0ok = ph >= (int)(e_rows & 255u) && ph < (int)((e_rows >> 8) & 255u);
    const bool r1ok = ph >= (int)((e_rows >> 16) & 255u) && ph < (int)(e_rows >> 24);
    const bool c0ok = pw >= (int)(e_cols & 255u) && pw < (int)((e_cols >> 8) & 255u);
    const bool c1ok = pw >= (int)((e_cols >> 16) & 255u) && pw < (int)(e_cols >> 24);
    validv = (r0ok && c0ok ? 1 : 0) | (r0ok && c1ok ? 2 : 0) | (r1ok && c0ok ? 4 : 0) | (r1ok && c1ok ? 8 : 0);
  };
  // the pooled vector of one bin against the block's pixels: compare-into-EXEC, add under it, EXEC back to all lanes (every lane
  // of the wave is live here) -- two vector instructions per (pixel, component) instead of three; pixels the reference's window
  // test excludes for this bin (v) are skipped by uniform branches
  auto add_bin = [&](const ivec& a_, const fvec& g_, const int v) {
    if (ROI_BX & 1) { asm volatile("" :: "v"(a_), "v"(g_)); return; }
    int ax[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) ax[i] = a_[i] - c - i;      // = pixel * C where it is this lane's channel
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!((v >> p) & 1)) continue;
      const int wq = pix0 + (p & 1) * C + (p >> 1) * W * C;
#pragma unroll
      for (int i = 0; i < CPL; ++i)
        asm volatile("v_cmpx_eq_i32_e32 vcc, %2, %1\n\tv_add_f32_e32 %0, %0, %3\n\ts_mov_b64 exec, -1"
                     : "+v"(acc[p][i]) : "v"(ax[i]), "s"(wq), "v"(g_[i]) : "vcc");
    }
  };
  // the walk over a round's T bins (LDS list): groups of D, the next group's loads in flight while one is added; the loads are
  // unconditional -- a slot past the end re-reads element 0 with no pixel valid -- so that the waits are counted, not drained
  // (D = 2, 4, 8 measured 628 / 643 / 652 us at the contract shape: loads in flight are not what bounds the walk)
  auto walk_list = [&](const int* __restrict__ list, const int T) {
    ivec amA[D], amB[D]; fvec gA[D], gB[D]; int vA[D], vB[D];
    auto issue = [&](int e, ivec (&a_)[D], fvec (&g_)[D], int (&v_)[D]) {
      int pk[D];
#pragma unroll
      for (int q = 0; q < D; ++q) pk[q] = list[min(e + q, RPB_CAP - 1)];
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const int pq = e + q < T ? __builtin_amdgcn_readfirstlane(pk[q]) : 0;
        v_[q] = pq & 15;
        const size_t o = (size_t)(pq >> 4) * C;
        if (ROI_BX & 2) { for (int i_ = 0; i_ < CPL; ++i_) { a_[q][i_] = (int)o + i_; g_[q][i_] = 1.f; } }
        else { a_[q] = *reinterpret_cast<const ivec*>(am_lane + o); g_[q] = *reinterpret_cast<const fvec*>(g_lane + o); }
      }
    };
    issue(0, amA, gA, vA);
    for (int e = 0; e < T; e += 2 * D) {
      issue(e + D, amB, gB, vB);
#pragma unroll
      for (int q = 0; q < D; ++q) add_bin(amA[q], gA[q], vA[q]);
      issue(e + 2 * D, amA, gA, vA);
#pragma unroll
      for (int q = 0; q < D; ++q) add_bin(amB[q], gB[q], vB[q]);
    }
  };
  // ... and over m <= 64 bins placed in this wave's registers (the unshared path)
  auto walk_regs = [&](const int binv, const int validv, const int m) {
    for (int e = 0; e < m; ++e) {
      const size_t o = (size_t)__builtin_amdgcn_readlane(binv, e) * C;
      const ivec a_ = *reinterpret_cast<const ivec*>(am_lane + o);
      const fvec g_ = *reinterpret_cast<const fvec*>(g_lane + o);
      add_bin(a_, g_, __builtin_amdgcn_readlane(validv, e));
    }
  };

  int par = 0;
  for (int rr = 0; rr < R; rr += 64 * NW, par ^= 1) {
    // this wave's step of the round: scan; its bins' place in the round's list is known when every wave has scanned
    const int r0 = rr + 64 * wv;
    total = 0;
    if (r0 < R) scan(r0);
    if (lane == 0) sTotal[par][wv] = total;
    __syncthreads();
    int off = 0, T = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) { const int tw = sTotal[par][w2]; off += w2 < wv ? tw : 0; T += tw; }
    const bool shared = T <= RPB_CAP && ((long long)R * PH * PW < (1ll << 27));
    if (shared)
      for (int base = 0; base < total; base += 64) {
        int binv, validv;
        place(r0, base, binv, validv);
        if (base + lane < total) sList[par][off + base + lane] = (binv << 4) | validv;
      }
    __syncthreads();       // (a list of this parity is rewritten two rounds on, behind the next round's barriers)
    if (!live || T == 0) continue;
    if (shared) walk_list(sList[par], T);
    else
      for (int w2 = 0; w2 < NW; ++w2) {            // too many bins for the list: every step in full, here
        const int q0 = rr + 64 * w2;
        if (q0 >= R) break;
        scan(q0);
        for (int base = 0; base < total; base += 64) {
          int binv, validv;
          place(q0, base, binv, validv);
          walk_regs(binv, validv, min(64, total - base));
        }
      }
  }
  if (!live) return;
  auto put = [&](float* dst, const float (&a_)[CPL]) {
    fvec v;
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = a_[i];
    *reinterpret_cast<fvec*>(dst) = v;
  };
  put(out, acc[0]);
  if (col1) put(out + C, acc[1]);
  if (row1) put(out + o_row, acc[2]);
  if (row1 && col1) put(out + o_row + C, acc[3]);
}

__global__ void __launch_bounds__(256) roi_pool_bwd_atomic(const float* __restrict__ top_diff, const int* __restrict__ argmax,
                                                           const float* __restrict__ rois, long long total, int per_roi,
                                                           long long image_elems, int B, float* __restrict__ bottom_diff) {
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < total; b += (long long)gridDim.x * 256) {
    const int idx = argmax[b];
    if (idx < 0 || idx >= image_elems) continue;            // (-1 = an empty bin; an index past the image: not ours to write)
    const long long r = b / per_roi;
    const float bf = rois[r * 5];
    const int bi = bf == bf ? (int)bf : -1;
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
  if ((C == 256 || C == 512 || C == 1024 || C == 2048) && (long long)R * pooled_h <= 0x0fffffffLL) {
    const int nrows = R * pooled_h;
    const int xps = 8 / (C / 256);                             // a workgroup = 4 rows of one slice; 8 workgroups = 4 xps rows of every slice
    const unsigned g = (unsigned)(8 * ((nrows + 4 * xps - 1) / (4 * xps)));
    static const bool rows1 = getenv("GNET_ROI_FWD_ROWS1") != nullptr;     // measurement only: round 3's kernel (scalar walkers)
    static const bool rows2 = getenv("GNET_ROI_FWD_ROWS2") != nullptr;     // measurement only: rows of bins (no shared loads)
    if (!rows1 && !rows2 && (long long)R * pooled_w <= 0x0fffffffLL) {
      const int ncols = R * pooled_w;
      const unsigned gc = (unsigned)(8 * ((ncols + 4 * xps - 1) / (4 * xps)));
#define GNET_COLS(C_) roi_pool_fwd_cols<C_><<<gc, 256, 0, (hipStream_t)stream>>>(bottom_data, B, H, W, bottom_rois, ncols, pooled_h, pooled_w, spatial_scale, top_data, argmax)
      if (C == 256) GNET_COLS(256); else if (C == 512) GNET_COLS(512); else if (C == 1024) GNET_COLS(1024); else GNET_COLS(2048);
#undef GNET_COLS
      return launch_status();
    }
#define GNET_ROWS(C_) do { if (rows1) roi_pool_fwd_rows<C_><<<g, 256, 0, (hipStream_t)stream>>>(bottom_data, B, H, W, bottom_rois, nrows, pooled_h, pooled_w, spatial_scale, top_data, argmax); \
                           else roi_pool_fwd_rows2<C_><<<g, 256, 0, (hipStream_t)stream>>>(bottom_data, B, H, W, bottom_rois, nrows, pooled_h, pooled_w, spatial_scale, top_data, argmax); } while (0)
    if (C == 256) GNET_ROWS(256); else if (C == 512) GNET_ROWS(512); else if (C == 1024) GNET_ROWS(1024); else GNET_ROWS(2048);
#undef GNET_ROWS
  }
  else if ((C & 3) == 0)
    roi_pool_fwd<true><<<grid, 256, 0, (hipStream_t)stream>>>(bottom_data, B, H, W, C, bottom_rois, nbins, pooled_h, pooled_w, spatial_scale, top_data, argmax);
  else
    roi_pool_fwd<false><<<grid, 256, 0, (hipStream_t)stream>>>(bottom_data, B, H, W, C, bottom_rois, nbins, pooled_h, pooled_w, spatial_scale, top_data, argmax);
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
  const long long blocks = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);       // one workgroup per 2 x 2 block of pixels
  if (blocks > 0x7fffffffLL) return GNET_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)blocks;
  if ((C & 255) == 0 && pooled_h <= 255 && pooled_w <= 255 && (long long)R * pooled_h * pooled_w < 0x7fffffffLL) {
    // 4 channels per lane, 4 slices per workgroup.  Measured at the contract shape: 2 channels per lane x 8 waves 707 us, 8 x 2
    // 719 us, 8 x 4 (two of the waves bookkeeping only) 683 us, this one 617 us
    constexpr int CPL = 4, NW = 4;
    const long long nwork = (long long)B * ((H + 3) / 4) * ((W + 3) / 4) * 4 * ((C / (64 * CPL) + NW - 1) / NW);   // workgroup = a 2 x 2 block, NW slices
    if (nwork > 0x7ffffff0LL) return GNET_ERR_UNSUPPORTED;
    const unsigned g8 = 8u * (unsigned)((nwork + 7) / 8);
    roi_pool_bwd_block<2, CPL, NW><<<g8, 64 * NW, 0, s>>>(top_diff, argmax, bottom_rois, B, H, W, C, R, pooled_h, pooled_w, spatial_scale,
                                                         (int)nwork, bottom_diff);
  }
  else if ((C & 3) == 0)
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
