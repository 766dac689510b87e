// Probe for DESIGN.md "what comes next" 6: fp32 products emulated on the bf16 pipe by splitting every operand into three bf16 terms
// (x = hi + mid + lo EXACTLY: two truncations and a remainder) and issuing six products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid)
// with fp32 accumulation.  Two questions, answered on the MI355X:
//   accuracy    C = A[32,64] . B[64,64] (activations ~ |N(0,1)| rectified, weights xavier) by v_mfma_f32_32x32x2_f32 and by the six-product
//               form, both against an fp64 reference: error relative to sum |a||b| per element
//   throughput  a tile loop shaped like edge_fwd_w's (per 32-edge tile: 96 fp32 MFMAs + 330 vector instructions) against the emulated
//               form (72 bf16 MFMAs of 32x32x16 + 330 + the split of the 32 layer-1 activations per lane), three waves per SIMD
// Build / run:  hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o gpurun_out/bf16x3_probe && gpurun_out/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// x = hi + mid + lo exactly (normal numbers): hi, mid by truncation to the upper 16 bits, lo the 8-bit remainder
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned xb = __float_as_uint(x);
  hi = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);
  lo = __float_as_uint(r2);            // at most 8 significant bits left: its upper half IS the bf16
}
// two fp32 bit patterns -> their upper halves packed (element 0 in the low half)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- accuracy: one wave, C[32,64] = A[32,64] . B[64,64]; Bt = B transposed [64 n][64 k]
__global__ void __launch_bounds__(64) acc_fp32(const float* A, const float* Bt, float* C) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int nb = 0; nb < 2; ++nb) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 64 + k + h], Bt[(32 * nb + r) * 64 + k + h], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[crow(i, h) * 64 + 32 * nb + r] = acc[i];
  }
}
__global__ void __launch_bounds__(64) acc_bf16x3(const float* A, const float* Bt, float* C, int nprod) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int nb = 0; nb < 2; ++nb) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < 64; k0 += 16) {
      unsigned ah[8], am[8], al[8], bh[8], bm[8], bl[8];
      for (int i = 0; i < 8; ++i) {
        split3(A[r * 64 + k0 + 8 * h + i], ah[i], am[i], al[i]);
        split3(Bt[(32 * nb + r) * 64 + k0 + 8 * h + i], bh[i], bm[i], bl[i]);
      }
      u32x4 pa[3], pb[3];
      for (int i = 0; i < 4; ++i) {
        pa[0][i] = pack_hi(ah[2 * i], ah[2 * i + 1]); pa[1][i] = pack_hi(am[2 * i], am[2 * i + 1]); pa[2][i] = pack_hi(al[2 * i], al[2 * i + 1]);
        pb[0][i] = pack_hi(bh[2 * i], bh[2 * i + 1]); pb[1][i] = pack_hi(bm[2 * i], bm[2 * i + 1]); pb[2][i] = pack_hi(bl[2 * i], bl[2 * i + 1]);
      }
      // smallest terms first
      const int order[9][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}, {0, 2}, {2, 0}, {0, 1}, {1, 0}, {0, 0}};
      for (int q = 9 - nprod; q < 9; ++q) {
        bf16x8 av, bv;
        memcpy(&av, &pa[order[q][0]], 16); memcpy(&bv, &pb[order[q][1]], 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) C[crow(i, h) * 64 + 32 * nb + r] = acc[i];
  }
}

// ---- throughput: per "tile" NM MFMAs + NV filler vector instructions (+ the split of 32 values per lane when SPLIT)
#define VALU8(x) \
  asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(ca), "v"(cb))

template <bool BF16>
__global__ void __launch_bounds__(256, 3) tile_loop(float* out, int tiles, int nv8) {
  f32x16 h1a, h1b, h2a, h2b;
  for (int i = 0; i < 16; ++i) { h1a[i] = 0.001f * threadIdx.x + i; h1b[i] = 0.002f * threadIdx.x - i; h2a[i] = 0.f; h2b[i] = 0.f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  const float ca = 0.999f, cb = 0.001f;
  u32x4 wq[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) wq[i][j] = 0x3f803f80u + threadIdx.x + i + j;
  for (int t = 0; t < tiles; ++t) {
    for (int j = 0; j < nv8; ++j) { VALU8(x); }
    if (!BF16) {
      // layer 1 (32 MFMAs) + layer 2 (64 MFMAs): operands from registers (the kernel reads its B operands from LDS: not modelled)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        h1a = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k & 7], ca, h1a, 0, 0, 0);
        h1b = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(k + 1) & 7], cb, h1b, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(h1a[k], ca, h2a, 0, 0, 0);
        h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(h1a[k], cb, h2b, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        h2a = __builtin_amdgcn_mfma_f32_32x32x2f32(h1b[k], ca, h2a, 0, 0, 0);
        h2b = __builtin_amdgcn_mfma_f32_32x32x2f32(h1b[k], cb, h2b, 0, 0, 0);
      }
    } else {
      // layer 1: 2 feature blocks x 2 k-steps x 6 products = 24 MFMAs (operands pre-split: P and the weights)
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        bf16x8 av, bv;
        memcpy(&av, &wq[q & 3], 16); memcpy(&bv, &wq[(q + 1) & 3], 16);
        h1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, h1a, 0, 0, 0);
        h1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, h1b, 0, 0, 0);
      }
      // split the 32 activations of the lane: k-step j of block a / b = registers 8 j .. 8 j + 7
      u32x4 ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned hi[8], mid[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split3(j < 2 ? h1a[8 * j + i] : h1b[8 * (j - 2) + i], hi[i], mid[i], lo[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ph[j][i] = pack_hi(hi[2 * i], hi[2 * i + 1]); pm[j][i] = pack_hi(mid[2 * i], mid[2 * i + 1]); pl[j][i] = pack_hi(lo[2 * i], lo[2 * i + 1]); }
      }
      // layer 2: 2 column blocks x 4 k-steps x 6 products = 48 MFMAs
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 ah, am, al, w0, w1, w2;
        memcpy(&ah, &ph[j], 16); memcpy(&am, &pm[j], 16); memcpy(&al, &pl[j], 16);
        memcpy(&w0, &wq[j], 16); memcpy(&w1, &wq[(j + 1) & 3], 16); memcpy(&w2, &wq[(j + 2) & 3], 16);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, w0, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, w1, h2b, 0, 0, 0);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w2, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w0, h2b, 0, 0, 0);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w1, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w2, h2b, 0, 0, 0);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w0, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w1, h2b, 0, 0, 0);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w1, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w2, h2b, 0, 0, 0);
        h2a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w0, h2a, 0, 0, 0); h2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w1, h2b, 0, 0, 0);
      }
    }
    // keep the values bounded and the chain alive
#pragma unroll
    for (int i = 0; i < 16; ++i) { h1a[i] = h2a[i] * 1e-9f + 0.5f; h1b[i] = h2b[i] * 1e-9f + 0.25f; h2a[i] = 0.f; h2b[i] = 0.f; }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += h1a[i] + h1b[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  // ---- accuracy
  std::vector<float> A(32 * 64), Bt(64 * 64), C0(32 * 64), C1(32 * 64);
  srand(7);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  auto gauss = [&]() { float s = 0.f; for (int i = 0; i < 12; ++i) s += rnd(); return s - 6.f; };
  for (auto& v : A) { v = gauss(); if (v < 0.f) v = 0.f; }
  const float lim = sqrtf(6.f / 128.f);
  for (auto& v : Bt) v = (2.f * rnd() - 1.f) * lim;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, Bt.size() * 4)); CK(hipMalloc(&dC, C0.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice));
  auto report = [&](const char* name, const std::vector<float>& C) {
    double worst = 0., worst_abs = 0.;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 64; ++n) {
      double ref = 0., mag = 0.;
      for (int k = 0; k < 64; ++k) { ref += (double)A[m * 64 + k] * Bt[n * 64 + k]; mag += fabs((double)A[m * 64 + k] * Bt[n * 64 + k]); }
      const double e = fabs(C[m * 64 + n] - ref);
      if (mag > 0. && e / mag > worst) worst = e / mag;
      if (e > worst_abs) worst_abs = e;
    }
    printf("accuracy %-28s max |err| / sum|a||b| = %.3e   max |err| = %.3e\n", name, worst, worst_abs);
  };
  acc_fp32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C0.data(), dC, C0.size() * 4, hipMemcpyDeviceToHost)); report("fp32 MFMA 32x32x2", C0);
  for (int np : {3, 6, 9}) {
    acc_bf16x3<<<1, 64>>>(dA, dB, dC, np); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C1.data(), dC, C1.size() * 4, hipMemcpyDeviceToHost));
    char nm[64]; snprintf(nm, sizeof nm, "bf16 x 3 terms, %d products", np); report(nm, C1);
  }
  // ---- throughput: 768 workgroups x 4 waves = three waves per SIMD on 256 CUs
  float* dO; CK(hipMalloc(&dO, 768 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = 2000;
  for (int nv8 : {0, 41}) {          // 41 x 8 = 328 filler vector instructions per tile (edge_fwd_w: 330)
    for (int bf = 0; bf < 2; ++bf) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (bf) tile_loop<true><<<768, 256>>>(dO, tiles, nv8); else tile_loop<false><<<768, 256>>>(dO, tiles, nv8);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      // per SIMD: 3 waves x tiles; cycles at 2.4 GHz per tile and SIMD-share
      printf("tile loop %-22s filler %3d vector instr / tile: %.3f ms = %.0f cycles (2.4 GHz) per tile and wave, %.0f per tile and SIMD share\n",
             bf ? "72 bf16 MFMAs + split" : "96 fp32 MFMAs", 8 * nv8, best, best * 1e-3 * 2.4e9 / tiles, best * 1e-3 * 2.4e9 / tiles / 3.0);
    }
  }
  return 0;
}
