// Probe for DESIGN.md "what comes next" 6: can pw_fwd2's fc2 (a 256 x 32 weight slice per wave, RESIDENT in 128 registers as fp32) move to
// the bf16 pipe?  Three bf16 terms of the slice are 192 registers -- too many beside the accumulators.  Variant measured here: the high and
// the middle term stay resident (128 registers), the low term (used by ONE of the six products) is streamed from L2 through a small ring,
// the h1 tile's three terms come from LDS (16-byte reads).  Per wave and 32-edge tile, fc2 only:
//   fp32   16 k-steps x (one 16-byte LDS read + 8 v_mfma_f32_32x32x2_f32)                        = 128 MFMAs, 8192 pipe cycles
//   bf16   16 k-steps x (three 16-byte LDS reads + one 16-byte L2 load + 6 v_mfma_f32_32x32x16_bf16) =  96 MFMAs, 3072 pipe cycles
// One 8-wave workgroup per CU, one barrier per tile (as pw_fwd2).  Operand values are arbitrary: only the schedule is measured.
// Build / run:  hipcc --offload-arch=gfx950 -O3 tools/pw_bf16x3_probe.hip -o tools/pw_bf16x3_probe.bin && tools/pw_bf16x3_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int LDH = 132;        // words per edge row of an h1 term tile [32 edges][128 words = 256 bf16] + pad

template <int MODE>            // 0 = fp32, 1 = bf16 x 3 with the low weight term from L2, 2 = the same with all three terms resident (upper bound)
__global__ void __launch_bounds__(512) fc2_loop(float* out, const u32x4* __restrict__ wlo, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < 3 * 32 * LDH; i += 512) smem[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float s = 0.f;
  if (MODE == 0) {
    f32x4 w[32];                                     // the wave's 256 x 32 fp32 slice: 128 registers
    for (int i = 0; i < 32; ++i) { w[i] = f32x4{0.01f * i, 0.02f, 0.03f, 0.001f * lane}; asm volatile("" : "+v"(w[i])); }   // (opaque: really resident)
    const float* hb = reinterpret_cast<const float*>(smem) + col * LDH + 4 * half;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int f = 0; f < 32; ++f) {                 // 32 fragments of 8 k: one 16-byte read, 4 MFMAs
        const f32x4 b = *reinterpret_cast<const f32x4*>(hb + 8 * (f & 15) + (f >> 4) * 32 * LDH);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[f].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[f].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[f].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[f].w, b.w, acc, 0, 0, 0);
      }
      for (int i = 0; i < 16; ++i) { s += acc[i]; acc[i] = 0.f; }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    u32x4 wh[16], wm[16];                            // high and middle terms of the slice: 128 registers
    u32x4 wl[16];                                    // (MODE 2 only: the low term resident as well)
    for (int i = 0; i < 16; ++i) { wh[i] = u32x4{0x3f803f80u + i, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u + lane}; wm[i] = wh[i] ^ 0x00100010u; if (MODE == 2) wl[i] = wh[i] ^ 0x00010001u;
      asm volatile("" : "+v"(wh[i]), "+v"(wm[i])); if (MODE == 2) asm volatile("" : "+v"(wl[i])); }
    const unsigned* hb = smem + col * LDH + half * 4;                 // term t: + t * 32 * LDH; k-step j: + 8 j
    const u32x4* lp = wlo + (size_t)(wave * 16) * 64 + lane;          // the wave's low-term slice: [16 k-steps][64 lanes] 16-byte words
    constexpr int PF = 4;
    for (int t = 0; t < tiles; ++t) {
      u32x4 ring[PF];
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = lp[i * 64];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(hb + 8 * j);
        const u32x4 bm = *reinterpret_cast<const u32x4*>(hb + 32 * LDH + 8 * j);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(hb + 64 * LDH + 8 * j);
        u32x4 al;
        if (MODE == 1) { al = ring[j % PF]; if (j + PF < 16) ring[j % PF] = lp[(j + PF) * 64]; } else al = wl[j];
        acc = mfma_bf16(al, bh, acc);
        acc = mfma_bf16(wh[j], bl, acc);
        acc = mfma_bf16(wm[j], bm, acc);
        acc = mfma_bf16(wm[j], bh, acc);
        acc = mfma_bf16(wh[j], bm, acc);
        acc = mfma_bf16(wh[j], bh, acc);
      }
      for (int i = 0; i < 16; ++i) { s += acc[i]; acc[i] = 0.f; }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  out[blockIdx.x * 512 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
int run(const char* name, float* dO, const u32x4* dW, int tiles, int mfmas, int cycles_each) {
  const size_t lds = (size_t)3 * 32 * LDH * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    fc2_loop<MODE><<<256, 512, lds>>>(dO, dW, tiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double per_tile_us = best * 1e3 / tiles;
  printf("%-52s %.3f ms: %.2f us per tile (two waves per SIMD: pipe time of %d MFMAs x 2 = %.2f us at 2.4 GHz)\n", name, best, per_tile_us,
         mfmas, 2.0 * mfmas * cycles_each / 2400.0);
  return 0;
}

int main() {
  float* dO; CK(hipMalloc(&dO, 256 * 512 * 4));
  u32x4* dW; CK(hipMalloc(&dW, 8 * 16 * 64 * 16)); CK(hipMemset(dW, 0x3c, 8 * 16 * 64 * 16));
  const int tiles = 2000;
  if (run<0>("fp32: 128 MFMAs / wave / tile, weights resident", dO, dW, tiles, 128, 64)) return 1;
  if (run<1>("bf16 x 3: 96 MFMAs, hi + mid resident, lo from L2", dO, dW, tiles, 96, 32)) return 1;
  if (run<2>("bf16 x 3: 96 MFMAs, all three terms resident", dO, dW, tiles, 96, 32)) return 1;
  return 0;
}
