// Micro-benchmark: do fp32 MFMAs and VALU instructions overlap on a CDNA4 SIMD?
//   mode 1: MFMA only      mode 2: VALU only      mode 3: both in one wave's loop
//   mode 4: 8-wave workgroups, waves 0-3 MFMA only, waves 4-7 VALU only (two waves per SIMD)
//   mode 5: 8-wave workgroups, every wave alternates a pure MFMA phase and a pure VALU phase
//   mode 6: as 5, with a bare s_barrier behind every phase and waves 4-7 started ONE PHASE LATE: the two waves of a SIMD are
//           held in anti-phase (one in its MFMA phase while the other is in its vector phase); mode 7: the same with the
//           MFMA phase at wave priority 3; mode 8: barriers but NO offset (both waves of a SIMD in the same phase: the control)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o gpurun_out/overlap
//        (-DBF16PIPE: the same with v_mfma_f32_32x32x16_bf16, 32 cycles, in place of the fp32 MFMA's 64)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef BF16PIPE
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, qb), acc_, 0, 0, 0)
#define MFMA_CYCLES 32.0
#else
#define MFMA(acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc_, 0, 0, 0)
#define MFMA_CYCLES 64.0
#endif

#define VALU8(x) \
  asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b))

template <int NM, int NV>
__global__ void __launch_bounds__(512) bench(float* out, int iters, int mode, long long* ticks) {
  const long long c0 = clock64();
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  const float a = 0.999f, b = 0.001f;
#ifdef BF16PIPE
  const u32x4 qa = {0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u + threadIdx.x}, qb = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#endif
  const int wave = threadIdx.x >> 6;
  const bool split = mode == 4;
  const bool do_m = mode == 1 || mode == 3 || (split && wave < 4);
  const bool do_v = mode == 2 || mode == 3 || (split && wave >= 4);
  if (mode >= 6) {
    const bool late = mode != 8 && wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
      if (mode == 7) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int j = 0; j < NM; ++j) { MFMA(acc0); }
      if (mode == 7) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < NV; ++j) { VALU8(x); }
      __builtin_amdgcn_s_barrier();
    }
    if (mode != 8 && wave < 4) __builtin_amdgcn_s_barrier();
  } else if (mode == 5) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NM; ++j) { MFMA(acc0); }
#pragma unroll
      for (int j = 0; j < NV; ++j) { VALU8(x); }
    }
  } else if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        MFMA(acc0);
#pragma unroll
        for (int q = 0; q < NV / NM; ++q) { VALU8(x); }
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NM; ++j) MFMA(acc0);
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NV; ++j) { VALU8(x); }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (ticks && blockIdx.x == 0 && threadIdx.x == 0) *ticks = clock64() - c0;
}

template <int NM, int NV>
void run(const char* name, int mode, int threads, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  static long long* ticks = nullptr; if (!ticks) hipMalloc(&ticks, 8);
  bench<NM, NV><<<256, threads>>>(out, 10, mode, nullptr);
  hipEventRecord(e0);
  bench<NM, NV><<<256, threads>>>(out, iters, mode, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long th = 0; hipMemcpy(&th, ticks, 8, hipMemcpyDeviceToHost);
  // per wave per iteration: NM MFMAs (64 cycles each), NV*8 VALU (4 cycles each)
  printf("%-46s NM=%2d NV8=%2d threads=%d  %.3f ms  -> %.0f ns/iter, %.0f s_memtime ticks/iter (MFMA alone %.0f cyc)\n", name, NM, NV, threads, ms,
         ms * 1e6 / iters, (double)th / iters, NM * MFMA_CYCLES);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  run<16, 16>("1 MFMA only, 1 wave/SIMD", 1, 256, out);
  run<16, 16>("2 VALU only, 1 wave/SIMD", 2, 256, out);
  run<16, 16>("3 interleaved in one wave, 1 wave/SIMD", 3, 256, out);
  run<16, 16>("5 phases MFMA then VALU, 1 wave/SIMD", 5, 256, out);
  run<16, 16>("1 MFMA only, 2 waves/SIMD", 1, 512, out);
  run<16, 16>("2 VALU only, 2 waves/SIMD", 2, 512, out);
  run<16, 16>("3 interleaved, 2 waves/SIMD", 3, 512, out);
  run<16, 16>("4 split: waves 0-3 MFMA, 4-7 VALU", 4, 512, out);
  run<16, 16>("5 phases MFMA then VALU, 2 waves/SIMD", 5, 512, out);
  run<16, 16>("6 anti-phase by barriers, 2 waves/SIMD", 6, 512, out);
  run<16, 16>("7 anti-phase + MFMA phase at priority 3", 7, 512, out);
  run<16, 16>("8 barriers, same phase (control)", 8, 512, out);
  run<16, 32>("3 interleaved, 2 waves/SIMD", 3, 512, out);
  run<16, 32>("4 split: waves 0-3 MFMA, 4-7 VALU", 4, 512, out);
  run<16, 32>("5 phases MFMA then VALU, 2 waves/SIMD", 5, 512, out);
  run<16, 32>("6 anti-phase by barriers, 2 waves/SIMD", 6, 512, out);
  run<16, 32>("7 anti-phase + MFMA phase at priority 3", 7, 512, out);
  run<16, 32>("8 barriers, same phase (control)", 8, 512, out);
  run<16, 8>("5 phases MFMA then VALU, 2 waves/SIMD", 5, 512, out);
  run<16, 8>("6 anti-phase by barriers, 2 waves/SIMD", 6, 512, out);
  return 0;
}
