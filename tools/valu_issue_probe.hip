// Micro-benchmark: what does ONE wave pay per vector instruction on a CDNA4 SIMD -- dependent chains against independent ones, the
// compare + add-with-carry pair of edge_fwd_w's hit bits, scalar instructions between vector ones -- at 1, 2 and 3 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o tools/valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int MODE>
__global__ void __launch_bounds__(768) bench(float* out, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  const float a = 1.0001f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 64 dependent v_mul_f32 (one chain)
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %1\n")) : "+v"(x0) : "v"(a));
    } else if (MODE == 1) {   // 64 v_mul_f32 in two chains
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n")) : "+v"(x0), "+v"(x1) : "v"(a));
    } else if (MODE == 2) {   // four chains
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
    } else if (MODE == 3) {   // eight chains
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"))
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 4) {   // hit bits: compare + add-with-carry, ONE accumulator (the carry chain is serial)
      asm volatile(REP16(REP4("v_cmp_eq_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n")) : "+v"(b0) : "v"(x1), "v"(x2) : "vcc");
    } else if (MODE == 5) {   // hit bits, two accumulators alternating (edge_fwd_w's form)
      asm volatile(REP16(REP4("v_cmp_eq_f32 vcc, %2, %3\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_eq_f32 vcc, %3, %2\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n"))
                   : "+v"(b0), "+v"(b1) : "v"(x1), "v"(x2) : "vcc");
    } else if (MODE == 6) {   // hit bits, four accumulators, the compares into four SGPR pairs first
      asm volatile(REP16(REP4("v_cmp_eq_f32 s[40:41], %4, %5\n v_cmp_eq_f32 s[42:43], %5, %4\n v_cmp_eq_f32 s[44:45], %4, %4\n v_cmp_eq_f32 s[46:47], %5, %5\n"
                               "v_addc_co_u32 %0, s[40:41], %0, %0, s[40:41]\n v_addc_co_u32 %1, s[42:43], %1, %1, s[42:43]\n"
                               "v_addc_co_u32 %2, s[44:45], %2, %2, s[44:45]\n v_addc_co_u32 %3, s[46:47], %3, %3, s[46:47]\n"))
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x1), "v"(x2) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (MODE == 7) {   // eight independent chains with an s_nop 0 behind every vector instruction
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %8\n s_nop 0\n v_mul_f32 %1, %1, %8\n s_nop 0\n v_mul_f32 %2, %2, %8\n s_nop 0\n v_mul_f32 %3, %3, %8\n s_nop 0\n"
                               "v_mul_f32 %4, %4, %8\n s_nop 0\n v_mul_f32 %5, %5, %8\n s_nop 0\n v_mul_f32 %6, %6, %8\n s_nop 0\n v_mul_f32 %7, %7, %8\n s_nop 0\n"))
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 8) {   // eight independent chains with a scalar add behind every vector instruction
      asm volatile(REP16(REP4("v_mul_f32 %0, %0, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 s40, s40, 1\n"
                               "v_mul_f32 %4, %4, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %5, %5, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %7, %7, %8\n s_add_u32 s40, s40, 1\n"))
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "s40", "scc");
    } else if (MODE == 9) {   // v_pk_add_f32, eight independent
      asm volatile(REP16(REP4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"))
                   : "+v"(*(double*)&x0), "+v"(*(double*)&x2), "+v"(*(double*)&x4), "+v"(*(double*)&x6) : "v"(*(double*)&x0));
    } else if (MODE == 10) {  // v_perm_b32, four independent
      asm volatile(REP16(REP4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")) : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x1), "v"(x2));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + b0 + b1 + b2 + b3;
}

template <int MODE>
void run(const char* name, int per_iter, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int threads = 256; threads <= 768; threads += 256) {
    bench<MODE><<<256, threads>>>(out, 10);
    (void)hipEventRecord(e0);
    bench<MODE><<<256, threads>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %d wave(s)/SIMD: %.2f ns per vector instruction and wave, %.2f per SIMD\n", name, threads / 256, ms * 1e6 / iters / per_iter,
           ms * 1e6 / iters / per_iter / (threads / 256));
    fflush(stdout);
  }
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 768 * sizeof(float));
  run<0>("v_mul_f32, one dependent chain", 64, out);
  run<1>("v_mul_f32, two chains", 128, out);
  run<2>("v_mul_f32, four chains", 256, out);
  run<3>("v_mul_f32, eight chains", 512, out);
  run<4>("v_cmp + v_addc, one accumulator", 128, out);
  run<5>("v_cmp + v_addc, two accumulators (edge_fwd_w)", 256, out);
  run<6>("4 x v_cmp -> SGPR pairs, 4 x v_addc, four accumulators", 512, out);
  run<7>("eight chains, s_nop 0 behind every instruction", 512, out);
  run<8>("eight chains, s_add_u32 behind every instruction", 512, out);
  run<9>("v_pk_add_f32, four independent", 256, out);
  run<10>("v_perm_b32, four independent", 256, out);
  return 0;
}
