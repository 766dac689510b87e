// Micro-benchmark: how does the fp32 MFMA pipe of a CDNA4 SIMD treat accumulator CHAINS?
//   one accumulator (every MFMA depends on the previous one), two / four / eight accumulators round-robin, one long chain with
//   a link of a second chain every fourth MFMA, and a chain with an LDS read + wait between groups of four links -- at one and
//   two waves per SIMD.  Cycles (s_memtime) per MFMA of one wave; 64 = the pipe's issue rate for v_mfma_f32_32x32x2_f32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip -o tools/mfma_chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(512) bench(float* out, int iters, long long* ticks) {
  __shared__ float sbuf[512 * 4];
  f32x16 c[8];
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) c[k][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1e-3f;
  sbuf[threadIdx.x] = a; sbuf[512 + threadIdx.x] = b;
  __syncthreads();
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 32; ++j) MF(c[0], a, b);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { MF(c[0], a, b); MF(c[1], a, b); }
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { MF(c[0], a, b); MF(c[1], a, b); MF(c[2], a, b); MF(c[3], a, b); }
    } else if (MODE == 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { MF(c[0], a, b); MF(c[1], a, b); MF(c[2], a, b); MF(c[3], a, b); MF(c[4], a, b); MF(c[5], a, b); MF(c[6], a, b); MF(c[7], a, b); }
    } else if (MODE == 5) {       // a chain with a link of a second chain every fourth MFMA (32 MFMAs in all: 26 + 6)
#pragma unroll
      for (int j = 0; j < 32; ++j) { if (j % 5 == 4) MF(c[1], a, b); else MF(c[0], a, b); }
    } else if (MODE == 6) {       // a chain whose A operand comes from LDS, read + waited per group of four links
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sbuf + 4 * ((threadIdx.x + 4 * j + it) & 127));
        MF(c[0], v.x, b); MF(c[0], v.y, b); MF(c[0], v.z, b); MF(c[0], v.w, b);
      }
    } else if (MODE == 7) {       // the same with the NEXT group's operand requested before this group's MFMAs
      float4 v = *reinterpret_cast<const float4*>(sbuf + 4 * ((threadIdx.x + it) & 127));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 vn = *reinterpret_cast<const float4*>(sbuf + 4 * ((threadIdx.x + 4 * (j + 1) + it) & 127));
        __builtin_amdgcn_sched_barrier(0);
        MF(c[0], v.x, b); MF(c[0], v.y, b); MF(c[0], v.z, b); MF(c[0], v.w, b);
        __builtin_amdgcn_sched_barrier(0);
        v = vn;
      }
    } else if (MODE == 9) {       // two chains alternating, operands from LDS per group (acc0 / acc1 as in mma_abt2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sbuf + 4 * ((threadIdx.x + 4 * j + it) & 127));
        const float4 w = *reinterpret_cast<const float4*>(sbuf + 4 * ((threadIdx.x + 4 * j + it + 64) & 127));
        MF(c[0], v.x, b); MF(c[1], w.x, b); MF(c[0], v.y, b); MF(c[1], w.y, b); MF(c[0], v.z, b); MF(c[1], w.z, b); MF(c[0], v.w, b); MF(c[1], w.w, b);
      }
    }
  }
  const long long c1 = clock64();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += c[k][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = c1 - c0;
}

template <int MODE>
void run(const char* name, int threads, float* out, long long* ticks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE><<<256, threads>>>(out, 10, ticks);
  hipEventRecord(e0);
  bench<MODE><<<256, threads>>>(out, iters, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long th = 0; hipMemcpy(&th, ticks, 8, hipMemcpyDeviceToHost);
  const double tflops = 256.0 * (threads / 64) * iters * 32.0 * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-64s %d waves/SIMD: %6.1f ticks per MFMA of one wave (pipe: %5.1f per MFMA); %.3f ms = %.1f TFLOP/s; ticks at %.2f GHz\n", name, threads / 256,
         (double)th / iters / 32.0, (double)th / iters / 32.0 / (threads / 256), ms, tflops, (double)th / (ms * 1e-3) / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  long long* ticks; hipMalloc(&ticks, 8);
  for (int threads = 256; threads <= 512; threads += 256) {
    run<1>("one accumulator chain", threads, out, ticks);
    run<2>("two accumulators alternating", threads, out, ticks);
    run<4>("four accumulators round-robin", threads, out, ticks);
    run<8>("eight accumulators round-robin", threads, out, ticks);
    run<5>("a chain with a link of a second chain every fifth MFMA", threads, out, ticks);
    run<6>("a chain, A operand from LDS read + waited per four links", threads, out, ticks);
    run<7>("a chain, the next group's LDS operand requested a group ahead", threads, out, ticks);
    run<9>("two chains alternating, LDS operands per group", threads, out, ticks);
  }
  return 0;
}
