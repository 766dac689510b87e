// Micro-benchmark behind DESIGN.md's decision NOT to fuse a block's phases into one persistent launch:
// what does a grid-wide barrier cost on this chip against a dependent kernel boundary?
//   A: P dependent launches of a trivial kernel (each reads the value the previous one wrote): time per boundary
//   B: ONE launch of the same grid running P phases separated by an XCD-hierarchical grid barrier
//      (per-XCD arrival counter -> the XCD's last arriver arrives at the top counter -> the last of those bumps the
//       generation word every workgroup polls; agent-scope release before arriving, acquire after leaving)
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o gpurun_out/gbar && gpurun_out/gbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) phase_kernel(float* buf, int phase) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  buf[i] = buf[i] * 0.5f + (float)phase;              // a dependent touch of one cache line per wave-quarter
}

struct Bar { unsigned xcd[8 * 32]; unsigned top[32]; unsigned gen[32]; };   // one 128-byte line per word

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned wgs_per_xcd, unsigned n_xcd, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7;
    ++epoch;
    __atomic_thread_fence(__ATOMIC_RELEASE);                                 // (agent scope on AMDGPU)
    const unsigned a = __hip_atomic_fetch_add(&b->xcd[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == epoch * wgs_per_xcd) {                                      // last arriver of this XCD
      const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 == epoch * n_xcd) __hip_atomic_store(&b->gen[0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long spins = 0;
    while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1ll << 26)) break;                                      // bounded: never hang the box
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) persistent_kernel(float* buf, int phases, Bar* bar, unsigned wgs_per_xcd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned epoch = 0;
  for (int p = 0; p < phases; ++p) {
    buf[i] = buf[i] * 0.5f + (float)p;
    grid_barrier(bar, wgs_per_xcd, 8, epoch);
  }
}

int main() {
  const int phases = 64, reps = 20;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 512, 768}) {
    float* buf; hipMalloc(&buf, (size_t)grid * 256 * 4); hipMemsetAsync(buf, 0, (size_t)grid * 256 * 4, s);
    Bar* bar; hipMalloc(&bar, sizeof(Bar));
    float ms_a = 0.f, ms_b = 0.f;
    for (int w = 0; w < 3; ++w) for (int p = 0; p < phases; ++p) phase_kernel<<<grid, 256, 0, s>>>(buf, p);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) for (int p = 0; p < phases; ++p) phase_kernel<<<grid, 256, 0, s>>>(buf, p);
    hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    for (int w = 0; w < 3; ++w) { hipMemsetAsync(bar, 0, sizeof(Bar), s); persistent_kernel<<<grid, 256, 0, s>>>(buf, phases, bar, grid / 8); }
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) { hipMemsetAsync(bar, 0, sizeof(Bar), s); persistent_kernel<<<grid, 256, 0, s>>>(buf, phases, bar, grid / 8); }
    hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("error\n"); return 1; }
    printf("grid %4d workgroups x 256 threads: %d dependent launches %.2f us per phase | one persistent launch, XCD-hierarchical grid barrier %.2f us per phase\n",
           grid, phases, ms_a * 1e3 / (reps * phases), ms_b * 1e3 / (reps * phases));
    hipFree(buf); hipFree(bar);
  }
  return 0;
}
