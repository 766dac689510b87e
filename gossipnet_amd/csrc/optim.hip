// Training step around the hot path: the update that train.py:64-77 builds with
// slim.learning.create_train_op -- optional per-tensor gradient-norm clipping (clip_by_norm of every
// gradient, cfg.train.gradient_clipping), then tf.train.AdamOptimizer (defaults beta1 .9, beta2 .999,
// eps 1e-8, train.py:66-67) or tf.train.MomentumOptimizer (train.py:68-70) -- as fused elementwise HIP
// kernels over the flat parameter / gradient buffers that the all-reduce already uses.
// TensorFlow's Adam (un-vendored; restated from its published update rule):
//   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   var -= lr_t * m / (sqrt(v) + eps)
// HBM-bound: 16 B read + 12 B written per parameter.
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps, float gscale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

__global__ void __launch_bounds__(256) momentum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ acc, long long n, float lr,
                                                       float momentum, float gscale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = momentum * acc[i] + g[i] * gscale;   // accum = momentum * accum + grad
  acc[i] = a;
  p[i] = p[i] - lr * a;                                // var -= lr * accum
}

// tf.clip_by_norm per gradient tensor: t * clip / max(||t||_2, clip).  One workgroup per tensor.
__global__ void __launch_bounds__(256) clip_kernel(float* __restrict__ g, const long long* __restrict__ offs, float clip) {
  __shared__ float red[256];
  const long long b = offs[blockIdx.x], e = offs[blockIdx.x + 1];
  float s = 0.f;
  for (long long i = b + threadIdx.x; i < e; i += 256) s = fmaf(g[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float norm = sqrtf(red[0]);
  if (norm > clip) {
    const float sc = clip / norm;
    for (long long i = b + threadIdx.x; i < e; i += 256) g[i] *= sc;
  }
}

}  // namespace

extern "C" int gnet_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr,
                              float beta1, float beta2, float eps, int64_t t, float grad_scale, gnet_stream_t stream) {
  clear_hip_error();
  if (n < 0 || t < 1) return GNET_ERR_INVALID;
  if (n == 0) return GNET_OK;
  if (!params || !grads || !m || !v) return GNET_ERR_INVALID;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  adam_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, grads, m, v, n, (float)lr_t, beta1, beta2,
                                                                      eps, grad_scale);
  return launch_status();
}

extern "C" int gnet_momentum_step(float* params, const float* grads, float* accum, int64_t n, float lr, float momentum,
                                  float grad_scale, gnet_stream_t stream) {
  clear_hip_error();
  if (n < 0) return GNET_ERR_INVALID;
  if (n == 0) return GNET_OK;
  if (!params || !grads || !accum) return GNET_ERR_INVALID;
  momentum_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, grads, accum, n, lr, momentum, grad_scale);
  return launch_status();
}

extern "C" int gnet_clip_by_norm(float* grads, const int64_t* tensor_offsets, int32_t n_tensors, float clip_norm,
                                 gnet_stream_t stream) {
  clear_hip_error();
  if (n_tensors < 0 || !(clip_norm > 0.f)) return GNET_ERR_INVALID;
  if (n_tensors == 0) return GNET_OK;
  if (!grads || !tensor_offsets) return GNET_ERR_INVALID;
  clip_kernel<<<n_tensors, 256, 0, (hipStream_t)stream>>>(grads, (const long long*)tensor_offsets, clip_norm);
  return launch_status();
}
