// placeholder until the backward kernels land
#include "common.hpp"
extern "C" int gnet_backward(const gnet_config* cfg, const gnet_shape* shape, const gnet_inputs* in,
                             const float* params, gnet_buffers* buf, float* grads, gnet_stream_t stream) {
  (void)cfg; (void)shape; (void)in; (void)params; (void)buf; (void)grads; (void)stream;
  return GNET_ERR_UNSUPPORTED;
}
