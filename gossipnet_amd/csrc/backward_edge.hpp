// Host entry points of the backward edge stage (backward_edge.hip), called from gnet_backward.
#pragma once
#include "common.hpp"

// winner maps + lists of every block (depends on the forward pass only); also the row list of the pw-MLP backward
int edge_stage_clear(const gnet_config* cfg, const gnet_shape* shape, gnet_buffers* buf, hipStream_t s);
// (part 0 = everything, 1 = what the edge kernels read, 2 = the reversed pairs' list positions, read by gather_winners only)
int edge_stage_prepare(const gnet_config* cfg, const gnet_shape* shape, const ParamLayout& L, const float* params,
                       gnet_buffers* buf, hipStream_t s, int part);
// edge_bwd_w of block b: d_pw (+=), compact g1 rows, partial d W(pw_fc1 rows 0-31), d W(pw_fc2), d b(pw_fc2)
int edge_stage_block(const gnet_config* cfg, const gnet_shape* shape, const ParamLayout& L, const float* params, int b,
                     gnet_buffers* buf, int n_partials, hipStream_t s);
int edge_stage_set_attributes();
