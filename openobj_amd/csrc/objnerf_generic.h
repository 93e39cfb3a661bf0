// Internal interface of the layer-wise (any hidden width) training path, objnerf_generic.hip.
#pragma once
#include "../../include/objnerf_hip.h"
namespace objgen {
size_t train_workspace_bytes(const objnerf_net* net, int K, int R, int S, int feat);
int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream);
size_t eval_workspace_bytes(const objnerf_net* net, int K, long N);
int eval_points(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* scale,
                const float* pts, float* out_alpha, float* out_color, float* out_hfeat, float* out_clip,
                void* workspace, size_t workspace_bytes, void* stream);
}
