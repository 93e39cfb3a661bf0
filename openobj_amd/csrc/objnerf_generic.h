// Internal interface of the layer-wise (any hidden width) training path, objnerf_generic.hip.
#pragma once
#include "../../include/objnerf_hip.h"
namespace objgen {
size_t train_workspace_bytes(const objnerf_net* net, int K, int R, int S, int feat);
int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream);
}
