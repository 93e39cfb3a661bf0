// train_fused_bf16v2_kernel without the feature loss (BASELINE configs[1]): see objnerf_bf16v2_body.h
#define V2_FEAT 0
#include "objnerf_bf16v2_body.h"
