// train_fused_bf16v2f_kernel: with the 512-d feature-distillation loss (BASELINE configs[2] / [3]): see objnerf_bf16v2_body.h
#define V2_FEAT 1
#include "objnerf_bf16v2_body.h"
