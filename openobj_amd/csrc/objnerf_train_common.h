// Declarations shared by the fp32 and bf16 fused training kernels.
#pragma once
#include "objnerf_mlp.h"

namespace objtrain {
using namespace obj32;

constexpr int TS = 128;            // samples per workgroup tile
constexpr int NWAVE = 8;
constexpr int NTHR = 64 * NWAVE;
constexpr int STG_LD = 130;        // staging row stride (floats): = 2 (mod 32) -> wgrad operand reads hit 32 banks
constexpr int STG_ROWS = 192;
constexpr int SM_FLOATS = 4 * TS;  // s_alpha | s_col[3]; overwritten in place by d raw-alpha | d raw-colour
constexpr int NRED = 6 * 32 + 4 + 4;

struct TrainDev {
  int K, R, S, G, TR, NT;
  float color_scaling, opacity_scaling, feat_scaling, obj_center;
  const float* params; long p_stride; const float* scale;
  const float* pts; const float* origins; const float* dirs; const float* z;
  const float* gt_depth; const float* gt_rgb; const uint8_t* labels; const float* gt_feat;
  const int* counts; const int* flags;
  float* slab;        // [K][G][slab_stride]
  long slab_stride;
  float* loss_part;   // [K][G][4]
  // feature-distillation branch (gt_feat != NULL)
  const float* rayin;   // [K][R][RAYIN]  u = W_of^T g (32), beta = b_of . g, |g|     (feat_pre_kernel)
  const float* gram;    // [K][GRAM]      G = W_of^T W_of (32x32), wb = W_of^T b_of (32), b_of . b_of
  float* rayfeat;       // [K][R][RAYFEAT] composited hidden fh (32), a, c, opacity   (-> feat_post kernels)
  Layout L;
};
constexpr int RAYIN = 34, GRAM = 1088, RAYFEAT = 36;
// LDS aliases inside the staging area, valid from the forward pass until phase B of the backward pass
constexpr int HF_LD = 33;                       // hfbuf [128][33] at stg + 0
constexpr int OFF_GBUF = TS * HF_LD;            // G [32][33], wb [32], bb          (4224 ..)
constexpr int OFF_FHB = OFF_GBUF + 32 * 33 + 64;   // fh exchange buffer [NWAVE][2][32] (one per wave)
constexpr int OFF_SW = 80 * STG_LD;             // rows 80..95 are untouched by the phase-A staging
constexpr int OFF_GFH = OFF_SW + TS;            // gfh [16][32]
constexpr int OFF_GOF = OFF_GFH + 16 * 32;      // gO_feat [16], O [16]
static_assert(OFF_FHB + 64 * NWAVE <= 80 * STG_LD && OFF_GOF + 32 <= 96 * STG_LD, "feat lds aliases");


// bf16 MFMA variant (objnerf_train_bf16.hip): same tile structure, bf16 operands, fp32 accumulation
size_t bf16_lds_bytes();
void launch_train_bf16(const TrainDev& d, void* stream, bool feat);

}  // namespace objtrain
