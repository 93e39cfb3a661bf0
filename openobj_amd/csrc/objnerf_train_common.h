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
  int derive_flags;   // OBJNERF_TRAIN_SELF_COUNTS: `flags` is not an input -- every workgroup derives the early-return pair
                      // from counts [K][2] (batch_flags below) and finalize_kernel publishes it
  float* slab;        // [K][G][slab_stride]  (flat mode: [K][Gs][slab_stride])
  long slab_stride;
  float* loss_part;   // [K][G][4]
  // Work distribution.  flat_nwg == 0: workgroup (k, gi) sweeps tiles gi, gi + G, ... of object k.  flat_nwg > 0 (the
  // second-generation fp32 kernel): workgroup b of flat_nwg takes the b-th share of the flat (object, tile) space --
  // one to a few segments (object, [t0, t1)) -- so that every CU carries the same number of tiles whatever K is
  // (K = 50 on 256 CUs: 400 tiles each instead of 410 on 250 of them); an object has at most Gs partial slabs.
  int flat_nwg, Gs;
  // feature-distillation branch (gt_feat != NULL)
  const float* rayin;   // [K][R][RAYIN]  u = W_of^T g (32), beta = b_of . g, |g|     (feat_pre_kernel)
  const float* gram;    // [K][GRAM]      G = W_of^T W_of (32x32), wb = W_of^T b_of (32), b_of . b_of
  float* rayfeat;       // [K][R][RAYFEAT] composited hidden fh (32), a, c, opacity   (-> feat_post kernels)
  uint8_t* relu_masks;  // objnerf_train_args.relu_masks (test hook) or NULL
  float* emb_debug;     // objnerf_train_args.emb_debug (test hook) or NULL
  Layout L;
};
constexpr int RAYIN = 34, GRAM = 1088, RAYFEAT = 36;
// The early-return pair of render_rays.py:89-94 ("some object of the batch has an empty mask"): the caller's (possibly
// reduced over GPUs), or derived from the K objects' counts by the calling wave (all 64 lanes active).
__device__ __forceinline__ void batch_flags(const TrainDev& a, int& f0, int& f1) {
  if (!a.derive_flags) { f0 = a.flags[0]; f1 = a.flags[1]; return; }
  int e0 = 0, e1 = 0;
  for (int k = threadIdx.x & 63; k < a.K; k += 64) { e0 |= a.counts[2 * k] == 0; e1 |= a.counts[2 * k + 1] == 0; }
  f0 = __ballot(e0) != 0; f1 = __ballot(e1) != 0;
}
// flat mode: the workgroup whose share [T b / nwg, T (b + 1) / nwg) holds flat tile x
__host__ __device__ inline int flat_wg_of(const long T, const int nwg, const long x) { return (int)(((x + 1) * nwg - 1) / T); }
// LDS aliases inside the staging area, valid from the forward pass until phase B of the backward pass
constexpr int HF_LD = 33;                       // hfbuf [128][33] at stg + 0
constexpr int OFF_GBUF = TS * HF_LD;            // G [32][33], wb [32], bb          (4224 ..)
constexpr int OFF_FHB = OFF_GBUF + 32 * 33 + 64;   // fh exchange buffer [NWAVE][2][32] (one per wave)
constexpr int OFF_SW = 80 * STG_LD;             // rows 80..95 are untouched by the phase-A staging
constexpr int OFF_GFH = OFF_SW + TS;            // gfh [16][32]
constexpr int OFF_GOF = OFF_GFH + 16 * 32;      // gO_feat [16], O [16]
static_assert(OFF_FHB + 64 * NWAVE <= 80 * STG_LD && OFF_GOF + 32 <= 96 * STG_LD, "feat lds aliases");


// ReLU branch bits of one 16-sample block (test hook, objnerf_train_args.relu_masks): lane (c, g) holds features
// 4 g + r (tile 0) and 16 + 4 g + r (tile 1) of its sample; the lane groups g, g ^ 1 share bytes g >> 1 and 2 + (g >> 1).
__device__ __forceinline__ void write_relu_mask(uint8_t* dst /* masks of this lane's sample */, const int layer,
                                                const int g, const T32& act, const bool valid) {
  unsigned n0 = 0, n1 = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    n0 |= (act.t[0][r] > 0.0f ? 1u : 0u) << r;
    n1 |= (act.t[1][r] > 0.0f ? 1u : 0u) << r;
  }
  const unsigned v = (n0 << (4 * (g & 1))) | (n1 << (8 + 4 * (g & 1)));
  const unsigned o = v | (unsigned)__shfl_xor((int)v, 16, 64);
  if (valid && (g & 1) == 0) {
    dst[layer * 4 + (g >> 1)] = (uint8_t)(o & 0xff);
    dst[layer * 4 + 2 + (g >> 1)] = (uint8_t)(o >> 8);
  }
}

// second-generation fp32 kernel (objnerf_train32.hip): hidden 32, S <= 64, with or without the feature loss
size_t fused32_lds_bytes();
void launch_train32(const TrainDev& d, void* stream, bool feat);

// bf16 MFMA variant (objnerf_train_bf16.hip): same tile structure, bf16 operands, fp32 accumulation
size_t bf16_lds_bytes();
void launch_train_bf16(const TrainDev& d, void* stream, bool feat);
// second generation (objnerf_bf16v2_body.h): 64 samples per ray; without / with the feature loss
size_t bf16v2_lds_bytes();
void launch_train_bf16_v2(const TrainDev& d, void* stream);
void launch_train_bf16_v2f(const TrainDev& d, void* stream);

}  // namespace objtrain
