// Internal interface of the layer-wise (any hidden width) training path, objnerf_generic.hip.
#pragma once
#include "../../include/objnerf_hip.h"
// loss with the feature term hoisted past the compositing (objnerf_misc.hip); hz == NULL: the public
// objnerf_step_batch_loss
namespace objmisc {
struct LossHoisted {
  int Hh;
  const float* hf; const float* rayin; const float* gram;
  float* d_hf; float* rayfeat;
};
// loss_part: NULL or K * R * 4 floats of scratch -> bit-reproducible per-object loss terms (no float atomics)
int step_batch_loss_impl(const objnerf_loss_args* a, const LossHoisted* hz, void* stream, float* loss_part);
// objnerf_adamw_step_flags with the entries [ng_lo, ng_hi) skipped like has_grad == 0 (no byte mask needed)
int adamw_flags_range(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads, float* exp_avg,
                      float* exp_avg_sq, const uint8_t* has_grad, const int32_t* flags, int32_t* group_steps, int32_t bank,
                      int64_t colour_lo, int64_t feature_lo, int64_t feature_hi, int64_t ng_lo, int64_t ng_hi, float lr,
                      float beta1, float beta2, float eps, float weight_decay, void* stream);
// counts [K][2] alone (one workgroup per object; no flags, no zero fill)
int label_counts_only(int32_t K, int32_t R, const uint8_t* labels, int32_t* counts, void* stream);
}  // namespace objmisc

namespace objgen {
size_t train_workspace_bytes(const objnerf_net* net, int K, int R, int S, int feat, int sixteen = 0);
// done (optional): bit 0 = the call derived counts / flags itself (OBJNERF_TRAIN_SELF_COUNTS), bit 1 = the call applied
// a->optim; what is not reported is left to the caller (objnerf_train_step)
int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream, int* done = nullptr);
// the batched fp32 MFMA GEMM of this path, for the feature-head kernels of objnerf_train.hip:
//   C[z][m][n] (+)= sum_k A(z; m,k) B(z; k,n), element strides (sam, sak), (sbk, sbn), (scm, scn), batch strides bs*
void feat_gram(void* stream, int K, const float* params, long p_stride, int off_w, int off_b, int C, int Hh, float* gram,
               long gstride);
void feature_head(void* stream, int K, long n, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                  const float* hfeat, const float* weight, float* out);
void gemm_f32(void* stream, int batch, int M, int N, int Kd, const float* A, long sam, long sak, long bsa, const float* B,
              long sbk, long sbn, long bsb, float* C, long scm, long scn, long bsc, bool accumulate);
//   long contraction over n, few output tiles: split-K with float atomics into a PRE-ZEROED C (scn = 1)
// split-K weight gradient over n samples (C pre-zeroed only for the atomics fallback).  parts: scratch of
// wgrad_parts_floats(batch, M, N, n) floats for the deterministic slab-and-reduce form; NULL = float atomics
size_t wgrad_parts_floats(int batch, int M, int N, long n);
void wgrad_f32(void* stream, int batch, int M, int N, long n, const float* A, long sam, long sak, long bsa, const float* B,
               long sbk, long sbn, long bsb, float* C, long scm, long bsc, float* parts, size_t parts_floats);
// The hoisted 512-d feature head around a fused kernel of ANOTHER width (the hidden-256 path, objnerf_train256.hip):
// buffers, the preparation ahead of the kernel (G = W_of^T W_of, wb, bb per object; u = W_of^T g, beta, |g| per ray) and
// the head's gradient behind it (the rays' moment GEMMs + featg_finish_kernel).  operands: 0 fp32, 1 bf16, 2 fp16 GEMMs.
struct FeatHead {
  float *gram, *rayin, *rayfeat, *X1, *X2, *Tm, *mom, *parts; size_t parts_floats;
};
size_t feat_head_workspace_bytes(int K, int R, int Hh, int C);
FeatHead feat_head_carve(char* base, int K, int R, int Hh, int C);
int feat_head_prep(void* stream, int K, int R, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                   const float* gt_feat, const FeatHead& f, int operands);
int feat_head_grads(void* stream, int K, int R, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                    const float* gt_feat, const FeatHead& f, float* grads, int operands);
size_t eval_workspace_bytes(const objnerf_net* net, int K, long N);
// pts == NULL: emb_in [K][N][129] is the embedding (OccupancyMap.forward on a caller-supplied tensor)
int eval_points(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* scale,
                const float* pts, float* out_alpha, float* out_color, float* out_hfeat, float* out_clip,
                void* workspace, size_t workspace_bytes, void* stream, const float* emb_in = nullptr);
// backward halves of the mirrored modules (objnerf_mlp_backward_ws / objnerf_embed_bwd)
size_t mlp_backward_workspace_bytes(const objnerf_net* net, int K, long N, int with_clip);
int mlp_backward(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* emb,
                 const float* d_alpha, const float* d_color, const float* d_clip, float* grads, float* d_emb,
                 void* workspace, size_t workspace_bytes, void* stream);
int embed_backward(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* scale,
                   const float* pts, const float* d_emb, float* d_B, float* scratch, void* stream);
}

// hidden 256 in the 16-bit operand modes without the feature loss (BASELINE configs[4]): objnerf_train256.hip
namespace obj256 {
bool applicable(const objnerf_net* net, const objnerf_train_args* a);
size_t workspace_bytes(int K, int R, int S, int feat = 0, int C = 0);
int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream);
}
