// Shared by the two generations of the bf16-operand fused training kernels (objnerf_train_bf16.hip,
// objnerf_train_bf16v2.hip): the forward weight images in LDS, their staging from the fp32 master weights, operand
// packing and the forward MFMA block.
//
// One v_mfma_f32_16x16x32_bf16 consumes a whole 32-feature block: lane (c, g) supplies k-slots 8g..8g+7 = features
// phi(g,e) = (e < 4 ? 4g + e : 16 + 4g + e - 4), i.e. exactly the 8 fp32 registers the lane holds of a D16-layout
// activation block, packed to bf16.  A forward image row is [out][block][g][e] (position 8 g + e of a block <-> feature
// phi(g, e)); row pitches are 32 B x odd (mod 256 B), which the 16-lane ds_read_b128 groups read conflict-free.
#pragma once
#define OBJ_PE_NO_LOW 1      // these kernels' angles have no low part (pe_project_b): the anchors skip its fma
#include "objnerf_mlp32.h"
#include "objnerf_train_common.h"

namespace objtrain {
namespace bf16k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// V2_ABL (ceiling builds of the second-generation bf16 kernels only, tools/bf16_ablation.sh; the Makefile's value is 0):
//   1 no compositing   2 no tile-loop barriers   8 no weight-gradient MFMAs   64 no MFMA at all   (+ -DOBJ32_ABL=16: no v_sin / v_cos)
#ifndef V2_ABL
#define V2_ABL 0
#endif
#if (V2_ABL) & 64
#define MFMA_BF16(a, b, c) (c)
#else
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

// forward images (byte offsets), 32 rows each
constexpr int RS_IN = 224, RS_M = 96, RS_CAT = 288, RS_CL = 224;
constexpr int B_IN = 0;
constexpr int B_M1 = B_IN + 32 * RS_IN;
constexpr int B_CAT = B_M1 + 32 * RS_M;
constexpr int B_M2 = B_CAT + 32 * RS_CAT;
constexpr int B_CL = B_M2 + 32 * RS_M;
constexpr int FWD_IMG_END = B_CL + 32 * RS_CL;
// fp32 small vectors (float offsets inside their block): bm1[32] bm2[32] wa[32] woc[96] hb[4] peb[72]
constexpr int S_BM1 = 0, S_BM2 = 32, S_WA = 64, S_WOC = 96, S_HB = 192, S_PEB = 196;
constexpr int SMALL_BYTES = 1280;

__device__ __forceinline__ int phi(int g, int e) { return e < 4 ? 4 * g + e : 16 + 4 * g + e - 4; }

// value of layer weight (out i, K-order input feature f).  Embedding features are in the direction-owner order of
// objnerf_mlp32.h: feature kappa <-> entry (t, g) <-> reference column x1_col / x2_col, the layer's bias on the
// constant-1 entry, zero on the padding entries.
__device__ __forceinline__ float w_emb(const float* P, const int w_off, const int b_off, const int ncols, const int hid,
                                       const bool x2, const int i, const int kappa) {
  int t, g;
  obj32n::kappa_tg(kappa, t, g);
  const int col = x2 ? obj32n::x2_col(t, g) : obj32n::x1_col(t, g);
  if (col >= 0) return P[w_off + i * ncols + hid + col];
  return col == obj32n::BIAS_COL ? P[b_off + i] : 0.f;
}
__device__ __forceinline__ float w_in(const float* P, const Layout& L, int i, int f) {
  return w_emb(P, L.in_w, L.in_b, OBJ_E1, 0, false, i, f);
}
__device__ __forceinline__ float w_cat(const float* P, const Layout& L, int i, int f) {
  if (f < H) return P[L.cat_w + i * (H + OBJ_E1) + f];
  return w_emb(P, L.cat_w, L.cat_b, H + OBJ_E1, H, false, i, f - H);
}
__device__ __forceinline__ float w_cl(const float* P, const Layout& L, int i, int f) {
  if (f < H) return P[L.cl_w + i * (H + OBJ_E2) + f];
  return w_emb(P, L.cl_w, L.cl_b, H + OBJ_E2, H, true, i, f - H);
}
__device__ __forceinline__ float w_fl(const float* P, const Layout& L, int i, int f) {
  if (f < H) return P[L.fl_w + i * (H + OBJ_E2) + f];
  return w_emb(P, L.fl_w, L.fl_b, H + OBJ_E2, H, true, i, f - H);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = (__bf16)lo[e];
    o[4 + e] = (__bf16)hi[e];
  }
  return o;
}
__device__ __forceinline__ bf16x8 pack32(const T32& x) { return pack8(x.t[0], x.t[1]); }

// acc (32 outs) += W[:, block] * x        img_lane = lds + B_X + c * RS + 16 g
template <int RS>
__device__ __forceinline__ void fwd_blk(T32& acc, const char* img_lane, const int blk, const bf16x8 xb) {
  const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(img_lane + 64 * blk);
  const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(img_lane + 16 * RS + 64 * blk);
  acc.t[0] = MFMA_BF16(a0, xb, acc.t[0]);
  acc.t[1] = MFMA_BF16(a1, xb, acc.t[1]);
}
__device__ __forceinline__ void pe_project_b(const float* sm, const int g, const float px, const float py,
                                             const float pz, const float scale, obj32n::Pe32& pe) {
  pe.t[0] = px * scale;      // `scale` is 1 / obj_scale here (one division per workgroup instead of three per sample)
  pe.t[1] = py * scale;
  pe.t[2] = pz * scale;
  const float* bl = sm + S_PEB + 3 * g;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    // revolutions of a = p pi: a / (2 pi) = p / 2 -- the staged directions are B / 2 (exact), no low part needed
    pe.vh[i] = fmaf(pe.t[2], bl[12 * i + 2], fmaf(pe.t[1], bl[12 * i + 1], pe.t[0] * bl[12 * i]));
    pe.vl[i] = 0.0f;
  }
}

// forward images + small vectors of one object (all threads of the workgroup); `small` = the fp32 block
__device__ __forceinline__ void stage_forward_bf16(char* lds, float* sm, const float* __restrict__ P, const Layout& L,
                                                   int tid, const bool feat, const int b_fl, const int NTHR = objtrain::NTHR) {
  __bf16* img = reinterpret_cast<__bf16*>(lds);
  // element (i, b, g, e) <- W[i][32 b + phi(g, e)]
  for (int x = tid; x < 32 * 3 * 32; x += NTHR) {
    const int i = x / 96, rem = x % 96, b = rem >> 5, ge = rem & 31;
    const int f = 32 * b + phi(ge >> 3, ge & 7);
    img[(B_IN + i * RS_IN) / 2 + rem] = (__bf16)w_in(P, L, i, f);
    img[(B_CL + i * RS_CL) / 2 + rem] = (__bf16)w_cl(P, L, i, f);
    if (feat) img[(b_fl + i * RS_CL) / 2 + rem] = (__bf16)w_fl(P, L, i, f);
  }
  for (int x = tid; x < 32 * 4 * 32; x += NTHR) {
    const int i = x >> 7, rem = x & 127, b = rem >> 5, ge = rem & 31;
    img[(B_CAT + i * RS_CAT) / 2 + rem] = (__bf16)w_cat(P, L, i, 32 * b + phi(ge >> 3, ge & 7));
  }
  for (int x = tid; x < 32 * 32; x += NTHR) {
    const int i = x >> 5, ge = x & 31;
    const int f = phi(ge >> 3, ge & 7);
    img[(B_M1 + i * RS_M) / 2 + ge] = (__bf16)P[L.m1_w + i * H + f];
    img[(B_M2 + i * RS_M) / 2 + ge] = (__bf16)P[L.m2_w + i * H + f];
  }
  for (int i = tid; i < H; i += NTHR) {
    sm[S_BM1 + i] = P[L.m1_b + i];
    sm[S_BM2 + i] = P[L.m2_b + i];
    sm[S_WA + i] = P[L.a_w + i];
  }
  for (int i = tid; i < 3 * H; i += NTHR) sm[S_WOC + i] = P[L.oc_w + i];
  if (tid == 0) sm[S_HB] = P[L.a_b];
  if (tid < 3) sm[S_HB + 1 + tid] = P[L.oc_b + tid];
  // B rows in slot order [slot i][group g][3] (= B's own row-major order), zero for j = 4 i + g >= 21
  for (int i = tid; i < 72; i += NTHR) sm[S_PEB + i] = (i / 3) < OBJ_NDIR ? 0.5f * P[L.pe_b + i] : 0.0f;   // (B / 2: see pe_project_b)
}

}  // namespace bf16k
}  // namespace objtrain
