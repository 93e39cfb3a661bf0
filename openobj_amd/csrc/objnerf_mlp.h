// Per-object MLP (hidden = 32) on gfx950: LDS weight image, register-resident forward chain and
// its backward (dgrad in registers, wgrad through an LDS transpose) on v_mfma_f32_32x32x2_f32.
//
// Reference math: OccupancyMap.forward (model.py:61-103) on UniDirsEmbed.forward
// (embedding.py:46-55).  Biases of in/cat/color/feature layers ride along as an extra weight
// column against a constant-1 embedding row; mid1/mid2 biases initialise the accumulator.
#pragma once
#include "objnerf_device.h"

namespace obj32 {

constexpr int H = 32;
// LDS weight image (float offsets); row strides are odd so the 32 output rows hit 32 banks.
constexpr int ST_IN = 89;    // [32][89]  cols 0..86 W_in, 87 bias, 88 zero
constexpr int ST_M = 33;     // [32][33]
constexpr int ST_CAT = 121;  // [32][121] cols 0..31 (h2) | 32..118 (x1) | 119 bias | 120 zero
constexpr int ST_CL = 81;    // [32][81]  cols 0..31 (h4) | 32..73 (x2) | 74 bias | 75..80 zero
constexpr int OFF_IN = 0;
constexpr int OFF_M1 = OFF_IN + H * ST_IN;       // 2848
constexpr int OFF_CAT = OFF_M1 + H * ST_M;       // 3904
constexpr int OFF_M2 = OFF_CAT + H * ST_CAT;     // 7776
constexpr int OFF_CL = OFF_M2 + H * ST_M;        // 8832
constexpr int OFF_BM1 = OFF_CL + H * ST_CL;      // 11424
constexpr int OFF_BM2 = OFF_BM1 + H;
constexpr int OFF_WA = OFF_BM2 + H;
constexpr int OFF_WOC = OFF_WA + H;              // [3][32]
constexpr int OFF_HB = OFF_WOC + 3 * H;          // ba, boc[3]
constexpr int OFF_PEB = OFF_HB + 4;              // B [21][3]
constexpr int OFF_FL = 11712;                    // feature layer image [32][81] (only if used)
static_assert(OFF_PEB + 63 <= OFF_FL, "lds image overlap");
constexpr int W_FLOATS_NOFEAT = OFF_FL + 32;     // + slack for the over-reads of the T blocks
constexpr int W_FLOATS_FEAT = OFF_FL + H * ST_CL + 32;

// arena offsets of the 19 tensors of one object (objnerf_param_layout order)
struct Layout {
  int in_w, in_b, m1_w, m1_b, cat_w, cat_b, m2_w, m2_b, a_w, a_b, cl_w, cl_b, oc_w, oc_b, fl_w, fl_b,
      of_w, of_b, pe_b, total;
};
__host__ __device__ inline Layout make_layout(int C) {
  Layout L;
  int o = 0;
  L.in_w = o; o += H * OBJ_E1;
  L.in_b = o; o += H;
  L.m1_w = o; o += H * H;
  L.m1_b = o; o += H;
  L.cat_w = o; o += H * (H + OBJ_E1);
  L.cat_b = o; o += H;
  L.m2_w = o; o += H * H;
  L.m2_b = o; o += H;
  L.a_w = o; o += H;
  L.a_b = o; o += 1;
  L.cl_w = o; o += H * (H + OBJ_E2);
  L.cl_b = o; o += H;
  L.oc_w = o; o += 3 * H;
  L.oc_b = o; o += 3;
  L.fl_w = o; o += H * (H + OBJ_E2);
  L.fl_b = o; o += H;
  L.of_w = o; o += C * H;
  L.of_b = o; o += C;
  L.pe_b = o; o += OBJ_NDIR * 3;
  L.total = o;
  return L;
}

// Stage one object's weights into the LDS image.  All threads of the workgroup call this.
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ P, const Layout& L,
                                              bool with_feat, int tid, int nthr) {
  const int total = with_feat ? W_FLOATS_FEAT : W_FLOATS_NOFEAT;
  for (int i = tid; i < total; i += nthr) lds[i] = 0.0f;
  __syncthreads();
  for (int i = tid; i < H * OBJ_E1; i += nthr) lds[OFF_IN + (i / OBJ_E1) * ST_IN + (i % OBJ_E1)] = P[L.in_w + i];
  for (int i = tid; i < H * H; i += nthr) {
    lds[OFF_M1 + (i / H) * ST_M + (i % H)] = P[L.m1_w + i];
    lds[OFF_M2 + (i / H) * ST_M + (i % H)] = P[L.m2_w + i];
  }
  for (int i = tid; i < H * (H + OBJ_E1); i += nthr)
    lds[OFF_CAT + (i / (H + OBJ_E1)) * ST_CAT + (i % (H + OBJ_E1))] = P[L.cat_w + i];
  for (int i = tid; i < H * (H + OBJ_E2); i += nthr) {
    lds[OFF_CL + (i / (H + OBJ_E2)) * ST_CL + (i % (H + OBJ_E2))] = P[L.cl_w + i];
    if (with_feat) lds[OFF_FL + (i / (H + OBJ_E2)) * ST_CL + (i % (H + OBJ_E2))] = P[L.fl_w + i];
  }
  for (int i = tid; i < H; i += nthr) {
    lds[OFF_IN + i * ST_IN + OBJ_E1] = P[L.in_b + i];
    lds[OFF_CAT + i * ST_CAT + H + OBJ_E1] = P[L.cat_b + i];
    lds[OFF_CL + i * ST_CL + H + OBJ_E2] = P[L.cl_b + i];
    if (with_feat) lds[OFF_FL + i * ST_CL + H + OBJ_E2] = P[L.fl_b + i];
    lds[OFF_BM1 + i] = P[L.m1_b + i];
    lds[OFF_BM2 + i] = P[L.m2_b + i];
    lds[OFF_WA + i] = P[L.a_w + i];
  }
  for (int i = tid; i < 3 * H; i += nthr) lds[OFF_WOC + i] = P[L.oc_w + i];
  if (tid == 0) lds[OFF_HB] = P[L.a_b];
  if (tid < 3) lds[OFF_HB + 1 + tid] = P[L.oc_b + tid];
  for (int i = tid; i < OBJ_NDIR * 3; i += nthr) lds[OFF_PEB + i] = P[L.pe_b + i];
  __syncthreads();
}

// acc += W[:, col0 + (feature rows of x)] * x       (x: a 32-feature block in D layout)
// wl = &W[c * stride + 4*kh]  (c = lane & 31 is the OUTPUT row of the A operand here)
template <int NSTEPS>
__device__ __forceinline__ void mma_fwd(f32x16& acc, const float* wl, const int col0, const f32x16& x) {
#pragma unroll
  for (int r = 0; r < NSTEPS; ++r)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[col0 + row0(r)], x[r], acc, 0, 0, 0);
}
// acc += W[:, col0 : col0+32]^T * d     (d: 32 output-feature rows in D layout)
// wt = &W[(4*kh) * STRIDE + c]   (c = lane & 31 is the INPUT feature of the A operand here)
template <int STRIDE>
__device__ __forceinline__ void mma_bwd(f32x16& acc, const float* wt, const int col0, const f32x16& d) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[row0(r) * STRIDE + col0], d[r], acc, 0, 0, 0);
}

__device__ __forceinline__ f32x16 relu16(const f32x16& a) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = fmaxf(a[r], 0.0f);
  return o;
}
__device__ __forceinline__ f32x16 relu_mask16(const f32x16& g, const f32x16& act) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = act[r] > 0.0f ? g[r] : 0.0f;
  return o;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.0f;
  return o;
}

// Embedding of this lane's sample in B-operand / D layout: x1 = 3 blocks (88 rows), x2 = 2 blocks.
struct Emb {
  f32x16 x1[3];
  f32x16 x2[2];
};

__device__ __forceinline__ void project(const float* lds, const float px, const float py, const float pz,
                                        const float scale, float (&t)[3], float (&proj)[OBJ_NDIR]) {
  t[0] = px / scale;
  t[1] = py / scale;
  t[2] = pz / scale;
#pragma unroll
  for (int j = 0; j < OBJ_NDIR; ++j) {
    const float b0 = lds[OFF_PEB + 3 * j], b1 = lds[OFF_PEB + 3 * j + 1], b2 = lds[OFF_PEB + 3 * j + 2];
    proj[j] = fmaf(t[2], b2, fmaf(t[1], b1, t[0] * b0));
  }
}

__device__ __forceinline__ void embed(Emb& e, const int kh, const float (&t)[3], const float (&proj)[OBJ_NDIR]) {
#pragma unroll
  for (int b = 0; b < 3; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e0 = 32 * b + row0(r);
      e.x1[b][r] = (e0 + 4 <= OBJ_E1) ? pe_lane_value<false>(pe_sel_x1(e0), pe_sel_x1(e0 + 4), kh, t, proj) : 0.0f;
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e0 = 32 * b + row0(r);
      e.x2[b][r] = (e0 <= OBJ_E2) ? pe_lane_value<false>(pe_sel_x2(e0), pe_sel_x2(e0 + 4), kh, t, proj) : 0.0f;
    }
  }
}

// Forward activations of one 32-sample tile (all in D layout).
struct Acts {
  f32x16 h1, h2, h3, h4, hc, hf;
};

struct Heads {
  float alpha;      // 10 * raw (model.py:88)
  float col[3];     // sigmoid applied (model.py:96)
};

template <bool FEAT>
__device__ __forceinline__ void mlp_forward(const float* lds, const int c, const int kh, const Emb& e,
                                            Acts& a, Heads& hd) {
  const float* wl_in = lds + OFF_IN + c * ST_IN + 4 * kh;
  const float* wl_m1 = lds + OFF_M1 + c * ST_M + 4 * kh;
  const float* wl_cat = lds + OFF_CAT + c * ST_CAT + 4 * kh;
  const float* wl_m2 = lds + OFF_M2 + c * ST_M + 4 * kh;
  const float* wl_cl = lds + OFF_CL + c * ST_CL + 4 * kh;
  f32x16 acc = zero16();
  mma_fwd<16>(acc, wl_in, 0, e.x1[0]);
  mma_fwd<16>(acc, wl_in, 32, e.x1[1]);
  mma_fwd<12>(acc, wl_in, 64, e.x1[2]);
  a.h1 = relu16(acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = lds[OFF_BM1 + row0(r) + 4 * kh];
  mma_fwd<16>(acc, wl_m1, 0, a.h1);
  a.h2 = relu16(acc);
  acc = zero16();
  mma_fwd<16>(acc, wl_cat, 0, a.h2);
  mma_fwd<16>(acc, wl_cat, 32, e.x1[0]);
  mma_fwd<16>(acc, wl_cat, 64, e.x1[1]);
  mma_fwd<12>(acc, wl_cat, 96, e.x1[2]);
  a.h3 = relu16(acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = lds[OFF_BM2 + row0(r) + 4 * kh];
  mma_fwd<16>(acc, wl_m2, 0, a.h3);
  a.h4 = relu16(acc);
  acc = zero16();
  mma_fwd<16>(acc, wl_cl, 0, a.h4);
  mma_fwd<16>(acc, wl_cl, 32, e.x2[0]);
  mma_fwd<7>(acc, wl_cl, 64, e.x2[1]);
  a.hc = relu16(acc);
  if (FEAT) {
    const float* wl_fl = lds + OFF_FL + c * ST_CL + 4 * kh;
    acc = zero16();
    mma_fwd<16>(acc, wl_fl, 0, a.h4);
    mma_fwd<16>(acc, wl_fl, 32, e.x2[0]);
    mma_fwd<7>(acc, wl_fl, 64, e.x2[1]);
    a.hf = relu16(acc);
  }
  // heads: each half holds 16 of the 32 hidden rows of its sample
  float pa = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0(r) + 4 * kh;
    pa = fmaf(lds[OFF_WA + row], a.h4[r], pa);
    pc0 = fmaf(lds[OFF_WOC + row], a.hc[r], pc0);
    pc1 = fmaf(lds[OFF_WOC + H + row], a.hc[r], pc1);
    pc2 = fmaf(lds[OFF_WOC + 2 * H + row], a.hc[r], pc2);
  }
  hd.alpha = (xhalf_sum(pa) + lds[OFF_HB]) * 10.0f;
  hd.col[0] = sigmoid_acc(xhalf_sum(pc0) + lds[OFF_HB + 1]);
  hd.col[1] = sigmoid_acc(xhalf_sum(pc1) + lds[OFF_HB + 2]);
  hd.col[2] = sigmoid_acc(xhalf_sum(pc2) + lds[OFF_HB + 3]);
}

}  // namespace obj32
