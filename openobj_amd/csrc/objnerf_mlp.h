// Per-object MLP (hidden = 32) on gfx950: LDS weight image, register-resident forward chain and its
// backward on v_mfma_f32_16x16x4_f32 (exact fp32; C/D: col = lane & 15, row = 4*(lane >> 4) + reg).
//
// "D16 layout": a 32(feature) x 16(sample) fp32 block lives in one wave64 as 8 registers per lane:
// lane l = (c = l & 15 sample column, g = l >> 4), T32.t[tt][r] <-> feature 16*tt + 4*g + r.  A layer's
// output block is directly the B operand of the next layer's MFMAs (k-step (tt, r) consumes register
// (tt, r); the A operand supplies the weight column of the same feature), so activations never leave
// registers between layers, and a wave needs only 8 registers per activation -- two waves per SIMD fit.
//
// Reference math: OccupancyMap.forward (model.py:61-103) on UniDirsEmbed.forward (embedding.py:46-55).
// Biases of in/cat/color/feature layers ride along as an extra weight column against a constant-1
// embedding row; mid1/mid2 biases initialise the accumulator.
#pragma once
#include "objnerf_device.h"

namespace obj32 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 32;
// LDS weight image (float offsets).  Row strides are = 3 (mod 8): forward (row = lane) and transposed
// (column = lane) A-operand reads are then at most 2-way conflicted on 4 of 16 lanes.
constexpr int ST_IN = 99;    // [32][99]   cols 0..86 W_in | 87 bias | 88..98 zero          (x1: 96 rows)
constexpr int ST_M = 35;     // [32][35]
constexpr int ST_CAT = 131;  // [32][131]  cols 0..31 (h2) | 32..118 (x1) | 119 bias | zero
constexpr int ST_CL = 83;    // [32][83]   cols 0..31 (h4) | 32..73 (x2) | 74 bias | zero    (x2: 48 rows)
constexpr int OFF_IN = 0;
constexpr int OFF_M1 = OFF_IN + H * ST_IN;       // 3168
constexpr int OFF_CAT = OFF_M1 + H * ST_M;       // 4288
constexpr int OFF_M2 = OFF_CAT + H * ST_CAT;     // 8480
constexpr int OFF_CL = OFF_M2 + H * ST_M;        // 9600
constexpr int OFF_BM1 = OFF_CL + H * ST_CL;      // 12256
constexpr int OFF_BM2 = OFF_BM1 + H;
constexpr int OFF_WA = OFF_BM2 + H;
constexpr int OFF_WOC = OFF_WA + H;              // [3][32]
constexpr int OFF_HB = OFF_WOC + 3 * H;          // ba, boc[3]
constexpr int OFF_PEB = OFF_HB + 4;              // B doubled: rows 0..32 = B[row % 21]  ([33][3])
constexpr int OFF_FL = 12608;                    // feature layer image [32][83] (only if used)
static_assert(OFF_PEB + 99 <= OFF_FL, "lds image overlap");
constexpr int W_FLOATS_NOFEAT = OFF_FL + 32;
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
  for (int i = tid; i < 33 * 3; i += nthr) lds[OFF_PEB + i] = P[L.pe_b + ((i / 3) % OBJ_NDIR) * 3 + (i % 3)];
  __syncthreads();
}

struct T32 {
  f32x4 t[2];
};
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ T32 zero32() { return T32{{zero4(), zero4()}}; }
__device__ __forceinline__ T32 relu32(const T32& a) {
  T32 o;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[tt][r] = fmaxf(a.t[tt][r], 0.0f);   // (no inline asm: MFMA->VALU hazards are the compiler's)
  return o;
}
__device__ __forceinline__ T32 relu_mask32(const T32& gr, const T32& act) {
  T32 o;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[tt][r] = act.t[tt][r] > 0.0f ? gr.t[tt][r] : 0.0f;
  return o;
}

#define OBJ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// acc(32 out rows) += W[:, col0 + 4g + r] * xt     (xt: one 16-feature tile, feature = 4g + r)
// wl = &W[c * ST + 4 * g]   (c = lane & 15 is the OUTPUT row inside each 16-row out tile)
// (the second out tile's rows sit 16 * ST floats further: that distance goes through `hi16`, an OPAQUE copy of the
// constant, so the compiler keeps a second base register for them and addresses both with DS immediates instead of
// materialising base + large constant with a v_add for every ds_read2 pair)
template <int ST>
__device__ __forceinline__ void mma_fwd16(T32& acc, const float* wl, const int col0, const f32x4& xt, const int hi16) {
  float a0[4], a1[4];
  const float* wl1 = wl + hi16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a0[r] = wl[col0 + r];
    a1[r] = wl1[col0 + r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc.t[0] = OBJ_MFMA(a0[r], xt[r], acc.t[0]);
    acc.t[1] = OBJ_MFMA(a1[r], xt[r], acc.t[1]);
  }
}
template <int ST>
__device__ __forceinline__ void mma_fwd32(T32& acc, const float* wl, const int col0, const T32& x, const int hi16) {
  mma_fwd16<ST>(acc, wl, col0, x.t[0], hi16);
  mma_fwd16<ST>(acc, wl, col0 + 16, x.t[1], hi16);
}
__device__ __forceinline__ int opaque_const(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// acc(16 in rows: col .. col+15) += W[:, col : col+16]^T * d      (d: 32 out rows, D16 layout)
// wt = &W[(4 * g) * ST + c]   (c = lane & 15 is the INPUT feature of the A operand here)
template <int ST>
__device__ __forceinline__ void mma_bwd16(f32x4& acc, const float* wt, const int col, const T32& d) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = wt[(16 * (i >> 2) + (i & 3)) * ST + col];
  // two accumulation chains (even / odd k-steps) keep the dependent-MFMA latency off the critical path
  f32x4 acc2 = zero4();
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    acc = OBJ_MFMA(a[i], d.t[i >> 2][i & 3], acc);
    acc2 = OBJ_MFMA(a[i + 1], d.t[(i + 1) >> 2][(i + 1) & 3], acc2);
  }
  acc += acc2;
}
template <int ST>
__device__ __forceinline__ void mma_bwd32(T32& acc, const float* wt, const int col0, const T32& d) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    float a0[4], a1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0[r] = wt[(16 * tt + r) * ST + col0];
      a1[r] = wt[(16 * tt + r) * ST + col0 + 16];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc.t[0] = OBJ_MFMA(a0[r], d.t[tt][r], acc.t[0]);
      acc.t[1] = OBJ_MFMA(a1[r], d.t[tt][r], acc.t[1]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Positional encoding in D16 layout.  Embedding entry e >= 3 is sin(pi * 2^f * proj[j]) with
// e - 3 = 21 f + j (embedding.py:49-52).  Lane group g owns entries e = e0 + 4 g for compile-time e0,
// i.e. direction (j0 + 4 g) mod 21: every lane keeps the ROTATED, octave-corrected projections
//     ps[i] = proj[(i + 4g) mod 21] * (i + 4g >= 21 ? 2 : 1)
// so that entry e0 + 4g is sin(pi * 2^f0 * ps[j0]) with compile-time (j0, f0) and no per-entry select
// (scaling by powers of two commutes with the reference's fp32 roundings of proj*2^f and of (.)*pi).
// ----------------------------------------------------------------------------------------------
struct Pe {
  float t[3];
  float ps[OBJ_NDIR];
};

__device__ __forceinline__ void pe_project(const float* lds, const int g, const float px, const float py,
                                           const float pz, const float scale, Pe& pe) {
  pe.t[0] = px / scale;        // embedding.py:47
  pe.t[1] = py / scale;
  pe.t[2] = pz / scale;
  const float* bl = lds + OFF_PEB + 12 * g;
#pragma unroll
  for (int i = 0; i < OBJ_NDIR; ++i) {
    const float p = fmaf(pe.t[2], bl[3 * i + 2], fmaf(pe.t[1], bl[3 * i + 1], pe.t[0] * bl[3 * i]));   // :48
    // i + 4g >= 21  <=>  the entry belongs to the next octave; never for i <= 8
    pe.ps[i] = (i > 8 && 4 * g + i >= OBJ_NDIR) ? 2.0f * p : p;
  }
}

// band value (or its derivative w.r.t. ps[j0]) for rotated index q0 = 21 f0 + j0, q0 in [-3, 125]
template <bool WANT_COS>
__device__ __forceinline__ float pe_band(const Pe& pe, const int q0) {
  const int qq = q0 + OBJ_NDIR;                 // >= 18
  const int j0 = qq % OBJ_NDIR, f0 = qq / OBJ_NDIR - 1;
  const float sc = f0 < 0 ? 0.5f : (float)(1 << (f0 < 0 ? 0 : f0));
  if (!WANT_COS) return band_sin(pe.ps[j0], sc);
  float sv, dc;
  band_sincos(pe.ps[j0], sc, sv, dc);
  return dc;
}

// x1 tile T (0..5): entries e = 16 T + 4 g + r  (87 = constant 1, >= 88 zero)
__device__ __forceinline__ f32x4 pe_x1_tile(const Pe& pe, const int T, const int g) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = pe_band<false>(pe, 16 * T + r - 3);
    if (T == 0 && r < 3) v = (g == 0) ? pe.t[r] : v;
    if (T == 5) {
      if (r == 3) v = (g == 1) ? 1.0f : v;
      v = (g >= 2) ? 0.0f : v;
    }
    o[r] = v;
  }
  return o;
}
// x2 tile T (0..2): entries e2 = 16 T + 4 g + r  <->  embedding entry 87 + e2  (42 = constant 1, > 42 zero)
__device__ __forceinline__ f32x4 pe_x2_tile(const Pe& pe, const int T, const int g) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = pe_band<false>(pe, 84 + 16 * T + r);
    if (T == 2) {
      if (r == 2) v = (g == 2) ? 1.0f : v;
      if (r == 3) v = (g == 2) ? 0.0f : v;
      v = (g == 3) ? 0.0f : v;
    }
    o[r] = v;
  }
  return o;
}
// d ps[j] += d_x[e] * d sin / d ps  for one tile (entries that are not sin bands contribute nothing)
__device__ __forceinline__ void pe_x1_tile_bwd(const Pe& pe, const int T, const int g, const f32x4& dx,
                                               float (&dps)[OBJ_NDIR]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = 16 * T + r - 3 + OBJ_NDIR;
    float v = dx[r] * pe_band<true>(pe, 16 * T + r - 3);
    if (T == 0 && r < 3) v = (g == 0) ? 0.0f : v;
    if (T == 5) {
      if (r == 3) v = (g == 1) ? 0.0f : v;
      v = (g >= 2) ? 0.0f : v;
    }
    dps[qq % OBJ_NDIR] += v;
  }
}
__device__ __forceinline__ void pe_x2_tile_bwd(const Pe& pe, const int T, const int g, const f32x4& dx,
                                               float (&dps)[OBJ_NDIR]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = 84 + 16 * T + r + OBJ_NDIR;
    float v = dx[r] * pe_band<true>(pe, 84 + 16 * T + r);
    if (T == 2) {
      if (r >= 2) v = (g == 2) ? 0.0f : v;
      v = (g == 3) ? 0.0f : v;
    }
    dps[qq % OBJ_NDIR] += v;
  }
}

// Backward of one tile that ALSO re-creates the tile's forward values (one sincos gives both), so the
// embedding does not have to stay live between the forward and the backward pass.
__device__ __forceinline__ f32x4 pe_x1_tile_fb(const Pe& pe, const int T, const int g, const f32x4& dx,
                                               float (&dps)[OBJ_NDIR]) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q0 = 16 * T + r - 3, qq = q0 + OBJ_NDIR;
    const int j0 = qq % OBJ_NDIR, f0 = qq / OBJ_NDIR - 1;
    const float sc = f0 < 0 ? 0.5f : (float)(1 << (f0 < 0 ? 0 : f0));
    float sv, dc;
    band_sincos(pe.ps[j0], sc, sv, dc);
    float v = dx[r] * dc;
    if (T == 0 && r < 3) { v = (g == 0) ? 0.0f : v; sv = (g == 0) ? pe.t[r] : sv; }
    if (T == 5) {
      if (r == 3) { v = (g == 1) ? 0.0f : v; sv = (g == 1) ? 1.0f : sv; }
      v = (g >= 2) ? 0.0f : v;
      sv = (g >= 2) ? 0.0f : sv;
    }
    dps[j0] += v;
    asm volatile("" : "+v"(dps[j0]));   // consume v NOW: a deferred add keeps dx and cos live
    o[r] = sv;
  }
  return o;
}
__device__ __forceinline__ f32x4 pe_x2_tile_fb(const Pe& pe, const int T, const int g, const f32x4& dx,
                                               float (&dps)[OBJ_NDIR]) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q0 = 84 + 16 * T + r, qq = q0 + OBJ_NDIR;
    const int j0 = qq % OBJ_NDIR, f0 = qq / OBJ_NDIR - 1;
    const float sc = (float)(1 << f0);
    float sv, dc;
    band_sincos(pe.ps[j0], sc, sv, dc);
    float v = dx[r] * dc;
    if (T == 2) {
      if (r == 2) { v = (g == 2) ? 0.0f : v; sv = (g == 2) ? 1.0f : sv; }
      if (r == 3) { v = (g == 2) ? 0.0f : v; sv = (g == 2) ? 0.0f : sv; }
      v = (g == 3) ? 0.0f : v;
      sv = (g == 3) ? 0.0f : sv;
    }
    dps[j0] += v;
    asm volatile("" : "+v"(dps[j0]));   // consume v NOW: a deferred add keeps dx and cos live
    o[r] = sv;
  }
  return o;
}

struct Emb {
  f32x4 x1[6];
  f32x4 x2[3];
};
__device__ __forceinline__ void embed(Emb& e, const Pe& pe, const int g) {
#pragma unroll
  for (int T = 0; T < 6; ++T) e.x1[T] = pe_x1_tile(pe, T, g);
#pragma unroll
  for (int T = 0; T < 3; ++T) e.x2[T] = pe_x2_tile(pe, T, g);
}
// embedding supplied by the caller (OccupancyMap.forward on an explicit embedding tensor)
__device__ __forceinline__ void embed_load(Emb& e, const float* __restrict__ emb, const int g) {
#pragma unroll
  for (int T = 0; T < 6; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T + 4 * g + r;
      e.x1[T][r] = i < OBJ_E1 ? emb[i] : (i == OBJ_E1 ? 1.0f : 0.0f);
    }
#pragma unroll
  for (int T = 0; T < 3; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T + 4 * g + r;
      e.x2[T][r] = i < OBJ_E2 ? emb[OBJ_E1 + i] : (i == OBJ_E2 ? 1.0f : 0.0f);
    }
}

// Forward activations of one 16-sample block.
struct Acts {
  T32 h1, h2, h3, h4, hc, hf;
};
struct Heads {
  float alpha;      // 10 * raw (model.py:88)
  float col[3];     // sigmoid applied (model.py:96)
};

__device__ __forceinline__ float xgroup_sum(float v) {   // sum over the 4 lane groups of a sample, on the VALU
  // v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second: with both = v the
  // two results are (r0,r0,r2,r2) and (r1,r1,r3,r3); v_permlane32_swap likewise for the 32-lane halves.
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s16 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned u2 = __float_as_uint(s16);
  const auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <bool FEAT>
__device__ __forceinline__ void mlp_forward(const float* lds, const int c, const int g, const Emb& e, Acts& a,
                                            Heads& hd) {
  const float* wl_in = lds + opaque_const(OFF_IN) + c * ST_IN + 4 * g;
  const float* wl_m1 = lds + opaque_const(OFF_M1) + c * ST_M + 4 * g;
  const float* wl_cat = lds + opaque_const(OFF_CAT) + c * ST_CAT + 4 * g;
  const float* wl_m2 = lds + opaque_const(OFF_M2) + c * ST_M + 4 * g;
  const float* wl_cl = lds + opaque_const(OFF_CL) + c * ST_CL + 4 * g;
  const int h_in = opaque_const(16 * ST_IN), h_m = opaque_const(16 * ST_M), h_cat = opaque_const(16 * ST_CAT),
            h_cl = opaque_const(16 * ST_CL);
  T32 acc = zero32();
#pragma unroll
  for (int T = 0; T < 6; ++T) mma_fwd16<ST_IN>(acc, wl_in, 16 * T, e.x1[T], h_in);
  a.h1 = relu32(acc);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[tt][r] = lds[OFF_BM1 + 16 * tt + 4 * g + r];
  mma_fwd32<ST_M>(acc, wl_m1, 0, a.h1, h_m);
  a.h2 = relu32(acc);
  acc = zero32();
  mma_fwd32<ST_CAT>(acc, wl_cat, 0, a.h2, h_cat);
#pragma unroll
  for (int T = 0; T < 6; ++T) mma_fwd16<ST_CAT>(acc, wl_cat, 32 + 16 * T, e.x1[T], h_cat);
  a.h3 = relu32(acc);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[tt][r] = lds[OFF_BM2 + 16 * tt + 4 * g + r];
  mma_fwd32<ST_M>(acc, wl_m2, 0, a.h3, h_m);
  a.h4 = relu32(acc);
  acc = zero32();
  mma_fwd32<ST_CL>(acc, wl_cl, 0, a.h4, h_cl);
#pragma unroll
  for (int T = 0; T < 3; ++T) mma_fwd16<ST_CL>(acc, wl_cl, 32 + 16 * T, e.x2[T], h_cl);
  a.hc = relu32(acc);
  if (FEAT) {
    const float* wl_fl = lds + opaque_const(OFF_FL) + c * ST_CL + 4 * g;
    acc = zero32();
    mma_fwd32<ST_CL>(acc, wl_fl, 0, a.h4, h_cl);
#pragma unroll
    for (int T = 0; T < 3; ++T) mma_fwd16<ST_CL>(acc, wl_fl, 32 + 16 * T, e.x2[T], h_cl);
    a.hf = relu32(acc);
  }
  // heads: each lane group holds 8 of the 32 hidden rows of its sample
  float pa = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * tt + 4 * g + r;
      pa = fmaf(lds[OFF_WA + row], a.h4.t[tt][r], pa);
      pc0 = fmaf(lds[OFF_WOC + row], a.hc.t[tt][r], pc0);
      pc1 = fmaf(lds[OFF_WOC + H + row], a.hc.t[tt][r], pc1);
      pc2 = fmaf(lds[OFF_WOC + 2 * H + row], a.hc.t[tt][r], pc2);
    }
  hd.alpha = (xgroup_sum(pa) + lds[OFF_HB]) * 10.0f;
  hd.col[0] = sigmoid_acc(xgroup_sum(pc0) + lds[OFF_HB + 1]);
  hd.col[1] = sigmoid_acc(xgroup_sum(pc1) + lds[OFF_HB + 2]);
  hd.col[2] = sigmoid_acc(xgroup_sum(pc2) + lds[OFF_HB + 3]);
}

}  // namespace obj32
