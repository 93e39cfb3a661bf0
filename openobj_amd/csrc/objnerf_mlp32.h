// Per-object MLP (hidden = 32), fp32, second generation of the register-resident MFMA chain (gfx950).
//
// What changed against objnerf_mlp.h (which the bf16 kernel still uses) and why -- tools/ubench_alu.hip measured that a
// v_mfma_f32_16x16x4_f32 occupies the SIMD's vector ALU for its whole 32 cycles (VALU work of either wave of the SIMD
// does not overlap it), so the fused kernel's time is  34 * #MFMA + ~3..5 * #VALU  cycles and every VALU instruction
// and every non-algorithmic MFMA counts:
//
//  * DIRECTION-OWNER positional encoding.  Lane group g of a sample owns directions j = 4 i + g (slot i = 0..5) in ALL
//    six octaves.  The reference's argument fp32(fp32(p 2^f) pi) equals 2^f fp32(p pi) exactly, so one double-float
//    division by 2 pi per direction serves all octaves (4 VALU + v_sin / v_cos per entry instead of ~14 + 2), the
//    projection gradient d p_j is complete inside the owning lane (no cross-group sum: the 34 selection MFMAs and the
//    staging round trip of the first generation are gone) and d B accumulates in 18 registers per lane.
//  * IN-MAJOR weight image  W^T[input k][32 outputs]  with 40-float rows: the forward A operand of a k-step is ONE
//    ds_read_b64 (both 16-row output tiles) at a compile-time offset from a per-lane base, the transposed (input
//    gradient) operand TWO ds_read_b128 per 16-input tile; both conflict-free (row pitch 40 floats = 32 banks (mod 64)
//    for rows 4 apart; 16-byte chunks rotated by lane-group parity for the b128 groups).  The first generation issued
//    one 2-way-conflicted ds_read2_b32 per k-step and eight ds_read_b32 per transposed tile (36 % LDS bank conflicts).
//
// K-order: k-step (T, r) of a 16-feature tile T takes feature kappa = 16 T + 4 g + r from lane group g (so a layer's
// 32 x 16 accumulator block is the next layer's B operand without moving).  Hidden features use kappa = feature.
// Embedding entries: lane (sample c, group g) holds x1[t], t = 0..23 (kappa = 16 (t >> 2) + 4 g + (t & 3)):
//   t = 4 i + f, slot i, octave f = 0..3:  sin(2^f a_j), j = 4 i + g  (valid while j < 21, i.e. i < 5 or g == 0) -- a
//   16-row tile of the embedding is ONE direction slot with its four octaves;
//   the 12 free places carry  x / scale (f = 0, i = 5, g = 1..3),  the constant 1 that multiplies the bias column
//   (f = 1, i = 5, g = 1)  and zeros;   x2[t] = octave 4 + (t & 1) of slot t >> 1, the constant at (slot 5, octave 4, g = 1).
// OBJ_PE_ANCHORS (bit f set: octave f gets its own exact range reduction + v_sin / v_cos; clear: it is derived from
// octave f - 1 by angle doubling, 3 VALU instead of 4 + 2 transcendentals, error doubling per step).  Default 17 =
// octaves 0 and 4 anchored: embedding error <= ~2e-6 (octave 3) against ~3e-7 with all six anchored; measured on the
// 50 x 4096 x 64 step (MI355X): 63 -> 10.08 ms, 21 -> 9.89, 17 -> 9.77, 1 -> 9.65; every parity test passes with each.
// Reference: OccupancyMap.forward (model.py:61-103) on UniDirsEmbed.forward (embedding.py:46-55).
#pragma once
#include "objnerf_mlp.h"

namespace obj32n {
using namespace obj32;

constexpr int WROW = 40;                 // floats per image row (32 outputs + 8 pad)
constexpr int R_IN = 0;                  // image rows: in-layer  [x1 96]
constexpr int R_M1 = R_IN + 96;          //             mid1      [h1 32]
constexpr int R_CAT = R_M1 + 32;         //             cat       [h2 32 | x1 96]
constexpr int R_M2 = R_CAT + 128;        //             mid2      [h3 32]
constexpr int R_CL = R_M2 + 32;          //             colour    [h4 32 | x2 48]
constexpr int R_FL = R_CL + 80;          //             feature   [h4 32 | x2 48]   (only if used)
constexpr int ROWS_NOFEAT = R_FL, ROWS_FEAT = R_FL + 80;
// small vectors behind the image (float offsets from the start of the image)
constexpr int SV_BM1 = 0, SV_BM2 = 32, SV_WA = 64, SV_WOC = 96, SV_HB = 192, SV_PEB = 196, SV_FLOATS = 196 + 72 + 4;
__host__ __device__ constexpr int img_floats(bool feat) { return (feat ? ROWS_FEAT : ROWS_NOFEAT) * WROW + SV_FLOATS; }
__host__ __device__ constexpr int sv_base(bool feat) { return (feat ? ROWS_FEAT : ROWS_NOFEAT) * WROW; }

constexpr int BIAS_COL = -2, ZERO_COL = -1;
// reference column (inside emb[:87]) of x1 entry (t, g); BIAS_COL for the constant-1 entry, ZERO_COL for padding
__host__ __device__ inline int x1_col(int t, int g) {
  const int i = t >> 2, f = t & 3, j = 4 * i + g;
  if (j < OBJ_NDIR) return 3 + OBJ_NDIR * f + j;
  if (f == 0) return g - 1;                      // x / scale, components 0..2 (i == 5, g = 1..3)
  if (f == 1 && g == 1) return BIAS_COL;
  return ZERO_COL;
}
// reference column (inside emb[87:]) of x2 entry (t, g)
__host__ __device__ inline int x2_col(int t, int g) {
  const int i = t >> 1, f = t & 1, j = 4 * i + g;
  if (j < OBJ_NDIR) return OBJ_NDIR * f + j;
  if (f == 0 && g == 1) return BIAS_COL;
  return ZERO_COL;
}
// kappa (0..95 / 0..47) -> (t, g)
__host__ __device__ inline void kappa_tg(int kappa, int& t, int& g) {
  t = 4 * (kappa >> 4) + (kappa & 3);
  g = (kappa >> 2) & 3;
}
// float position of output o inside an image row: the pair (o, o + 16) is adjacent (one ds_read_b64 in the forward),
// the pairs r = 0..3 of a 4-output group are contiguous (two ds_read_b128 in the transposed direction) and rotated by
// two pairs for odd groups, which puts the b128 lane groups on disjoint banks (see the header comment)
__host__ __device__ inline int out_pos(int o) {
  const int tt = o >> 4, cc = o & 15, gg = cc >> 2, rr = cc & 3;
  return 8 * gg + 2 * ((rr + 2 * (gg & 1)) & 3) + tt;
}

// Stage one object's weights into the LDS image.  All threads of the workgroup call this.
__device__ __forceinline__ void stage_weights32(float* lds, const float* __restrict__ P, const Layout& L, bool with_feat,
                                                int tid, int nthr) {
  const int rows = with_feat ? ROWS_FEAT : ROWS_NOFEAT;
  const int total = rows * WROW + SV_FLOATS;
  for (int i = tid; i < total; i += nthr) lds[i] = 0.0f;
  __syncthreads();
  for (int e = tid; e < rows * H; e += nthr) {
    const int row = e >> 5, o = e & 31;
    float v = 0.0f;
    int w_off, b_off, ncols, col;
    int local;
    if (row < R_M1) {                      // in-layer: x1 rows
      local = row - R_IN; w_off = L.in_w; b_off = L.in_b; ncols = OBJ_E1;
      int t, g; kappa_tg(local, t, g); col = x1_col(t, g);
    } else if (row < R_CAT) {
      local = row - R_M1; w_off = L.m1_w; b_off = -1; ncols = H; col = local;
    } else if (row < R_M2) {
      local = row - R_CAT; w_off = L.cat_w; b_off = L.cat_b; ncols = H + OBJ_E1;
      if (local < H) col = local;
      else { int t, g; kappa_tg(local - H, t, g); col = x1_col(t, g); if (col >= 0) col += H; }
    } else if (row < R_CL) {
      local = row - R_M2; w_off = L.m2_w; b_off = -1; ncols = H; col = local;
    } else {
      const bool fl = row >= R_FL;
      local = row - (fl ? R_FL : R_CL); w_off = fl ? L.fl_w : L.cl_w; b_off = fl ? L.fl_b : L.cl_b; ncols = H + OBJ_E2;
      if (local < H) col = local;
      else { int t, g; kappa_tg(local - H, t, g); col = x2_col(t, g); if (col >= 0) col += H; }
    }
    if (col >= 0) v = P[w_off + o * ncols + col];
    else if (col == BIAS_COL && b_off >= 0) v = P[b_off + o];
    lds[row * WROW + out_pos(o)] = v;
  }
  float* sv = lds + rows * WROW;
  for (int i = tid; i < H; i += nthr) {
    sv[SV_BM1 + i] = P[L.m1_b + i];
    sv[SV_BM2 + i] = P[L.m2_b + i];
    sv[SV_WA + i] = P[L.a_w + i];
  }
  for (int i = tid; i < 3 * H; i += nthr) sv[SV_WOC + i] = P[L.oc_w + i];
  if (tid == 0) sv[SV_HB] = P[L.a_b];
  if (tid < 3) sv[SV_HB + 1 + tid] = P[L.oc_b + tid];
  // B rows in slot order: [slot i][group g][3], zero for j = 4 i + g >= 21
  for (int i = tid; i < 72; i += nthr) {
    const int j = i / 3;              // = 4 * slot + g
    sv[SV_PEB + i] = j < OBJ_NDIR ? P[L.pe_b + i] : 0.0f;
  }
  __syncthreads();
}

// ----------------------------------------------------------------------------------------------------------------
// MFMA building blocks.  wf = lds + 4 g * WROW + out_pos(c) (forward base), wt0 / wt1 = transposed bases (below).
// ----------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc(32 outputs) += W[:, features of tile (row0 .. row0 + 15)] * xt     (xt: one 16-feature tile in K-order)
#ifndef OBJ32_ABL
#define OBJ32_ABL 0          // objnerf_train32.hip: ceiling-measurement builds only
#endif
__device__ __forceinline__ void mma_f16(T32& acc, const float* wf, const int row0, const f32x4& xt) {
  if ((OBJ32_ABL) & 64) { asm volatile("" :: "v"(xt)); return; }
  f32x2 a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const f32x2*>(wf + (row0 + r) * WROW);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc.t[0] = OBJ_MFMA(a[r][0], xt[r], acc.t[0]);
    acc.t[1] = OBJ_MFMA(a[r][1], xt[r], acc.t[1]);
  }
}
__device__ __forceinline__ void mma_f32(T32& acc, const float* wf, const int row0, const T32& x) {
  mma_f16(acc, wf, row0, x.t[0]);
  mma_f16(acc, wf, row0 + 16, x.t[1]);
}
// acc(16 inputs: image rows row0 .. row0 + 15) += W[:, those inputs]^T * d    (d: 32 outputs)
// wt0 = lds + c * WROW + 8 g + 4 (g & 1), wt1 = lds + c * WROW + 8 g + 4 (1 - (g & 1)): after the rotation of out_pos
// both lane-group parities find the pairs r = 0, 1 at wt0 and r = 2, 3 at wt1
template <bool ONE_CHAIN = false>
__device__ __forceinline__ void mma_t16(f32x4& acc, const float* wt0, const float* wt1, const int row0, const T32& d) {
  if ((OBJ32_ABL) & 64) { asm volatile("" :: "v"(d.t[0]), "v"(d.t[1])); return; }
  const f32x4 lo = *reinterpret_cast<const f32x4*>(wt0 + row0 * WROW);    // (r0 t0, r0 t1, r1 t0, r1 t1)
  const f32x4 hi = *reinterpret_cast<const f32x4*>(wt1 + row0 * WROW);    // (r2 t0, r2 t1, r3 t0, r3 t1)
  // ONE_CHAIN (round 6; the headline instantiation of train_fused32_kernel): one dependent chain of eight MFMAs instead of
  // two chains of four and a join (4 zero moves + 4 adds per call, 23 calls per tile).  A dependent
  // v_mfma_f32_16x16x4_f32 issues every 34 cycles instead of 32, which the other wave of the SIMD absorbs, while the eight
  // vector instructions of the join overlap with nothing (section 4.1).  With the weight-gradient loops fully unrolled:
  // 8.90 -> 8.82 ms.  NOT for the feature instantiation: there the longer live ranges spill (12.0 -> 19.7 ms).
  if (ONE_CHAIN) {
  acc = OBJ_MFMA(lo[0], d.t[0][0], acc);
  acc = OBJ_MFMA(lo[1], d.t[1][0], acc);
  acc = OBJ_MFMA(lo[2], d.t[0][1], acc);
  acc = OBJ_MFMA(lo[3], d.t[1][1], acc);
  acc = OBJ_MFMA(hi[0], d.t[0][2], acc);
  acc = OBJ_MFMA(hi[1], d.t[1][2], acc);
  acc = OBJ_MFMA(hi[2], d.t[0][3], acc);
  acc = OBJ_MFMA(hi[3], d.t[1][3], acc);
  return;
  }
  f32x4 acc2 = zero4();
  acc = OBJ_MFMA(lo[0], d.t[0][0], acc);
  acc2 = OBJ_MFMA(lo[1], d.t[1][0], acc2);
  acc = OBJ_MFMA(lo[2], d.t[0][1], acc);
  acc2 = OBJ_MFMA(lo[3], d.t[1][1], acc2);
  acc = OBJ_MFMA(hi[0], d.t[0][2], acc);
  acc2 = OBJ_MFMA(hi[1], d.t[1][2], acc2);
  acc = OBJ_MFMA(hi[2], d.t[0][3], acc);
  acc2 = OBJ_MFMA(hi[3], d.t[1][3], acc2);
  acc += acc2;
}
template <bool ONE_CHAIN = false>
__device__ __forceinline__ void mma_t32(T32& acc, const float* wt0, const float* wt1, const int row0, const T32& d) {
  mma_t16<ONE_CHAIN>(acc.t[0], wt0, wt1, row0, d);
  mma_t16<ONE_CHAIN>(acc.t[1], wt0, wt1, row0 + 16, d);
}

// ----------------------------------------------------------------------------------------------------------------
// Positional encoding, direction-owner layout.
// ----------------------------------------------------------------------------------------------------------------
struct Pe32 {
  float t[3];          // x / scale (embedding.py:47)
  float vh[6], vl[6];  // a_j / (2 pi) as an unevaluated sum, a_j = fp32(proj_j * pi), j = 4 i + g
};

#define OBJ_INV2PI_HI 0.15915494f            // fp32(1 / (2 pi))
#define OBJ_INV2PI_LO 4.4620826e-09f         // 1 / (2 pi) - OBJ_INV2PI_HI

__device__ __forceinline__ void pe32_project(const float* sv, const int g, const float px, const float py, const float pz,
                                             const float scale, Pe32& pe) {
  // embedding.py:47 x / scale.  A power-of-two scale (obj_scale = 2, room_0.json) makes x * (1 / scale) the SAME fp32
  // value as the IEEE division, which is ~10 instructions per component on this part: the (wave-uniform) test is
  // made once per call site.
  const unsigned sb = __float_as_uint(scale);
  if ((sb & 0x807fffffu) == 0u && sb != 0u && sb < 0x7f000000u) {
    const float inv = __uint_as_float(0x7f000000u - sb);       // 2^-e, exact
    pe.t[0] = px * inv;
    pe.t[1] = py * inv;
    pe.t[2] = pz * inv;
  } else {
    pe.t[0] = px / scale;
    pe.t[1] = py / scale;
    pe.t[2] = pz / scale;
  }
  const float* bl = sv + SV_PEB + 3 * g;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float p = fmaf(pe.t[2], bl[12 * i + 2], fmaf(pe.t[1], bl[12 * i + 1], pe.t[0] * bl[12 * i]));   // :48
    const float a0 = p * OBJ_PI_F;                 // :52 at octave 0; octave f is 2^f a0 exactly
    const float vh = a0 * OBJ_INV2PI_HI;
    pe.vh[i] = vh;
    pe.vl[i] = fmaf(a0, OBJ_INV2PI_LO, fmaf(a0, OBJ_INV2PI_HI, -vh));
  }
}
#ifndef OBJ_PE_ANCHORS
#define OBJ_PE_ANCHORS 17
#endif
constexpr unsigned PE_ANCHORS = (OBJ_PE_ANCHORS) | 1u;

// sin / cos of 2^f a, f = F0 .. F1, for the angle held as revolutions (vh + vl).  An anchor octave: exact range
// reduction (vh 2^f and the subtraction of its nearest integer are exact), then v_sin / v_cos; a derived octave:
// sin 2x = 2 sin x cos x, cos 2x = 1 - 2 sin^2 x from the octave below.  s[f], c[f] are written for f in [F0, F1]
// (c only where WANT_COS or a later doubling needs it).
template <int F0, int F1, bool WANT_COS>
__device__ __forceinline__ void pe32_octaves(const float vh, const float vl, float (&s)[6], float (&c)[6]) {
  int A = F0;                                   // the anchor at or below F0 (octave 0 always is one)
  while (!((PE_ANCHORS >> A) & 1)) --A;
  float sp = 0.f, cp = 0.f;
#pragma unroll
  for (int f = 0; f <= F1; ++f) {
    if (f < A) continue;
    const bool anchor = (f == A) || ((PE_ANCHORS >> f) & 1);
    const bool next_derived = (f < F1) && !((PE_ANCHORS >> (f + 1)) & 1);
    const bool need_cos = next_derived || (WANT_COS && f >= F0);
    float sn, cn = 0.f;
    if (anchor) {
      const float sc = (float)(1 << f);
      const float u = vh * sc;                   // exact
      const float r = u - rintf(u);              // exact, |r| <= 1/2
#ifdef OBJ_PE_NO_LOW
      const float w = r;
      (void)vl;
#else
      const float w = fmaf(vl, sc, r);
#endif
      if ((OBJ32_ABL) & 16) { sn = w; cn = w * 0.5f; }
      else {
      sn = __builtin_amdgcn_sinf(w);
      if (need_cos) cn = __builtin_amdgcn_cosf(w);
      }
    } else {
      const float t2 = sp + sp;
      sn = t2 * cp;
      if (need_cos) cn = fmaf(-t2, sp, 1.0f);
    }
    sp = sn; cp = cn;
    if (f >= F0) { s[f] = sn; c[f] = cn; }
  }
}

// x1 tile i = slot i (octaves 0..3 in the four registers), forward values
__device__ __forceinline__ f32x4 pe32_x1_tile(const Pe32& pe, const int i, const int g, const float (&s)[6]) {
  f32x4 o = {s[0], s[1], s[2], s[3]};
  if (i == 5) {                                  // only group 0 has a sixth direction
    o[0] = (g == 0) ? o[0] : ((g == 1) ? pe.t[0] : ((g == 2) ? pe.t[1] : pe.t[2]));
    o[1] = (g == 0) ? o[1] : ((g == 1) ? 1.0f : 0.0f);
    o[2] = (g == 0) ? o[2] : 0.0f;
    o[3] = (g == 0) ? o[3] : 0.0f;
  }
  return o;
}
// the two x2 entries (octaves 4, 5) of slot i
__device__ __forceinline__ void pe32_x2_pair(const int i, const int g, const float (&s)[6], float& v4, float& v5) {
  v4 = s[4]; v5 = s[5];
  if (i == 5) {
    v4 = (g == 0) ? v4 : ((g == 1) ? 1.0f : 0.0f);
    v5 = (g == 0) ? v5 : 0.0f;
  }
}
// backward of the x1 tile of slot i that also re-creates its forward values: returns the values, adds
// sum_f d_x[f] * d sin / d proj to dps  (d arg / d proj = pi 2^f, embedding.py:49-52)
__device__ __forceinline__ f32x4 pe32_x1_tile_fb(const Pe32& pe, const int i, const int g, const f32x4& dx, float& dps) {
  if ((OBJ32_ABL) & 32) { dps += dx[0]; asm volatile("" : "+v"(dps)); return f32x4{pe.vh[i], pe.vl[i], pe.vh[i], pe.vl[i]}; }
  float s[6], c[6];
  pe32_octaves<0, 3, true>(pe.vh[i], pe.vl[i], s, c);
  float v = 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f) v = fmaf(dx[f], (c[f] * OBJ_PI_F) * (float)(1 << f), v);
  if (i == 5) v = (g == 0) ? v : 0.0f;
  dps += v;
  asm volatile("" : "+v"(dps));        // consume v now (a deferred add keeps d_x and the cosines live)
  return pe32_x1_tile(pe, i, g, s);
}
// the same for the x2 pair of slot i (d_x4, d_x5: gradients of its octave-4 and octave-5 entries)
__device__ __forceinline__ void pe32_x2_pair_fb(const Pe32& pe, const int i, const int g, const float dx4, const float dx5,
                                                float& dps, float& v4, float& v5) {
  if ((OBJ32_ABL) & 32) { dps += dx4 + dx5; asm volatile("" : "+v"(dps)); v4 = pe.vh[i]; v5 = pe.vl[i]; return; }
  float s[6], c[6];
  pe32_octaves<4, 5, true>(pe.vh[i], pe.vl[i], s, c);
  float v = fmaf(dx5, (c[5] * OBJ_PI_F) * 32.0f, dx4 * ((c[4] * OBJ_PI_F) * 16.0f));
  if (i == 5) v = (g == 0) ? v : 0.0f;
  dps += v;
  asm volatile("" : "+v"(dps));
  pe32_x2_pair(i, g, s, v4, v5);
}

struct Emb32 {
  f32x4 x1[6];     // x1[i][f] = entry t = 4 i + f
  f32x4 x2[3];     // x2[T][r] = entry t = 4 T + r = 2 (slot) + (octave - 4)
};
__device__ __forceinline__ void embed32(Emb32& e, const Pe32& pe, const int g) {
  if ((OBJ32_ABL) & 32) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { e.x1[i] = f32x4{pe.vh[i], pe.vl[i], pe.vh[i], pe.vl[i]}; e.x2[i >> 1][2 * (i & 1)] = pe.vh[i]; e.x2[i >> 1][2 * (i & 1) + 1] = pe.vl[i]; }
    return;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s[6], c[6];
    pe32_octaves<0, 5, false>(pe.vh[i], pe.vl[i], s, c);
    e.x1[i] = pe32_x1_tile(pe, i, g, s);
    float v4, v5;
    pe32_x2_pair(i, g, s, v4, v5);
    e.x2[i >> 1][2 * (i & 1)] = v4;
    e.x2[i >> 1][2 * (i & 1) + 1] = v5;
  }
}
// embedding supplied by the caller in the reference's order (OccupancyMap.forward on an explicit embedding tensor)
__device__ __forceinline__ void embed32_load(Emb32& e, const float* __restrict__ emb, const int g) {
#pragma unroll
  for (int T = 0; T < 6; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = x1_col(4 * T + r, g);
      e.x1[T][r] = col >= 0 ? emb[col] : (col == BIAS_COL ? 1.0f : 0.0f);
    }
#pragma unroll
  for (int T = 0; T < 3; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = x2_col(4 * T + r, g);
      e.x2[T][r] = col >= 0 ? emb[OBJ_E1 + col] : (col == BIAS_COL ? 1.0f : 0.0f);
    }
}

// Forward chain of one 16-sample block.  wf: forward base of this lane, sv: small vectors.  Heads: every lane group
// ends with ONE of the four outputs of its sample -- group 0: 10 * raw alpha (model.py:88), groups 1..3: the colour
// channel g - 1 after the sigmoid (model.py:96) -- so only one sigmoid per lane is evaluated.
// Experiment (-DOBJ_PRIO_TOGGLE): the two waves of a SIMD alternate issue priority layer by layer.  The arbiter serves
// the older wave first, so waves 0..3 run ahead and then wait at the phase's barrier while waves 4..7 finish alone.
#ifdef OBJ_PRIO_TOGGLE
__device__ __forceinline__ void prio_layer(const int layer) {
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) & 1;
  if (layer >= 0 && ((layer ^ grp) & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
}
#else
__device__ __forceinline__ void prio_layer(const int) {}
#endif

template <bool FEAT>
__device__ __forceinline__ void mlp32_trunk(const float* wf, const float* sv, const int g, const Emb32& e, Acts& a) {
  T32 acc = zero32();
  prio_layer(1);
#pragma unroll
  for (int T = 0; T < 6; ++T) mma_f16(acc, wf, R_IN + 16 * T, e.x1[T]);
  a.h1 = relu32(acc);
  prio_layer(2);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[tt][r] = sv[SV_BM1 + 16 * tt + 4 * g + r];
  mma_f32(acc, wf, R_M1, a.h1);
  a.h2 = relu32(acc);
  prio_layer(3);
  acc = zero32();
  mma_f32(acc, wf, R_CAT, a.h2);
#pragma unroll
  for (int T = 0; T < 6; ++T) mma_f16(acc, wf, R_CAT + 32 + 16 * T, e.x1[T]);
  a.h3 = relu32(acc);
  prio_layer(4);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[tt][r] = sv[SV_BM2 + 16 * tt + 4 * g + r];
  mma_f32(acc, wf, R_M2, a.h3);
  a.h4 = relu32(acc);
  prio_layer(5);
  acc = zero32();
  mma_f32(acc, wf, R_CL, a.h4);
#pragma unroll
  for (int T = 0; T < 3; ++T) mma_f16(acc, wf, R_CL + 32 + 16 * T, e.x2[T]);
  a.hc = relu32(acc);
  if (FEAT) {
    acc = zero32();
    mma_f32(acc, wf, R_FL, a.h4);
#pragma unroll
    for (int T = 0; T < 3; ++T) mma_f16(acc, wf, R_FL + 32 + 16 * T, e.x2[T]);
    a.hf = relu32(acc);
  }
  prio_layer(-1);
}
// the four head pre-activations of the lane's sample, summed over the lane groups (every lane ends with all four)
__device__ __forceinline__ void mlp32_head_sums(const float* sv, const int g, const Acts& a, float& sa, float& s0, float& s1,
                                                float& s2) {
  float pa = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * tt + 4 * g + r;
      pa = fmaf(sv[SV_WA + row], a.h4.t[tt][r], pa);
      pc0 = fmaf(sv[SV_WOC + row], a.hc.t[tt][r], pc0);
      pc1 = fmaf(sv[SV_WOC + H + row], a.hc.t[tt][r], pc1);
      pc2 = fmaf(sv[SV_WOC + 2 * H + row], a.hc.t[tt][r], pc2);
    }
  sa = xgroup_sum(pa); s0 = xgroup_sum(pc0); s1 = xgroup_sum(pc1); s2 = xgroup_sum(pc2);
}
template <bool FEAT>
__device__ __forceinline__ float mlp32_forward(const float* wf, const float* sv, const int g, const Emb32& e, Acts& a) {
  mlp32_trunk<FEAT>(wf, sv, g, e, a);
  float sa, s0, s1, s2;
  mlp32_head_sums(sv, g, a, sa, s0, s1, s2);
  const float mine = (g == 0) ? sa : ((g == 1) ? s0 : ((g == 2) ? s1 : s2));
  const float z = mine + sv[SV_HB + g];
  return (g == 0) ? z * 10.0f : sigmoid_acc(z);
}
// the rendering form: EVERY lane group gets the density head (10 * raw alpha, model.py:88) and, groups 1..3, its own
// colour channel after the sigmoid (group 0: 0)
template <bool FEAT>
__device__ __forceinline__ void mlp32_forward_all(const float* wf, const float* sv, const int g, const Emb32& e, Acts& a,
                                                  float& alpha10, float& colour) {
  mlp32_trunk<FEAT>(wf, sv, g, e, a);
  float sa, s0, s1, s2;
  mlp32_head_sums(sv, g, a, sa, s0, s1, s2);
  alpha10 = (sa + sv[SV_HB]) * 10.0f;
  const float mine = (g == 1) ? s0 : ((g == 2) ? s1 : s2);
  colour = (g == 0) ? 0.0f : sigmoid_acc(mine + sv[SV_HB + g]);
}

}  // namespace obj32n
