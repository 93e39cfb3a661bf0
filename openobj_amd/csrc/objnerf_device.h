// Device-side building blocks shared by the gfx950 kernels of libobjnerf_hip.so.
//
// Data layout convention used by every MLP kernel ("D layout"): a 32(feature) x 32(sample) fp32
// tile lives in one wave64 as 16 registers per lane; lane l holds sample column c = l & 31 and,
// in register r, feature row  row0(r) + 4*(l >> 5),  row0(r) = (r & 3) + 8 * (r >> 2).
// This is the C/D map of v_mfma_f32_32x32x2_f32, so a layer's output tile is directly the B
// operand of the next layer's MFMAs (k-step r consumes register r; the A operand supplies the
// weight column of the same feature row) -- activations never leave registers between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OBJ_E1 87
#define OBJ_E2 42
#define OBJ_NDIR 21
#define OBJ_EMB 129

__host__ __device__ __forceinline__ constexpr int row0(int r) { return (r & 3) + 8 * (r >> 2); }

// ----------------------------------------------------------------------------------------------
// sin / cos accurate to ~1 ulp for |x| < 1e4 (3-term Cody-Waite reduction + degree-7/8 minimax).
// The reference evaluates torch.sin on fp32(fp32(proj*2^f) * fp32(pi)) (embedding.py:52); the
// argument reaches a few hundred, so the hardware v_sin_f32 is not accurate enough for the fp32
// parity path.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_acc(float x, float& s, float& c) {
  const float n = rintf(x * 0.636619747f);
  float r = fmaf(n, -1.5707964f, x);
  r = fmaf(n, 4.371139e-08f, r);
  r = fmaf(n, 1.7763568e-15f, r);
  const float r2 = r * r;
  float p = fmaf(r2, 2.715026e-06f, -0.00019838927f);
  p = fmaf(p, r2, 0.008333328f);
  p = fmaf(p, r2, -0.16666667f);
  const float sr = fmaf(r * r2, p, r);
  float q = fmaf(r2, -2.720123e-07f, 2.4799407e-05f);
  q = fmaf(q, r2, -0.0013888883f);
  q = fmaf(q, r2, 0.041666668f);
  const float cr = fmaf(r2 * r2, q, fmaf(-0.5f, r2, 1.0f));
  const int qi = (int)n;
  const float sv = (qi & 1) ? cr : sr;
  const float cv = (qi & 1) ? sr : cr;
  s = (qi & 2) ? -sv : sv;
  c = ((qi + 1) & 2) ? -cv : cv;
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

#define OBJ_PI_F 3.14159274f

// ----------------------------------------------------------------------------------------------
// Positional-encoding index maps.
// x1 (first 87 embedding entries + a constant-1 bias row at 87): feature index e in [0,88).
// x2 (last 42 entries + constant-1 bias row at 42, zero rows after): feature index e in [0,48).
// Both return, for a compile-time feature index, what the lane must produce.
// ----------------------------------------------------------------------------------------------
struct PeSel {
  int kind;   // 0 = t[idx], 1 = sin band (dir idx, octave f), 2 = one, 3 = zero
  int idx;
  int f;
};
__host__ __device__ __forceinline__ constexpr PeSel pe_sel_x1(int e) {
  return e < 3 ? PeSel{0, e, 0}
               : (e < OBJ_E1 ? PeSel{1, (e - 3) % OBJ_NDIR, (e - 3) / OBJ_NDIR}
                             : (e == OBJ_E1 ? PeSel{2, 0, 0} : PeSel{3, 0, 0}));
}
__host__ __device__ __forceinline__ constexpr PeSel pe_sel_x2(int e) {
  return e < OBJ_E2 ? PeSel{1, (e + OBJ_E1 - 3) % OBJ_NDIR, (e + OBJ_E1 - 3) / OBJ_NDIR}
                    : (e == OBJ_E2 ? PeSel{2, 0, 0} : PeSel{3, 0, 0});
}

// Value of one embedding feature for a lane whose half (kh) picks between two compile-time
// candidates s0 (kh = 0) and s1 (kh = 1).  One sincos at most.  If WANT_COS, returns
// d(value)/d(proj[dir]) = cos(arg) * pi * 2^f instead (0 for non-band entries).
template <bool WANT_COS>
__device__ __forceinline__ float pe_lane_value(const PeSel s0, const PeSel s1, const int kh,
                                               const float (&t)[3], const float (&proj)[OBJ_NDIR]) {
  const bool band0 = s0.kind == 1, band1 = s1.kind == 1;
  float out = 0.0f;
  if (band0 || band1) {
    const float pj = kh ? proj[band1 ? s1.idx : 0] : proj[band0 ? s0.idx : 0];
    const float sc = kh ? (float)(1 << (band1 ? s1.f : 0)) : (float)(1 << (band0 ? s0.f : 0));
    const float arg = (pj * sc) * OBJ_PI_F;
    float sv, cv;
    sincos_acc(arg, sv, cv);
    out = WANT_COS ? (cv * OBJ_PI_F) * sc : sv;
    if (!band0) out = kh ? out : 0.0f;
    if (!band1) out = kh ? 0.0f : out;
  }
  if (!WANT_COS) {
    if (s0.kind == 0) out = kh ? out : t[s0.idx];
    if (s1.kind == 0) out = kh ? t[s1.idx] : out;
    if (s0.kind == 2) out = kh ? out : 1.0f;
    if (s1.kind == 2) out = kh ? 1.0f : out;
  }
  return out;
}

// ----------------------------------------------------------------------------------------------
// wave64 helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

__device__ __forceinline__ float wave_sum32(float v) {   // sum over the 32 lanes of each half
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) { return xhalf_sum(wave_sum32(v)); }

// Sum over the 32 lanes of each wave half on the VALU (DPP), no LDS traffic.  The total lands in
// lanes 16..31 of the half (rows 1 and 3 of the wave); other lanes hold partial sums.
__device__ __forceinline__ float dpp_sum32_hi(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1,3
  return v;
}
// acc (one register) keeps 16 running sums per half: slot i lives in lane 16+i of the half.
__device__ __forceinline__ void slot_accum(float& acc, const float v, const int slot, const int c) {
  const float s = dpp_sum32_hi(v);
  acc += (c == 16 + slot) ? s : 0.0f;
}

// Segmented (segment = `seg` consecutive lanes starting at multiples of seg) inclusive scans.
__device__ __forceinline__ float seg_scan_mul(float v, int pos, int seg) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (d < seg && pos >= d) v *= o;
  }
  return v;
}
__device__ __forceinline__ float seg_scan_add(float v, int pos, int seg) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (d < seg && pos >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float seg_rscan_add(float v, int pos, int seg) {   // suffix inclusive
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(v, d, 64);
    if (d < seg && pos + d < seg) v += o;
  }
  return v;
}
