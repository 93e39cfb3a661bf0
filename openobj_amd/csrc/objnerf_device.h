// Device-side building blocks shared by the gfx950 kernels of libobjnerf_hip.so.
//
// (The MLP-specific layouts live in objnerf_mlp.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OBJ_E1 87
#define OBJ_E2 42
#define OBJ_NDIR 21
#define OBJ_EMB 129

// ----------------------------------------------------------------------------------------------
// sin / cos accurate to ~1.3 ulp (1.2e-7 / 1.5e-7 absolute) for |x| < 1e4: 3-term Cody-Waite reduction
// by pi to r in [-pi/2, pi/2], minimax polynomials, sign = (-1)^n.
// The reference evaluates torch.sin on fp32(fp32(proj*2^f) * fp32(pi)) (embedding.py:52); the argument
// reaches a few hundred, so the hardware v_sin_f32 is not accurate enough for the fp32 parity path.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float pi_reduce(float x, unsigned& sign) {
  const float n = rintf(x * 0.31830987f);
  float r = fmaf(n, -3.1415927f, x);
  r = fmaf(n, 8.742278e-08f, r);
  r = fmaf(n, 3.5527137e-15f, r);
  sign = ((unsigned)(int)n) << 31;
  return r;
}
__device__ __forceinline__ float sin_poly(float r, float r2) {
  float p = fmaf(r2, 2.5977522e-06f, -0.00019805509f);
  p = fmaf(p, r2, 0.008333f);
  p = fmaf(p, r2, -0.16666657f);
  return fmaf(r * r2, p, r);
}
__device__ __forceinline__ float cos_poly(float r2) {
  float q = fmaf(r2, -2.6058984e-07f, 2.4760864e-05f);
  q = fmaf(q, r2, -0.0013888384f);
  q = fmaf(q, r2, 0.041666638f);
  q = fmaf(q, r2, -0.5f);
  return fmaf(r2, q, 1.0f);
}
__device__ __forceinline__ float sin_acc(float x) {
#ifdef ABL_CHEAP_PE
  return x * 0.001f;
#endif
  unsigned sg;
  const float r = pi_reduce(x, sg);
  return __uint_as_float(__float_as_uint(sin_poly(r, r * r)) ^ sg);
}
__device__ __forceinline__ void sincos_acc(float x, float& s, float& c) {
#ifdef ABL_CHEAP_PE
  s = x * 0.001f; c = 1.0f - x * 0.002f; return;
#endif
  unsigned sg;
  const float r = pi_reduce(x, sg);
  const float r2 = r * r;
  s = __uint_as_float(__float_as_uint(sin_poly(r, r2)) ^ sg);
  c = __uint_as_float(__float_as_uint(cos_poly(r2)) ^ sg);
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

#define OBJ_PI_F 3.14159274f

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

// Sum over the 16 lanes of each DPP row (= one lane group g) on the VALU, no LDS traffic; every lane of
// the row ends with the row total.
__device__ __forceinline__ float dpp_rowsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}
// acc (one register) keeps 16 running row sums per lane group: slot i lives in lane i of the row.
__device__ __forceinline__ void slot_accum16(float& acc, const float v, const int slot, const int c) {
  const float s = dpp_rowsum16(v);
  acc += (c == slot) ? s : 0.0f;
  asm volatile("" : "+v"(acc));   // consume the row sum NOW (deferred adds keep dozens of partial sums live)
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
