// Device-side building blocks shared by the gfx950 kernels of libobjnerf_hip.so.
//
// (The MLP-specific layouts live in objnerf_mlp.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
// Kernel attributes (the opt-in to more than 64 KB of dynamic LDS) belong to a device: set them once per device of
// this process, thread-safely.  (Every lambda has its own type, hence every call site its own set of flags.)
template <class F>
inline void objnerf_once_per_device(F&& fn) {
  static std::once_flag flags[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(flags[dev & 63], fn);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OBJ_E1 87
#define OBJ_E2 42
#define OBJ_NDIR 21
#define OBJ_EMB 129

// ----------------------------------------------------------------------------------------------
// sin / cos accurate to ~1.4e-7 / 1.6e-7 absolute for |x| < 1e4: 3-term Cody-Waite reduction by pi to
// r in [-pi/2, pi/2], then v_sin_f32 / v_cos_f32 on r / 2 pi (measured on gfx950 over 2^20 points of that interval:
// max abs error 1.35e-7 / 1.61e-7, the same class as degree-9 / degree-10 minimax polynomials, which remain
// available under OBJ_POLY_SINCOS), sign = (-1)^n.
// The reference evaluates torch.sin on fp32(fp32(proj*2^f) * fp32(pi)) (embedding.py:52); the argument reaches a
// few hundred, so v_sin_f32 on the UNREDUCED argument is not accurate enough for the fp32 parity path.
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
// OBJ_HW_SINCOS (defined by objnerf_train_bf16.hip only): v_sin_f32 / v_cos_f32 on x / 2 pi.  Absolute error
// ~1e-6 at these arguments -- far below the bf16 rounding (4e-3) the embedding gets in that mode, and 3 instructions
// instead of ~28; never used on the fp32 parity path.
__device__ __forceinline__ float sin_acc(float x) {
#ifdef OBJ_HW_SINCOS
  return __builtin_amdgcn_sinf(x * 0.15915494f);
#endif
  unsigned sg;
  const float r = pi_reduce(x, sg);
#ifdef OBJ_POLY_SINCOS
  return __uint_as_float(__float_as_uint(sin_poly(r, r * r)) ^ sg);
#else
  return __uint_as_float(__float_as_uint(__builtin_amdgcn_sinf(r * 0.15915494f)) ^ sg);
#endif
}
__device__ __forceinline__ void sincos_acc(float x, float& s, float& c) {
#ifdef OBJ_HW_SINCOS
  const float rev = x * 0.15915494f;
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
  return;
#endif
  unsigned sg;
  const float r = pi_reduce(x, sg);
#ifdef OBJ_POLY_SINCOS
  const float r2 = r * r;
  s = __uint_as_float(__float_as_uint(sin_poly(r, r2)) ^ sg);
  c = __uint_as_float(__float_as_uint(cos_poly(r2)) ^ sg);
#else
  const float rr = r * 0.15915494f;
  s = __uint_as_float(__float_as_uint(__builtin_amdgcn_sinf(rr)) ^ sg);
  c = __uint_as_float(__float_as_uint(__builtin_amdgcn_cosf(rr)) ^ sg);
#endif
}

#define OBJ_PI_F 3.14159274f

// Embedding band and its derivative for a projection p and a compile-time octave scale sc:
//   band = sin(fp32(fp32(p * sc) * pi))  (the reference's roundings, embedding.py:49-52),  dband/dp = cos(.) * pi * sc.
// In the hardware mode the argument in revolutions is p * (sc / 2) directly and pi * sc is one constant.
__device__ __forceinline__ float band_sin(const float p, const float sc) {
#ifdef OBJ_HW_SINCOS
  return __builtin_amdgcn_sinf(p * (0.5f * sc));
#else
  return sin_acc((p * sc) * OBJ_PI_F);
#endif
}
__device__ __forceinline__ void band_sincos(const float p, const float sc, float& s, float& dcos) {
#ifdef OBJ_HW_SINCOS
  const float rev = p * (0.5f * sc);
  s = __builtin_amdgcn_sinf(rev);
  dcos = __builtin_amdgcn_cosf(rev) * (OBJ_PI_F * sc);
#else
  float cv;
  sincos_acc((p * sc) * OBJ_PI_F, s, cv);
  dcos = (cv * OBJ_PI_F) * sc;
#endif
}

__device__ __forceinline__ float sigmoid_acc(float x) {
#ifdef OBJ_HW_SINCOS      // bf16 mode only: v_exp_f32 + v_rcp_f32 (~1e-6 relative) instead of expf and an IEEE division
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * x));
#else
  return 1.0f / (1.0f + expf(-x));
#endif
}


// utils.stratified_bins (utils.py:342-379): torch.linspace(0, 1, n + 1)[i] in fp32 (the GPU formula: start + step i
// below the middle, end - step (n - i) above) and bin i of [lo, hi] with the draw u
__device__ __forceinline__ float lin01(int i, int n) {
  const float step = 1.0f / (float)n;
  return (i < (n + 1) / 2) ? step * (float)i : 1.0f - step * (float)(n - i);
}
__device__ __forceinline__ float strat(float lo, float hi, int i, int n, float u) {
  const float rng = hi - lo;
  return (rng * lin01(i, n) + lo) + u * (rng / (float)n);
}

// ----------------------------------------------------------------------------------------------
// wave64 helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }


// Sum over the 16 lanes of each DPP row (= one lane group g) on the VALU, no LDS traffic; every lane of
// the row ends with the row total.
__device__ __forceinline__ float dpp_rowsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}
// sum over the 32 lanes of each half (rows 0,1 / rows 2,3), on every lane: DPP row sums, then one
// v_permlane16_swap pairs the rows -- no LDS permutes
__device__ __forceinline__ float wave_sum32(float v) {
  const unsigned u = __float_as_uint(dpp_rowsum16(v));
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // (r0,r0,r2,r2), (r1,r1,r3,r3)
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float wave_sum64(float v) {
  const unsigned u = __float_as_uint(wave_sum32(v));
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// acc (one register) keeps 16 running row sums per lane group: slot i lives in lane i of the row.
__device__ __forceinline__ void slot_accum16(float& acc, const float v, const int slot, const int c) {
  const float s = dpp_rowsum16(v);
  acc += (c == slot) ? s : 0.0f;
  asm volatile("" : "+v"(acc));   // consume the row sum NOW (deferred adds keep dozens of partial sums live)
}

// The same for 16 (or 8) values at once, as a transposing butterfly: every step halves the number of live values
// while doubling the lanes each one has been summed over, so 16 row sums cost ~57 VALU ops instead of 16 x 7.
//   slot_sums16(A, B, c): lane c < 8 returns sum_row A[c], lane c >= 8 returns sum_row B[c - 8]
//   slot_sums8(A, c):     every lane returns sum_row A[c & 7]
template <int CTRL>
__device__ __forceinline__ float dpp_get(const float v) {   // v of the lane selected by the DPP control
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float slot_fold4(float (&v)[8], const int c) {
  // in: v[k] already summed over {c, c ^ 8}.  lane c ^ 4 is c + 4 (bit 2 clear) or c - 4 (set): two row shifts
  const bool b2 = (c & 4) != 0, b1 = (c & 2) != 0, b0 = (c & 1) != 0;
  float w4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = b2 ? v[k + 4] : v[k], send = b2 ? v[k] : v[k + 4];
    const float up = dpp_get<0x104>(send), dn = dpp_get<0x114>(send);     // row_shl:4 reads lane c + 4, row_shr:4 lane c - 4
    w4[k] = keep + (b2 ? dn : up);
  }
  float w2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = b1 ? w4[k + 2] : w4[k], send = b1 ? w4[k] : w4[k + 2];
    w2[k] = keep + dpp_get<0x4E>(send);                                   // quad_perm [2,3,0,1]: lane c ^ 2
  }
  const float keep = b0 ? w2[1] : w2[0], send = b0 ? w2[0] : w2[1];
  return keep + dpp_get<0xB1>(send);                                      // quad_perm [1,0,3,2]: lane c ^ 1
}
__device__ __forceinline__ float slot_sums16(const float (&A)[8], const float (&B)[8], const int c) {
  const bool b3 = (c & 8) != 0;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float keep = b3 ? B[k] : A[k], send = b3 ? A[k] : B[k];
    v[k] = keep + dpp_get<0x128>(send);                                   // row_ror:8: lane c ^ 8
  }
  return slot_fold4(v, c);
}
__device__ __forceinline__ float slot_sums8(const float (&A)[8], const int c) {
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = A[k] + dpp_get<0x128>(A[k]);
  return slot_fold4(v, c);
}

// Segmented scans over the 64 lanes of a wave; a segment = `seg` consecutive lanes starting at a multiple of seg
// (one ray of seg samples).  Two interchangeable policies:
//   SegRows    seg in {16, 32, 64}: row_shr / row_shl DPP steps inside each 16-lane row, rows linked by three
//              v_readlane broadcasts and per-lane selects fixed at kernel start -- no LDS permutes, no branches.
//   SegGeneric any seg (the reference's native 10 / 14 samples per ray): ds_bpermute shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(const float old, const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                              CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_bcast(const float v, const int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ bool seg_is_rows(const int seg) { return seg == 64 || seg == 32 || seg == 16; }

struct SegRows {
  bool u0, u1, u2;        // forward: this lane's row continues a segment that includes row 0 / 1 / 2's end
  bool r1, r2, r3;        // reverse: ... that includes the start of row 1 / 2 / 3
  bool m0, m1, m2, m3;    // row i belongs to this lane's segment
  __device__ __forceinline__ static SegRows make(const int seg, const int lane) {
    const int row = lane >> 4, sid = (16 * row) / seg;
    SegRows s;
    s.m0 = sid == 0; s.m1 = (16 / seg) == sid; s.m2 = (32 / seg) == sid; s.m3 = (48 / seg) == sid;
    s.u0 = s.m0 && row > 0; s.u1 = s.m1 && row > 1; s.u2 = s.m2 && row > 2;
    s.r1 = s.m1 && row < 1; s.r2 = s.m2 && row < 2; s.r3 = s.m3 && row < 3;
    return s;
  }
  __device__ __forceinline__ float scan_mul(float v, int) const {
    v *= dpp_mov<0x111>(1.0f, v); v *= dpp_mov<0x112>(1.0f, v); v *= dpp_mov<0x114>(1.0f, v); v *= dpp_mov<0x118>(1.0f, v);
    const float t0 = lane_bcast(v, 15), t1 = lane_bcast(v, 31), t2 = lane_bcast(v, 47);
    return v * (((u0 ? t0 : 1.0f) * (u1 ? t1 : 1.0f)) * (u2 ? t2 : 1.0f));
  }
  __device__ __forceinline__ float scan_add(float v, int) const {
    v += dpp_mov<0x111>(0.0f, v); v += dpp_mov<0x112>(0.0f, v); v += dpp_mov<0x114>(0.0f, v); v += dpp_mov<0x118>(0.0f, v);
    const float t0 = lane_bcast(v, 15), t1 = lane_bcast(v, 31), t2 = lane_bcast(v, 47);
    return v + (((u0 ? t0 : 0.0f) + (u1 ? t1 : 0.0f)) + (u2 ? t2 : 0.0f));
  }
  __device__ __forceinline__ float total_add(const float v, int) const {     // segment sum on every lane
    const float rs = dpp_rowsum16(v);
    const float q0 = lane_bcast(rs, 0), q1 = lane_bcast(rs, 16), q2 = lane_bcast(rs, 32), q3 = lane_bcast(rs, 48);
    return ((m0 ? q0 : 0.0f) + (m1 ? q1 : 0.0f)) + ((m2 ? q2 : 0.0f) + (m3 ? q3 : 0.0f));
  }
  __device__ __forceinline__ float rscan_add(float v, int) const {            // suffix inclusive
    v += dpp_mov<0x101>(0.0f, v); v += dpp_mov<0x102>(0.0f, v); v += dpp_mov<0x104>(0.0f, v); v += dpp_mov<0x108>(0.0f, v);
    const float t1 = lane_bcast(v, 16), t2 = lane_bcast(v, 32), t3 = lane_bcast(v, 48);
    return v + ((r1 ? t1 : 0.0f) + ((r2 ? t2 : 0.0f) + (r3 ? t3 : 0.0f)));
  }
};

struct SegGeneric {
  int seg;
  __device__ __forceinline__ float scan_mul(float v, const int pos) const {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(v, d, 64);
      if (d < seg && pos >= d) v *= o;
    }
    return v;
  }
  __device__ __forceinline__ float scan_add(float v, const int pos) const {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(v, d, 64);
      if (d < seg && pos >= d) v += o;
    }
    return v;
  }
  __device__ __forceinline__ float total_add(const float v, const int pos) const {
    const int last = (int)(threadIdx.x & 63) - pos + seg - 1;
    return __shfl(scan_add(v, pos), last, 64);
  }
  __device__ __forceinline__ float rscan_add(float v, const int pos) const {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_down(v, d, 64);
      if (d < seg && pos + d < seg) v += o;
    }
    return v;
  }
};
