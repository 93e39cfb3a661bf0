// Counter-based random numbers for the samplers (Philox4x32-10, Salmon et al. 2011): a draw is a pure function of
// (seed, stream, index), so every kernel that needs "the u of ray i, bin s" computes it in place and nothing random is
// ever written to HBM.  The reference draws with torch.rand / Tensor.normal_ (vmap.py:414-415,498-542, utils.py:371,391),
// whose stream cannot be replayed; parity for drawn samples is therefore distributional, exact parity uses injected draws.
#pragma once
#include <stdint.h>

namespace objrng {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * ctr.x, p1 = (uint64_t)M1 * ctr.z;
    const U4 n = {(uint32_t)(p1 >> 32) ^ ctr.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ ctr.w ^ k1, (uint32_t)p0};
    ctr = n;
    k0 += W0; k1 += W1;
  }
  return ctr;
}
// uniform in [0, 1) with 24 random bits, like torch.rand on float32
__host__ __device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// four uniforms of counter (stream, a, b, c) under `seed`
__host__ __device__ inline void uniform4(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint32_t c, float (&u)[4]) {
  const U4 r = philox4x32_10(U4{a, b, c, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[0] = u01(r.x); u[1] = u01(r.y); u[2] = u01(r.z); u[3] = u01(r.w);
}
// one uniform: element (c & 3) of the block c >> 2
__host__ __device__ inline float uniform1(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint32_t c) {
  float u[4];
  uniform4(seed, stream, a, b, c >> 2, u);
  return u[c & 3];
}
// one standard normal (Box-Muller on two of the four uniforms of block c >> 1)
__device__ inline float normal1(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint32_t c) {
  float u[4];
  uniform4(seed, stream, a, b, c >> 1, u);
  const float u1 = 1.0f - u[2 * (c & 1)];             // (0, 1]
  const float u2 = u[2 * (c & 1) + 1];
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

enum Stream : uint32_t { S_KEYFRAME = 1, S_PIXEL_W = 2, S_PIXEL_H = 3, S_BINS_U = 4, S_BINS_G = 5, S_HELPER_U = 6, S_HELPER_G = 7, S_BOX_U = 0 };

}  // namespace objrng
