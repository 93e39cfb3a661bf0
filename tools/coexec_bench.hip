// Microbenchmark behind DESIGN.md 6 ("MFMA and VALU time add up"):  hipcc -O3 --offload-arch=gfx950 tools/coexec_bench.hip
// Do MFMA and VALU work of the waves of one SIMD overlap?  mode 0: MFMA only, 1: VALU only, 2: both interleaved in
// every wave, 3: SIMDs 0,2 MFMA-only / SIMDs 1,3 VALU-only, 4: every SIMD one MFMA-only and one VALU-only wave.
// Measured on MI355X (256 workgroups x 8 waves, 20000 iterations): 2.47 / 1.52 / 3.90 / 2.35 / 2.49 ms:
//  * MFMA + VALU issued by the SAME waves nearly add up (3.9 vs 4.0 ms);
//  * one MFMA-only wave per SIMD reaches only HALF the MFMA rate (mode 4 takes as long as mode 0 with half the
//    MFMAs), but a VALU-only wave beside it is free.
// So two waves per SIMD running the same phase (what the fused kernel does) is the right shape, and its tile time is
// ~ MFMA time + VALU time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int w = threadIdx.x >> 6;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const float av = 1.0f + threadIdx.x * 1e-6f, bv = 0.5f;
  const bool do_m = mode == 0 || mode == 2 || (mode == 3 && (w & 1) == 0) || (mode == 4 && w < 4);
  const bool do_v = mode == 1 || mode == 2 || (mode == 3 && (w & 1) == 1) || (mode == 4 && w >= 4);
  for (int i = 0; i < iters; ++i) {
    if (do_m) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a3, 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {       // 32 independent-ish fmas = the issue time of 4 MFMAs (4 x 32 cycles)
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
        x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 1000, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 20000, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms\n", mode, ms);
  }
  return 0;
}
