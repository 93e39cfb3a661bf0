// Diagnostic: semantics of ds_read_b64_tr_b16 on gfx950 as the 16-bit GEMM uses it (k-major LDS tile -> MFMA operand).
// Expectation (cdna_hip_programming.md T10): in each 16-lane group, lane 4q+p supplies the address of row q, columns
// 4p..4p+3 of a 4 x 16 block; lane i receives column i of the four rows (row q in element q).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 136;
__global__ void k(const float* in, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 S[32 * PITCH];
  for (int i = threadIdx.x; i < 32 * PITCH; i += 64) S[i] = (__bf16)in[i];
  __syncthreads();
  const int lane = threadIdx.x, c = lane & 15, gg = lane >> 4;
  const __bf16* p = S + (8 * gg + (c >> 2)) * PITCH + 16 + 4 * (c & 3);     // block rows 8 gg .. +3, columns 16 .. 31
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)(uintptr_t)p) : "memory");
  bf16x4 v = __builtin_bit_cast(bf16x4, r);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
  std::vector<float> h(32 * PITCH);
  for (int r = 0; r < 32; ++r) for (int c = 0; c < PITCH; ++c) h[r * PITCH + c] = (float)(r * 256 + c);   // exact in bf16? no: use small ints
  for (int r = 0; r < 32; ++r) for (int c = 0; c < PITCH; ++c) h[r * PITCH + c] = (float)((r << 3) ^ 0) + (float)(c % 128) * 0.0f + (float)(r * 2 + (c & 1)) ;
  // value encodes (row, column) exactly representable in bf16: row in 0..31, col in 0..135 -> use two planes
  float *din, *dout;
  std::vector<float> o(256);
  int bad = 0;
  for (int plane = 0; plane < 2; ++plane) {
    for (int r = 0; r < 32; ++r) for (int c = 0; c < PITCH; ++c) h[r * PITCH + c] = plane ? (float)c : (float)r;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
      const int c = lane & 15, gg = lane >> 4;
      const float want = plane ? (float)(16 + c) : (float)(8 * gg + e);
      if (o[lane * 4 + e] != want) { if (bad < 8) printf("plane %d lane %d e %d: got %g want %g\n", plane, lane, e, o[lane * 4 + e], want); ++bad; }
    }
  }
  printf(bad ? "MISMATCH %d\n" : "tr_b16 semantics as expected\n", bad);
  return bad != 0;
}
