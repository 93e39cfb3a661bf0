// gfx950 micro-benchmark behind the structure of fwd256_kernel: v_mfma_f32_32x32x16_bf16 on ONE wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma32.hip -o /tmp/ubench_mfma32 && /tmp/ubench_mfma32
// modes: dependent chain on one accumulator / two / four independent chains; a chain with V independent VALU
// instructions, with S scalar instructions, or with an LDS read + counted wait behind every MFMA.
// Prints shader-clock cycles per MFMA (s_memtime) and the wall-clock rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define V1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(ka), "v"(kb))
#define S1(x) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x) : : "scc")

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc, const char* wsrc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.3f + 0.01f * lane + 0.02f * i); b[i] = (__bf16)(-0.2f + 0.007f * (lane ^ 9) - 0.01f * i); }
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
  float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
  const float ka = 0.9999f, kb = 1e-4f;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  const uint32_t la = (uint32_t)(uintptr_t)lds + lane * 16;
  bf16x8 r0 = a, r1 = a, r2 = a, r3 = a;
  // (modes 10-13) a 1 KB piece per wave and transfer, from a 4 MB L2-resident table, like the weight ring of fwd256_kernel
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long sa0 = (unsigned long long)(wsrc + ((blockIdx.x * 4 + wv) & 1023) * 4096);
  const unsigned long long sbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa0 >> 32)) << 32) |
                                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa0);
  const uint32_t dbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds + 32768 + wv * 4096);
  const uint32_t vo = lane * 16;
  typedef __attribute__((address_space(1))) const bf16x8 GV;
  bf16x8 g0 = a;
#define RDW(reg, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(reg) : "v"(la))
#define WT2(x, y, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(x), "+v"(y))
#define GLDS(off) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:" #off :: "s"(dbase), "v"(vo), "s"(sbase) : "memory", "m0")
#define QUAD() do { WT2(r0, r1, 2); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, b, c0, 0, 0, 0); RDW(r0, 0); \
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, b, c1, 0, 0, 0); RDW(r1, 1024); \
                    WT2(r2, r3, 2); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r2, b, c0, 0, 0, 0); RDW(r2, 2048); \
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r3, b, c1, 0, 0, 0); RDW(r3, 3072); } while (0)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { MF(c0); MF(c0); MF(c0); MF(c0); }
    if (MODE == 1) { MF(c0); MF(c1); MF(c0); MF(c1); }
    if (MODE == 2) { MF(c0); MF(c1); MF(c2); MF(c3); }
    if (MODE == 3) { MF(c0); V1(x0); V1(x1); MF(c0); V1(x2); V1(x3); MF(c0); V1(x4); V1(x5); MF(c0); V1(x6); V1(x7); }                 // 2 VALU / MFMA
    if (MODE == 4) { MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); MF(c0); V1(x4); V1(x5); V1(x6); V1(x7); MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); MF(c0); V1(x4); V1(x5); V1(x6); V1(x7); }   // 4
    if (MODE == 5) { MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7);
                     MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); }                    // 8
    if (MODE == 6) { MF(c0); S1(s0); S1(s1); S1(s2); S1(s3); MF(c0); S1(s0); S1(s1); S1(s2); S1(s3); MF(c0); S1(s0); S1(s1); S1(s2); S1(s3); MF(c0); S1(s0); S1(s1); S1(s2); S1(s3); }   // 4 SALU
    if (MODE == 7) {    // LDS read two ahead + counted wait, like block_mma
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r0)); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, b, c0, 0, 0, 0);
      asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(la));
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r1)); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, b, c0, 0, 0, 0);
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(r1) : "v"(la));
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r0)); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, b, c0, 0, 0, 0);
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(r0) : "v"(la));
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r1)); c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, b, c0, 0, 0, 0);
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(r1) : "v"(la));
    }
    if (MODE == 8) { MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); MF(c1); V1(x4); V1(x5); V1(x6); V1(x7); MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); MF(c1); V1(x4); V1(x5); V1(x6); V1(x7); }   // 2 chains + 4 VALU
    if (MODE == 9) { MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); MF(c1); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7);
                     MF(c0); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); MF(c1); V1(x0); V1(x1); V1(x2); V1(x3); V1(x4); V1(x5); V1(x6); V1(x7); }                    // 2 chains + 8 VALU
    if (MODE == 10) { QUAD(); }                                            // two chains, one ds_read_b128 per MFMA, a wait per pair
    if (MODE == 11) { QUAD(); GLDS(0); }                                   // + one LDS-DMA piece per 4 MFMAs
    if (MODE == 12) { QUAD(); GLDS(0); QUAD(); }                           // + one per 8 MFMAs (counted as 4 MFMAs per iteration: see main)
    if (MODE == 13) {                                                      // + one register-staged piece per 4 MFMAs
      QUAD();
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0));
      asm volatile("ds_write_b128 %0, %1 offset:32768" :: "v"(la), "v"(g0) : "memory");
      g0 = *(GV*)(wsrc + ((blockIdx.x * 4 + wv + i) & 1023) * 4096 + vo);
    }
    if ((i & 255) == 255) { c0 *= 1e-3f; c1 *= 1e-3f; c2 *= 1e-3f; c3 *= 1e-3f; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3) + (float)r0[0] + (float)r1[0];
  for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static char* wsrc;
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc, double mfma_per_iter = 4.0) {
  const int iters = 20000, nwg = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 100 * 1024, 0, iters, out, cyc, (const char*)wsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < nwg; ++i) s += (double)h[i];
  const double per = s / nwg / (mfma_per_iter * iters);
  const double tf = 32768.0 * mfma_per_iter * iters * 4 * nwg / (ms * 1e-3) / 1e12;
  printf("%-44s %7.1f ticks / MFMA   %8.3f ms   %7.1f TFLOP/s   %6.2f ns / MFMA\n", name, per, ms, tf, ms * 1e6 / (mfma_per_iter * iters));
  fflush(stdout);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  hipMalloc(&wsrc, 4 << 20); hipMemset(wsrc, 0x11, 4 << 20);
  run<0>("one dependent chain", out, cyc);
  run<1>("two chains", out, cyc);
  run<2>("four chains", out, cyc);
  run<3>("one chain + 2 VALU per MFMA", out, cyc);
  run<4>("one chain + 4 VALU per MFMA", out, cyc);
  run<5>("one chain + 8 VALU per MFMA", out, cyc);
  run<6>("one chain + 4 SALU per MFMA", out, cyc);
  run<7>("one chain + ds_read_b128 + counted wait", out, cyc);
  run<8>("two chains + 4 VALU per MFMA", out, cyc);
  run<9>("two chains + 8 VALU per MFMA", out, cyc);
  run<10>("two chains + ds_read_b128 per MFMA (paired waits)", out, cyc);
  run<11>("  + one LDS-DMA piece per 4 MFMAs", out, cyc);
  run<12>("  + one LDS-DMA piece per 8 MFMAs", out, cyc, 8.0);
  run<13>("  + one register-staged piece per 4 MFMAs", out, cyc);
  return 0;
}
