// gfx950 micro-benchmark: how do f32 MFMA, bf16 MFMA, plain VALU, transcendental and LDS instructions of the waves of
// ONE SIMD share it?   hipcc -O3 --offload-arch=gfx950 tools/ubench_alu.hip -o /tmp/ubench_alu && /tmp/ubench_alu
// Every mode runs 256 workgroups (one per CU); cycles are s_memtime deltas of the loop (median over waves of a role),
// wall time from HIP events.  Results and what they mean for the fused kernel: DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum Role { R_NONE = 0, R_MFMA32, R_VALU, R_MFMA32_VALU, R_MFMABF, R_MFMABF_VALU, R_TRANS, R_LDS, R_MFMA32_LDS, R_PK,
            R_MFMA32_DEP, R_MFMA32_2ACC, R_CNDMASK, R_DPP };

#define FMA8()                                                                                             \
  asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"          \
               "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"          \
               "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"                                       \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ka), "v"(kb))
#define FMA4(a, b, c, d)                                                                                   \
  asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n"          \
               "v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(ka), "v"(kb))
#define MFMA32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define MFMABF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(abf), "v"(bbf))

__global__ __launch_bounds__(512) void k(const int* roles, int vper, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ float lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(roles[__builtin_amdgcn_readfirstlane(w)]);
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const float av = 1.0f + threadIdx.x * 1e-6f, bv = 0.5f + lane * 1e-3f, ka = 0.99991f, kb = 0.0005f;
  bf16x8 abf, bbf;
  for (int i = 0; i < 8; ++i) { abf[i] = (short)(0x3f80 + lane + i); bbf[i] = (short)(0x3e00 + 3 * lane + i); }
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-4f;
  __syncthreads();
  const float* lp = lds + (lane * 4 + w * 256) % 8000;
  if (role == R_NONE) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define LOOP for (int i = 0; i < iters; ++i)
  switch (role) {
    case R_MFMA32: LOOP { MFMA32(a0); MFMA32(a1); MFMA32(a2); MFMA32(a3); } break;
    case R_MFMA32_DEP: LOOP { MFMA32(a0); MFMA32(a0); MFMA32(a0); MFMA32(a0); } break;
    case R_MFMA32_2ACC: LOOP { MFMA32(a0); MFMA32(a1); MFMA32(a0); MFMA32(a1); } break;
    case R_VALU: LOOP { FMA8(); FMA8(); FMA8(); FMA8(); } break;                 // 32 VALU
    case R_MFMA32_VALU:                                                          // 4 MFMA, vper VALU after each
      if (vper == 4) LOOP { MFMA32(a0); FMA4(x0, x1, x2, x3); MFMA32(a1); FMA4(x4, x5, x6, x7); MFMA32(a2); FMA4(x0, x1, x2, x3); MFMA32(a3); FMA4(x4, x5, x6, x7); }
      else if (vper == 8) LOOP { MFMA32(a0); FMA8(); MFMA32(a1); FMA8(); MFMA32(a2); FMA8(); MFMA32(a3); FMA8(); }
      else LOOP { MFMA32(a0); FMA8(); FMA8(); MFMA32(a1); FMA8(); FMA8(); MFMA32(a2); FMA8(); FMA8(); MFMA32(a3); FMA8(); FMA8(); }
      break;
    case R_MFMABF: LOOP { MFMABF(a0); MFMABF(a1); MFMABF(a2); MFMABF(a3); } break;
    case R_MFMABF_VALU:
      if (vper == 2) LOOP { MFMABF(a0); FMA4(x0, x1, x2, x3); MFMABF(a1); MFMABF(a2); FMA4(x4, x5, x6, x7); MFMABF(a3); }
      else LOOP { MFMABF(a0); FMA4(x0, x1, x2, x3); MFMABF(a1); FMA4(x4, x5, x6, x7); MFMABF(a2); FMA4(x0, x1, x2, x3); MFMABF(a3); FMA4(x4, x5, x6, x7); }
      break;
    case R_TRANS:
      LOOP asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n"
                   "v_cos_f32 %4, %4\n v_cos_f32 %5, %5\n v_cos_f32 %6, %6\n v_cos_f32 %7, %7"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      break;
    case R_LDS:
      LOOP {
        f32x4 t0_, t1_, t2_, t3_;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n"
                     "ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"((unsigned)(size_t)lp));
        x0 += t0_[0] + t1_[1] + t2_[2] + t3_[3];
      }
      break;
    case R_MFMA32_LDS:
      LOOP {
        f32x4 t0_, t1_;
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024" : "=&v"(t0_), "=&v"(t1_) : "v"((unsigned)(size_t)lp));
        MFMA32(a0); MFMA32(a1); MFMA32(a2); MFMA32(a3);
        asm volatile("s_waitcnt lgkmcnt(0)");
        x0 += t0_[0] + t1_[1];
      }
      break;
    case R_PK: {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, kk = {ka, ka}, kc = {kb, kb};
      LOOP asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n"
                   "v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                   "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk), "v"(kc));
      x0 = p0[0]; x1 = p0[1]; x2 = p1[0]; x3 = p1[1]; x4 = p2[0]; x5 = p2[1]; x6 = p3[0]; x7 = p3[1];
    } break;
    case R_CNDMASK:
      LOOP asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_f32 vcc, %1, %4\n"
                   "v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc\n"
                   "v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %0, %0, %1, vcc"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(kb) : "vcc");
      break;
    case R_DPP:
      LOOP asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      break;
    default: break;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

struct Mode { const char* name; int nthr; int roles[8]; int vper; };

int main() {
  float* d; unsigned long long* dc; int* dr;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&dc, 256 * 8 * 8); hipMalloc(&dr, 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int M = R_MFMA32, V = R_VALU, N = R_NONE;
  std::vector<Mode> modes = {
    {"mfma32 x8 waves", 512, {M, M, M, M, M, M, M, M}, 0},
    {"mfma32 x4 waves (1/SIMD)", 256, {M, M, M, M, N, N, N, N}, 0},
    {"mfma32 waves0-3, waves 4-7 exit", 512, {M, M, M, M, N, N, N, N}, 0},
    {"mfma32 dependent chain x4 waves", 256, {R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, N, N, N, N}, 0},
    {"mfma32 2 accumulators x4 waves", 256, {R_MFMA32_2ACC, R_MFMA32_2ACC, R_MFMA32_2ACC, R_MFMA32_2ACC, N, N, N, N}, 0},
    {"mfma32 dependent chain x8 waves", 512, {R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP, R_MFMA32_DEP}, 0},
    {"valu x8 waves", 512, {V, V, V, V, V, V, V, V}, 0},
    {"valu x4 waves", 256, {V, V, V, V, N, N, N, N}, 0},
    {"mfma32 w0-3 | valu w4-7", 512, {M, M, M, M, V, V, V, V}, 0},
    {"mfma32 even waves | valu odd waves", 512, {M, V, M, V, M, V, M, V}, 0},
    {"mfma32+4 valu each, x4 waves", 256, {R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, N, N, N, N}, 4},
    {"mfma32+8 valu each, x4 waves", 256, {R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, N, N, N, N}, 8},
    {"mfma32+16 valu each, x4 waves", 256, {R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, N, N, N, N}, 16},
    {"mfma32+4 valu each, x8 waves", 512, {R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU}, 4},
    {"mfma32+8 valu each, x8 waves", 512, {R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU, R_MFMA32_VALU}, 8},
    {"mfmabf16 x4 waves", 256, {R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, N, N, N, N}, 0},
    {"mfmabf16 x8 waves", 512, {R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF}, 0},
    {"mfmabf16+2 valu each, x4 waves", 256, {R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, N, N, N, N}, 2},
    {"mfmabf16+4 valu each, x4 waves", 256, {R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, N, N, N, N}, 4},
    {"mfmabf16+4 valu each, x8 waves", 512, {R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU, R_MFMABF_VALU}, 4},
    {"mfmabf16 w0-3 | valu w4-7", 512, {R_MFMABF, R_MFMABF, R_MFMABF, R_MFMABF, V, V, V, V}, 0},
    {"trans x4 waves (8 per iter)", 256, {R_TRANS, R_TRANS, R_TRANS, R_TRANS, N, N, N, N}, 0},
    {"trans x8 waves (8 per iter)", 512, {R_TRANS, R_TRANS, R_TRANS, R_TRANS, R_TRANS, R_TRANS, R_TRANS, R_TRANS}, 0},
    {"mfma32 w0-3 | trans w4-7", 512, {M, M, M, M, R_TRANS, R_TRANS, R_TRANS, R_TRANS}, 0},
    {"lds b128 x4 (4 per iter) x4 waves", 256, {R_LDS, R_LDS, R_LDS, R_LDS, N, N, N, N}, 0},
    {"mfma32 w0-3 | lds w4-7", 512, {M, M, M, M, R_LDS, R_LDS, R_LDS, R_LDS}, 0},
    {"mfma32 + 2 lds b128 per 4, x4 waves", 256, {R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, N, N, N, N}, 0},
    {"mfma32 + 2 lds b128 per 4, x8 waves", 512, {R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS, R_MFMA32_LDS}, 0},
    {"pk_fma x4 waves (8 per iter)", 256, {R_PK, R_PK, R_PK, R_PK, N, N, N, N}, 0},
    {"pk_fma x8 waves (8 per iter)", 512, {R_PK, R_PK, R_PK, R_PK, R_PK, R_PK, R_PK, R_PK}, 0},
    {"cmp+cndmask x8 waves (8 instr per iter)", 512, {R_CNDMASK, R_CNDMASK, R_CNDMASK, R_CNDMASK, R_CNDMASK, R_CNDMASK, R_CNDMASK, R_CNDMASK}, 0},
    {"dpp add x8 waves (8 per iter)", 512, {R_DPP, R_DPP, R_DPP, R_DPP, R_DPP, R_DPP, R_DPP, R_DPP}, 0},
    {"dpp add x4 waves (8 per iter)", 256, {R_DPP, R_DPP, R_DPP, R_DPP, N, N, N, N}, 0},
  };
  const int iters = 20000;
  for (auto& m : modes) {
    hipMemcpy(dr, m.roles, 32, hipMemcpyHostToDevice);
    hipMemset(dc, 0, 256 * 8 * 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(m.nthr), 32768, 0, dr, m.vper, 2000, d, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(m.nthr), 32768, 0, dr, m.vper, iters, d, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), dc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    std::string s;
    for (int w = 0; w < m.nthr / 64; ++w) {
      if (m.roles[w] == R_NONE) continue;
      std::vector<unsigned long long> c;
      for (int b = 0; b < 256; ++b) c.push_back(h[b * 8 + w]);
      std::sort(c.begin(), c.end());
      char buf[64]; snprintf(buf, sizeof buf, " w%d:%.1f", w, (double)c[128] / iters);
      s += buf;
    }
    printf("%-42s %7.3f ms | cycles/iter per wave:%s\n", m.name, ms, s.c_str());
  }
  return 0;
}
