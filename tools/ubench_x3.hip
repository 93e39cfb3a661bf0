// Diagnostic: the fused kernel's layer chain (32 -> 32, ReLU, activations register-resident in the MFMA D layout, 16
// samples per wave and tile, weights in LDS) in three arithmetics:
//   f32  v_mfma_f32_16x16x4_f32, 16 MFMAs per layer and tile (what objnerf_train32.hip runs today);
//   x3   every fp32 operand split into three bf16 pieces (hi + mid + lo = the value to 2^-24), six of the nine partial
//        products on v_mfma_f32_16x16x32_bf16 (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid): 12 MFMAs of half the
//        length per layer and tile, fp32 accumulation -- the weights are split once into three LDS images, the
//        activations in registers after every layer (11 VALU instructions per pair of values);
//   x1   plain bf16 operands (2 MFMAs per layer and tile): the floor of the bf16 pipe, not an fp32 substitute.
// EV extra dependent-free VALU instructions per layer and lane stand in for the positional encoding / compositing work
// of the real kernel (fp32 MFMAs do not co-execute with VALU on this part, bf16 MFMAs do: DESIGN.md section 4.1).
// Prints ns per layer-tile and wave, the speed-up, and the error of each variant against an fp64 evaluation.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_x3.hip -o /tmp/ubench_x3 && /tmp/ubench_x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int H = 32, L = 4, NW = 8;      // hidden, layers of the chain, waves per workgroup (2 per SIMD)

// feature held by lane group g in register (t, r) of the D layout: rows 16 t + 4 g + r
__host__ __device__ inline int feat_of(int g, int t, int r) { return 16 * t + 4 * g + r; }

struct Args {
  const float* w;        // [L][H out][H in]
  const float* x0;       // [tiles][H][16]
  float* out;            // [tiles][H][16]  (first `keep` tiles of workgroup 0 / wave 0)
  int tiles, keep, ev;
};

__device__ __forceinline__ float extra_valu(float a, int ev) {      // ev instructions, 8 independent chains
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  for (int i = 0; i < ev; i += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0000001f, 1e-9f);
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// ---- f32 --------------------------------------------------------------------------------------------------------
// LDS image: [L][u][half][lane][4]: lane (c, g) reads W[16 u + c][feat_of(g, half, 0..3)] with one ds_read_b128
__global__ __launch_bounds__(64 * NW) void chain_f32(const Args a) {
  __shared__ __attribute__((aligned(16))) float Wl[L * 2 * 2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  for (int i = tid; i < L * 2 * 2 * 64 * 4; i += 64 * NW) {
    const int r = i & 3, ln = (i >> 2) & 63, hf = (i >> 8) & 1, u = (i >> 9) & 1, l = i >> 10;
    Wl[i] = a.w[(l * H + 16 * u + (ln & 15)) * H + feat_of(ln >> 4, hf, r)];
  }
  __syncthreads();
  float sink = 0.f;
  for (int t = 0; t < a.tiles; ++t) {
    const long tile = ((long)blockIdx.x * NW + w) * a.tiles + t;
    float x[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[tt][r] = a.x0[(tile % a.keep) * H * 16 + feat_of(g, tt, r) * 16 + c];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wl[(((l * 2 + u) * 2 + hf) * 64 + lane) * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], x[hf][r], acc[u], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[u][r] = fmaxf(acc[u][r], 0.f) + 0.01f;
      sink += extra_valu(x[0][0], a.ev);
    }
    if (blockIdx.x == 0 && w == 0 && t < a.keep)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[t * H * 16 + feat_of(g, tt, r) * 16 + c] = x[tt][r];
  }
  if (sink == 123.456f) a.out[0] = sink;
}

// ---- bf16 pieces ------------------------------------------------------------------------------------------------
// split 8 fp32 values into NP bf16 pieces (round to nearest each time; the remainders are exact)
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[2][4], bf16x8 (&p)[NP]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = x[e >> 2][e & 3];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 b = (__bf16)r[e];
      p[q][e] = b;
      if (q + 1 < NP) r[e] -= (float)b;
    }
}
// LDS images: [L][u][piece][lane][8 bf16]: lane (c, g) reads the pieces of W[16 u + c][feat_of(g, e >> 2, e & 3)], e = 0..7
template <int NP>
__global__ __launch_bounds__(64 * NW) void chain_bf(const Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 Wl[L * 2 * NP * 64 * 8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  for (int i = tid; i < L * 2 * 64 * 8; i += 64 * NW) {
    const int e = i & 7, ln = (i >> 3) & 63, u = (i >> 9) & 1, l = i >> 10;
    float r = a.w[(l * H + 16 * u + (ln & 15)) * H + feat_of(ln >> 4, e >> 2, e & 3)];
    for (int q = 0; q < NP; ++q) {
      const __bf16 b = (__bf16)r;
      Wl[((((l * 2 + u) * NP + q) * 64) + ln) * 8 + e] = b;
      r -= (float)b;
    }
  }
  __syncthreads();
  float sink = 0.f;
  for (int t = 0; t < a.tiles; ++t) {
    const long tile = ((long)blockIdx.x * NW + w) * a.tiles + t;
    float x[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[tt][r] = a.x0[(tile % a.keep) * H * 16 + feat_of(g, tt, r) * 16 + c];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      bf16x8 xp[NP];
      split8<NP>(x, xp);
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      bf16x8 wp[2][NP];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < NP; ++q) wp[u][q] = *reinterpret_cast<const bf16x8*>(&Wl[((((l * 2 + u) * NP + q) * 64) + lane) * 8]);
      // (the two accumulators alternate: no MFMA waits for the one just issued; smallest products first)
      constexpr int NPROD = NP == 3 ? 6 : 1;
      constexpr int PW[6] = {1, 2, 0, 1, 0, 0}, PX[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int i = 6 - NPROD; i < 6; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[u][NP == 3 ? PW[i] : 0], xp[NP == 3 ? PX[i] : 0], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[u][r] = fmaxf(acc[u][r], 0.f) + 0.01f;
      sink += extra_valu(x[0][0], a.ev);
    }
    if (blockIdx.x == 0 && w == 0 && t < a.keep)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[t * H * 16 + feat_of(g, tt, r) * 16 + c] = x[tt][r];
  }
  if (sink == 123.456f) a.out[0] = sink;
}

int main(int argc, char** argv) {
  const int keep = 8, tiles = 512, blocks = 256 * 2;
  std::vector<float> w(L * H * H), x0(keep * H * 16);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.6f;
  for (auto& v : x0) v = rnd() * 2.0f;
  // fp64 evaluation of the chain for the kept tiles
  std::vector<double> ref(keep * H * 16);
  for (int t = 0; t < keep; ++t)
    for (int n = 0; n < 16; ++n) {
      double x[H], y[H];
      for (int f = 0; f < H; ++f) x[f] = x0[t * H * 16 + f * 16 + n];
      for (int l = 0; l < L; ++l) {
        for (int o = 0; o < H; ++o) {
          double acc = 0;
          for (int k = 0; k < H; ++k) acc += (double)w[(l * H + o) * H + k] * x[k];
          y[o] = (acc > 0 ? acc : 0) + (double)0.01f;
        }
        for (int f = 0; f < H; ++f) x[f] = y[f];
      }
      for (int f = 0; f < H; ++f) ref[t * H * 16 + f * 16 + n] = x[f];
    }
  float *dw, *dx, *dout;
  hipMalloc(&dw, w.size() * 4); hipMalloc(&dx, x0.size() * 4); hipMalloc(&dout, keep * H * 16 * 4);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x0.data(), x0.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> o(keep * H * 16);
  double refmax = 0;
  for (double v : ref) refmax = fmax(refmax, fabs(v));
  printf("chain of %d layers 32 -> 32, %d workgroups x %d waves, %d tiles of 16 samples per wave; max |out| = %.3f\n", L, blocks, NW,
         tiles, refmax);
  printf("%-6s %4s %12s %12s %12s\n", "kernel", "EV", "ns/layer-tile", "vs f32", "max err/max");
  for (int ev : {0, 32, 64, 128}) {
    double base = 0;
    for (int variant = 0; variant < 3; ++variant) {
      Args a{dw, dx, dout, tiles, keep, ev};
      auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(chain_f32, dim3(blocks), dim3(64 * NW), 0, 0, a);
        else if (variant == 1) hipLaunchKernelGGL(chain_bf<3>, dim3(blocks), dim3(64 * NW), 0, 0, a);
        else hipLaunchKernelGGL(chain_bf<1>, dim3(blocks), dim3(64 * NW), 0, 0, a);
      };
      hipMemset(dout, 0, keep * H * 16 * 4);
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
      double err = 0;
      for (size_t i = 0; i < o.size(); ++i) err = fmax(err, fabs((double)o[i] - ref[i]));
      // per SIMD: 2 waves x tiles x L layer-tiles in ms / 5 (two rounds of workgroups per CU)
      const double ns = ms / 5 * 1e6 / ((double)blocks / 256 * tiles * L);
      if (variant == 0) base = ns;
      printf("%-6s %4d %12.1f %12.2f %12.2e\n", variant == 0 ? "f32" : variant == 1 ? "x3" : "x1", ev, ns, base / ns, err / refmax);
    }
  }
  return 0;
}
