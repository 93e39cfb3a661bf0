// Layer-wise training iteration for ANY hidden width (multiple of 32): the shared background network
// (hidden 128, train.py:447-463) and the hidden-256 stress configuration.  The hidden-32 object networks
// use the fused kernel in objnerf_train.hip; wider networks do not fit one CU's LDS / registers, so this
// path materialises activations in the caller's workspace and runs the contraction as batched fp32 MFMA
// GEMMs (v_mfma_f32_16x16x4_f32, 64x64x16 tiles), with the reference's op order:
//   embedding.py:46-55 -> model.py:61-103 -> loss.py:5-103 (objnerf_step_batch_loss) -> reverse.
#include <algorithm>
#include "objnerf_device.h"
#include "../../include/objnerf_hip.h"
#include "objnerf_generic.h"

namespace objgen {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// C[m][n] (+)= sum_k A(m,k) B(k,n), generic strides, batched over blockIdx.z.  Epilogue (in order):
// + bias[n], * scale, relu, mask (C = mask(m,n) > 0 ? C : 0).
struct Gemm {
  int M, N, Kd;
  const float* A; long sam, sak, bsa;
  const float* B; long sbk, sbn, bsb;
  float* C; long scm, scn, bsc;
  const float* bias; long bsbias;
  const float* mask; long smm, smn, bsm;
  int accumulate, relu;
  int splitk;      // > 1: blockIdx.z = batch * splitk + slice; each slice atomically adds its partial into C
  float* rowsum; long bsrs;   // optional: rowsum[m] += sum_k A(m,k)  (bias gradient riding on the weight-gradient GEMM)
  const float* biasrow; long bsbr;   // optional per-row factor of the bias: + bias[n] * biasrow[m sbr]
  int sbr;
  float* part; float* rs_part;   // split-K with these set: slice s of batch z STORES its partial tile at part[((z sk + s) M + m)
                                 // N + n] (row sums: rs_part[(z sk + s) M + m]) and reduce_parts_kernel adds the slices in
                                 // order -- bit-reproducible; NULL: float atomics into C / rowsum
  int vec4;        // A is read with dwordx4 loads: AFULL panel rows / k-major A rows are 16-byte aligned
  const void* Bp;  // AFULL: B packed as 16-bit MFMA operands, [z][k / 32][n / 16][lane][8] (pack_b_kernel)
  int a16, b16, m16, c16;   // 16-bit kernels: A / B / mask / C are arrays of the kernel's operand type (the layer-wise
                            // path keeps h1 .. hc in 16 bit in the 16-bit modes); element strides as for floats
  const float* A2; long sam2, bsa2; int k2;   // AFULL: columns k >= k2 of A come from A2[m sam2 + (k - k2)] (the
                                              // concatenated layers [h | x] as ONE contraction); k2 >= Kd: none
  float a_scale;   // 16-bit operand kernels: A is multiplied by this power of two before it is rounded and the result
                   // divided by it (keeps back-propagated gradients out of fp16's subnormal range); 1 elsewhere
};

#ifndef OBJ_GEMM_BK
#define OBJ_GEMM_BK 16
#endif
constexpr int BK = OBJ_GEMM_BK;
#ifndef OBJ_WGRAD_MINPER
#define OBJ_WGRAD_MINPER 256
#endif

// Workgroup tile (32*TM*2) x (32*TN*2): 4 waves as 2 x 2, each wave TM x TN MFMA tiles of 16 x 16.
// <1,1> = 64 x 64 (small problems), <2,2> = 128 x 128 (the n x H x H layer GEMMs: 2 MFMAs per LDS read).
template <int TM, int TN, int BKT = BK, int WM = 2, int WN = 2>
__device__ __forceinline__ void gemm_tile(const Gemm& g, const int bx, const int by, const int bz) {
  constexpr int BK = BKT;      // k depth of one LDS stage (shadows the default)
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;      // (per wave 16*TM x 16*TN, workgroup WM x WN waves)
  constexpr int NTH = 64 * WM * WN;
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, gg = lane >> 4;
  const int wm = w / WN, wn = w % WN;
  const int m0 = by * BM, n0 = bx * BN;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  const long z = bz / sk;
  const int slice = bz % sk;
  const float* A = g.A + z * g.bsa;
  const float* B = g.B + z * g.bsb;
  float* C = g.C + z * g.bsc;
  const int kper = ((g.Kd + sk - 1) / sk + BK - 1) / BK * BK;
  const int kbeg = slice * kper, kend = min(g.Kd, kbeg + kper);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NA = BM * BK / NTH, NB = BN * BK / NTH;
  float ra[NA], rb[NB];
  // a thread's elements of the two tiles: fixed (row, k-slot) per element, so each keeps a pointer that advances by one
  // k stage per load (the per-element stride products of every stage were ~12 VALU instructions per load -- 40 % on top
  // of the MFMA time of the 8-deep wide kernel, and VALU time adds to fp32 MFMA time here)
  const float* pa[NA]; const float* pb[NB];
  int ka[NA], kb[NB];          // the element's current k; < 0: row out of range
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + NTH * i;
    int am, ak;
    if (g.sak == 1) { ak = e % BK; am = e / BK; } else { am = e % BM; ak = e / BM; }
    const int gm = m0 + am;
    pa[i] = A + (long)gm * g.sam + (long)(kbeg + ak) * g.sak;
    ka[i] = gm < g.M ? kbeg + ak : INT_MAX / 2;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + NTH * i;
    int bn, bk;
    if (g.sbk == 1) { bk = e % BK; bn = e / BK; } else { bn = e % BN; bk = e / BN; }
    const int gn = n0 + bn;
    pb[i] = B + (long)(kbeg + bk) * g.sbk + (long)gn * g.sbn;
    kb[i] = gn < g.N ? kbeg + bk : INT_MAX / 2;
  }
  const long stepa = (long)BK * g.sak, stepb = (long)BK * g.sbk;
  auto load_tiles = [&](int) {                 // (stages are requested in order: the argument is implied)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      ra[i] = ka[i] < kend ? *pa[i] : 0.f;
      pa[i] += stepa; ka[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      rb[i] = kb[i] < kend ? *pb[i] : 0.f;
      pb[i] += stepb; kb[i] += BK;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + NTH * i;
      int am, ak;
      if (g.sak == 1) { ak = e % BK; am = e / BK; } else { am = e % BM; ak = e / BM; }
      As[ak][am] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + NTH * i;
      int bn, bk;
      if (g.sbk == 1) { bk = e % BK; bn = e / BK; } else { bn = e % BN; bk = e / BN; }
      Bs[bk][bn] = rb[i];
    }
  };
  float rs = 0.f;
  const bool do_rs = g.rowsum != nullptr && bx == 0 && tid < BM;
  if (kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store_tiles();
    __syncthreads();
    if (k0 + BK < kend) load_tiles(k0 + BK);       // next tile's global loads fly under the MFMAs
    if (do_rs) {
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) rs += As[kk][tid];
    }
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[4 * ks + gg][16 * TM * wm + 16 * i + c];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[4 * ks + gg][16 * TN * wn + 16 * j + c];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_rs && m0 + tid < g.M) {
    if (g.rs_part) g.rs_part[(z * sk + slice) * g.M + m0 + tid] = rs;
    else atomicAdd(g.rowsum + z * g.bsrs + m0 + tid, rs);
  }
  // epilogue: D layout: col n = lane & 15, row m = 4 * (lane >> 4) + r.  Offsets are advanced incrementally (the 64-bit
  // stride products per element were ~15 VALU instructions each, and VALU time adds to fp32 MFMA time on this part).
  const int mb = m0 + 16 * TM * wm + 4 * gg, nb = n0 + 16 * TN * wn + c;
  long ci = (long)mb * g.scm + (long)nb * g.scn;                        // into C (this batch entry)
  long mi = z * g.bsm + (long)mb * g.smm + (long)nb * g.smn;            // into g.mask
  long ri = z * g.bsbr + (long)mb * g.sbr;                              // into g.biasrow
  const long c16m = 16 * g.scm, c16n = 16 * g.scn, m16m = 16 * g.smm, m16n = 16 * g.smn, r16 = 16 * (long)g.sbr;
#pragma unroll
  for (int i = 0; i < TM; ++i, ci += c16m, mi += m16m, ri += r16) {
    long cj = ci, mj = mi;
#pragma unroll
    for (int j = 0; j < TN; ++j, cj += c16n, mj += m16n) {
      const int n = nb + 16 * j;
      const float bn_ = (g.bias && n < g.N && sk <= 1) ? g.bias[z * g.bsbias + n] : 0.f;
      long co = cj, mo = mj, ro = ri;
#pragma unroll
      for (int r = 0; r < 4; ++r, co += g.scm, mo += g.smm, ro += g.sbr) {
        const int m = mb + 16 * i + r;
        if (m < g.M && n < g.N) {
          float v = acc[i][j][r];
          if (sk > 1) {
            if (g.part) g.part[((z * sk + slice) * g.M + m) * g.N + n] = v;
            else atomicAdd(C + co, v);
            continue;
          }
          if (g.accumulate) v += C[co];
          if (g.bias) v += bn_ * (g.biasrow ? g.biasrow[ro] : 1.0f);
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.mask) v = g.mask[mo] > 0.f ? v : 0.f;
          C[co] = v;
        }
      }
    }
  }
}


template <int TM, int TN, int BKT = BK>
__global__ __launch_bounds__(256) void gemm_kernel(const Gemm g) {
  gemm_tile<TM, TN, BKT>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}
// 128 x 128 tile on 8 waves (4 x 2 waves of 32 x 64)
#ifndef OBJ_GEMM_BK_WIDE
#define OBJ_GEMM_BK_WIDE 8      // measured on the hidden-256 layer GEMMs (configs[4] share): 8 -> 422 ms, 16 -> 452, 32 -> 457
#endif
#ifndef OBJ_GEMM_WIDE_TM
#define OBJ_GEMM_WIDE_TM 2      // row tiles per wave of the wide kernel: workgroup tile (64 TM) x 128
#endif
__global__ __launch_bounds__(512, 8) void gemm_kernel8(const Gemm g) {
  gemm_tile<OBJ_GEMM_WIDE_TM, 4, OBJ_GEMM_BK_WIDE, 4, 2>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent GEMMs in ONE launch (the weight-gradient GEMMs of a small-batch step: each alone is ~260
// workgroups of latency-bound work): blockIdx.z runs over the concatenated (batch x split-K) slices of all of them.
// Optional extra workgroups of a grouped launch (blockIdx.z >= zbeg[count], x = y = 0): the feature branch's small
// preparation passes of the one-launch background iteration -- wb = W_of^T b_of, bb; beta = b_of . g, |g| per ray; a copy of
// [W_of | b_of] for the head's finish -- ride on the launch of its two preparation GEMMs (round 6: three launches fewer).
struct FeatPrepTask {
  int blocks = 0;                         // 0: no task
  const float* params; long ps; int off_w, off_b, C, Hh, R, rin_ld, K;
  const float* gt_feat; float* rayin; float* gram; long gstride; float* snap;
  int nb_wb, nb_rs, nb_snap;              // workgroups per object of the three passes
};
struct GemmGroup {
  static constexpr int MAXG = 11;
  int count;
  int zbeg[MAXG + 1];
  Gemm g[MAXG];
  FeatPrepTask task;
};
__device__ void feat_prep_task(const FeatPrepTask& t, int b);
// 8 waves per workgroup on a 128 x 128 output tile (4 x 2 waves of 32 x 64): every operand slice is read once
// (64 x 64 tiles of 4 waves read it once per tile: 0.297 -> 0.287 ms for the background step)
__global__ __launch_bounds__(512) void gemm_group_kernel(const GemmGroup gr) {
  if ((int)blockIdx.z >= gr.zbeg[gr.count]) {            // (extra workgroups: FeatPrepTask)
    if (blockIdx.x == 0 && blockIdx.y == 0) feat_prep_task(gr.task, (int)blockIdx.z - gr.zbeg[gr.count]);
    return;
  }
  int i = 0;
  while (i + 1 < gr.count && (int)blockIdx.z >= gr.zbeg[i + 1]) ++i;
  const Gemm& g = gr.g[i];
  if ((int)blockIdx.x * 128 >= g.N || (int)blockIdx.y * 128 >= g.M) return;
  gemm_tile<2, 4, BK, 4, 2>(g, blockIdx.x, blockIdx.y, blockIdx.z - gr.zbeg[i]);
}

// ------------------------------------------------------------------------------------------------
// bf16-operand variant (OBJNERF_TRAIN_BF16 on the layer-wise path): same interface and epilogue, operands are
// rounded to bf16 when they are staged in LDS ([row][k], k contiguous: one ds_read_b128 per MFMA operand),
// v_mfma_f32_16x16x32_bf16, fp32 accumulation.  64 x 64 (or, for wide layers, 128 x 128) tiles, 32-deep k steps:
// 4 (16) MFMAs per wave and step, so the kernel is bound by the operand traffic, not by the matrix core.
//
// OT = _Float16 is the OBJNERF_TRAIN_FP16 mode (v_mfma_f32_16x16x32_f16): 11 significant bits instead of 8, but a
// narrow exponent -- operands are clamped to +-65504 when rounded, and the backward GEMMs scale their gradient
// operand (Gemm::a_scale).
#ifndef OBJ_G16_BK
#define OBJ_G16_BK 32
#endif
#ifndef OBJ_G16_AFULL
#define OBJ_G16_AFULL 1
#endif
#ifndef OBJ_G16_WIDE_NARROW
#define OBJ_G16_WIDE_NARROW 1
#endif
#ifndef OBJ_ACT16
#define OBJ_ACT16 1
#endif
#ifndef OBJ_G16_AFULL_NARROW
#define OBJ_G16_AFULL_NARROW 1
#endif
#ifndef OBJ_G16_XCD
#define OBJ_G16_XCD 1
#endif
#ifndef OBJ_G16_BK_WIDE
#define OBJ_G16_BK_WIDE 32
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename OT> struct Op16;
template <> struct Op16<__bf16> {
  typedef bf16x8 V;
  static __device__ __forceinline__ __bf16 cvt(float x) { return (__bf16)x; }
  static __device__ __forceinline__ f32x4 mfma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Op16<_Float16> {
  typedef f16x8 V;
  static __device__ __forceinline__ _Float16 cvt(float x) { return (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f); }
  static __device__ __forceinline__ f32x4 mfma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
// BKB: k-stage depth (OBJ_G16_BK / OBJ_G16_BK_WIDE).  Measured on MI355X (tools/gemm16_ab.sh): deeper stages LOSE --
// background step in bf16 mode 0.92 ms at 32, 1.19 ms at 64, 2.2 ms at 128; configs[4] share in fp16 2.18 / 2.37 /
// 2.60 s -- the prefetch registers and the larger LDS tiles cost more occupancy than the saved barriers return.  (Again
// with both weight-gradient operands 16-bit pass-through, 8 registers per stage: share per 8 objects 83 ms at 32,
// 104 at 64, 85 at 128.)
// WM x WN waves per workgroup, TM x TN MFMA tiles per wave.  The 128 x 128 tile runs on 8 waves (4 x 2, 32 x 64 per wave:
// 32 accumulator and 16 staging registers): as 4 waves of 64 x 64 it needed 197 registers, ONE wave per SIMD, and ran
// the hidden-256 layer GEMMs at 28 TFLOP/s -- slower than the fp32 kernel.
//
// AKM / BKM: the operand's ROWS (m / n) are the contiguous dimension in memory (weight gradients: A = d_out^T, B =
// activations; input gradients: B = W as stored).  Such a tile is staged K-MAJOR -- four consecutive rows per thread,
// one 8-byte LDS store, coalesced -- and the MFMA operand (8 consecutive k of one row) is gathered by two
// ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads a 4 k x 16 rows block column-wise;
// tools/ubench_tr.hip checks the lane map).  Before, those tiles went to the [row][k] image as 2-byte stores with a
// 4-way bank conflict and the hidden-256 weight gradients ran at 18 TFLOP/s.
// The read is the compiler's builtin (__builtin_amdgcn_ds_read_tr16_b64_*): it knows the instruction returns through
// LGKM, so it places the s_waitcnt itself and a batch of reads stays in flight together.  (Rounds 2-3 issued it as bare
// inline asm with a hand-written wait tied to the destination registers: correct only as long as the register
// allocator never copied a destination before that wait.)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const void* p) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(uint2, v);
}
// (the 8-wave tile is held to 128 registers -- two workgroups per CU: at 130 the fp16 forward variant lost one)
//
// AFULL (layer GEMMs over the sample axis: M = 10^6 rows, contraction <= 256, A rows k-contiguous): the workgroup's A
// panel -- BM rows x the WHOLE contraction -- is fetched with one burst of loads before the k-loop (64 KB in flight
// per workgroup) and stays in LDS.  B -- the layer's weights -- is NOT staged by the workgroup at all: pack_b_kernel
// rounds it ONCE per GEMM into a 16-bit image laid out as MFMA operands ([k / 32][n / 16][lane][8], 128 KB for 256 x
// 256, L2-resident) and every lane fetches its operand with one 16-byte load.  No barrier and no conversion in the
// k-loop.  (The plain kernel re-read and re-rounded the fp32 weights in every workgroup: rocprof counted 36 VALU
// instructions per MFMA on the hidden-256 layer GEMMs, which ran at ~2 TB/s of HBM traffic with the matrix core 90 %
// idle.)  With BN = 256 the panel is read exactly once.  The epilogue's LDS patch aliases the panel.
// (the body takes its block coordinates as arguments: gemm_group16_kernel below runs it for several GEMMs per launch)
template <int TM, int TN, typename OT, int BKB, int WM = 2, int WN = 2, bool AKM = false, bool BKM = false,
          bool AFULL = false, int KMAX = 256>
__device__ __forceinline__ void gemm_bf16_body(const Gemm& g, const int bx, const int by, const int bz) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN, LDK = BKB + 8, NTH = 64 * WM * WN;
  constexpr int PA = BM + 8, PB = BN + 8;                 // k-major row pitches (elements; 8-byte aligned rows)
  constexpr int LDA = AFULL ? KMAX + 8 : LDK;   // AFULL: panel row pitch (conflict-free b128 reads: 132 / 180 dwords)
  static_assert(!(AFULL && AKM), "the resident panel is row-major");
  typedef typename Op16<OT>::V OV;
  constexpr int EC_ = 16 * TN, EP_ = EC_ + 4;
  constexpr size_t a_bytes = sizeof(OT) * (AFULL ? BM * LDA : (AKM ? BKB * PA : BM * LDK));
  constexpr size_t ep_bytes = sizeof(float) * WM * WN * 16 * EP_;
  __shared__ __attribute__((aligned(16))) char a_raw[AFULL ? (a_bytes > ep_bytes ? a_bytes : ep_bytes) : a_bytes];
  OT* const Asm = reinterpret_cast<OT*>(a_raw);
  __shared__ __attribute__((aligned(16))) OT Bsm[AFULL ? 8 : (BKM ? BKB * PB : BN * LDK)];
  const float a_scale = g.a_scale, inv_scale = 1.0f / g.a_scale;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, gg = lane >> 4;
  const int wm = w / WN, wn = w % WN;
  const int m0 = by * BM, n0 = bx * BN;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  const long z = bz / sk;
  const int slice = bz % sk;
  const float* A = g.A + z * g.bsa;
  const float* B = g.B + z * g.bsb;
  const int kper = ((g.Kd + sk - 1) / sk + BKB - 1) / BKB * BKB;
  const int kbeg = slice * kper, kend = min(g.Kd, kbeg + kper);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NA = AFULL ? 0 : BM * BKB / NTH, NB = AFULL ? 0 : BN * BKB / NTH;
  float ra[NA ? NA : 1], rb[NB ? NB : 1];
  if constexpr (AFULL) {
    // the panel: columns [0, ka) from A (lanes along k; 16-byte loads when g.vec4: rows 16-byte aligned, ka a multiple
    // of 8), then, for the 352-deep variant, columns [256, 352) from the second source (x1 / x2 inside the 129-float
    // embedding rows: scalars).  Everything else of the panel is zero.
    constexpr int KA = 256;
    const int ka = min(g.k2, g.Kd);                     // columns [0, ka) from A, [ka, Kd) from A2 (then ka == KA)
    if (g.a16) {
      constexpr int N8 = BM * KA / 8 / NTH;
      const OT* A16 = reinterpret_cast<const OT*>(g.A) + z * g.bsa;
      uint4 pv[N8];
#pragma unroll
      for (int i = 0; i < N8; ++i) {
        const int q = tid + NTH * i, am = q / (KA / 8), ak = 8 * (q % (KA / 8));
        const int gm = m0 + am;
        pv[i] = (gm < g.M && ak < ka) ? *reinterpret_cast<const uint4*>(A16 + gm * g.sam + ak) : uint4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int i = 0; i < N8; ++i) {
        const int q = tid + NTH * i, am = q / (KA / 8), ak = 8 * (q % (KA / 8));
        *reinterpret_cast<uint4*>(&Asm[am * LDA + ak]) = pv[i];
      }
    } else if (g.vec4) {
      constexpr int NV = BM * KA / 4 / NTH;
      f32x4 pv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int q = tid + NTH * i, am = q / (KA / 4), ak = 4 * (q % (KA / 4));
        const int gm = m0 + am;
        pv[i] = (gm < g.M && ak < ka) ? *reinterpret_cast<const f32x4*>(A + gm * g.sam + ak) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int q = tid + NTH * i, am = q / (KA / 4), ak = 4 * (q % (KA / 4));
        typedef OT ot4 __attribute__((ext_vector_type(4)));
        ot4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Op16<OT>::cvt(a_scale != 1.0f ? pv[i][j] * a_scale : pv[i][j]);
        *reinterpret_cast<ot4*>(&Asm[am * LDA + ak]) = v;
      }
    } else {
      constexpr int NS = BM * KA / NTH / 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {          // two bursts of scalar loads
        float ps[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int e = tid + NTH * (NS * h + i), am = e / KA, ak = e % KA;
          const int gm = m0 + am;
          ps[i] = (gm < g.M && ak < ka) ? A[gm * g.sam + ak] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int e = tid + NTH * (NS * h + i), am = e / KA, ak = e % KA;
          Asm[am * LDA + ak] = Op16<OT>::cvt(ps[i] * a_scale);
        }
      }
    }
    if constexpr (KMAX > KA) {
      constexpr int K2 = KMAX - KA, N2 = BM * K2 / NTH;
      static_assert(BM * K2 % NTH == 0, "second-source share");
      const float* A2 = g.A2 + z * g.bsa2;
      float ps[N2];
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        const int e = tid + NTH * i, am = e / K2, ak = KA + e % K2;
        const int gm = m0 + am;
        ps[i] = (gm < g.M && ak >= ka && ak < g.Kd) ? A2[gm * g.sam2 + (ak - ka)] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        const int e = tid + NTH * i, am = e / K2, ak = KA + e % K2;
        Asm[am * LDA + ak] = Op16<OT>::cvt(ps[i] * a_scale);
      }
    }
  }
  // element e of a thread's share -> (row, k): lanes along the operand's contiguous dimension
  auto load_tiles = [&](int k0) {
    if (AKM && g.a16) {
      // A is a 16-bit gradient array (stored already scaled by a_scale): four consecutive m per lane, one 8-byte load,
      // passed through to the k-major image (host-checked alignment)
      const OT* A16 = reinterpret_cast<const OT*>(g.A) + z * g.bsa;
#pragma unroll
      for (int i = 0; i < NA / 4; ++i) {
        const int q = tid + NTH * i, am = 4 * (q % (BM / 4)), ak = q / (BM / 4);
        const int gm = m0 + am, gk = k0 + ak;
        const uint2 u = (gm < g.M && gk < kend) ? *reinterpret_cast<const uint2*>(A16 + gk * g.sak + gm) : uint2{0u, 0u};
        ra[2 * i] = __builtin_bit_cast(float, u.x); ra[2 * i + 1] = __builtin_bit_cast(float, u.y);
      }
    } else if (AKM && g.vec4) {
      // rows contiguous and 16-byte aligned (weight gradients: A = d_out^T): four consecutive m per lane, one dwordx4
#pragma unroll
      for (int i = 0; i < NA / 4; ++i) {
        const int q = tid + NTH * i, am = 4 * (q % (BM / 4)), ak = q / (BM / 4);
        const int gm = m0 + am, gk = k0 + ak;
        const f32x4 v = (gm < g.M && gk < kend) ? *reinterpret_cast<const f32x4*>(A + gk * g.sak + gm) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[4 * i + j] = v[j];
      }
    } else
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int am, ak;
      if (AKM) { const int e = tid + NTH * i; am = e % BM; ak = e / BM; }
      else { const int e = tid + NTH * i; ak = e % BKB; am = e / BKB; }
      const int gm = m0 + am, gk = k0 + ak;
      ra[i] = (gm < g.M && gk < kend) ? A[gm * g.sam + gk * g.sak] : 0.f;
    }
    if (BKM && g.b16) {
      // B is a 16-bit activation array (weight gradients in the 16-bit modes): four consecutive n per lane, one 8-byte
      // load, passed through to the k-major image unchanged (host-checked: sbn == 1, sbk and N multiples of 4)
      const OT* B16 = reinterpret_cast<const OT*>(g.B) + z * g.bsb;
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) {
        const int q = tid + NTH * i, bn = 4 * (q % (BN / 4)), bk = q / (BN / 4);
        const int gn = n0 + bn, gk2 = k0 + bk;
        const uint2 u = (gn < g.N && gk2 < kend) ? *reinterpret_cast<const uint2*>(B16 + gk2 * g.sbk + gn) : uint2{0u, 0u};
        rb[2 * i] = __builtin_bit_cast(float, u.x); rb[2 * i + 1] = __builtin_bit_cast(float, u.y);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int bn, bk;
      if (BKM) { const int e = tid + NTH * i; bn = e % BN; bk = e / BN; }
      else { const int e = tid + NTH * i; bk = e % BKB; bn = e / BKB; }
      const int gn = n0 + bn, gk2 = k0 + bk;
      rb[i] = (gn < g.N && gk2 < kend) ? B[gk2 * g.sbk + gn * g.sbn] : 0.f;
    }
  };
  auto store_tiles = [&]() {
    if (AFULL) {
    } else if (AKM && g.a16) {
#pragma unroll
      for (int i = 0; i < NA / 4; ++i) {
        const int q = tid + NTH * i;
        *reinterpret_cast<uint2*>(&Asm[(q / (BM / 4)) * PA + 4 * (q % (BM / 4))]) =
            uint2{__builtin_bit_cast(uint32_t, ra[2 * i]), __builtin_bit_cast(uint32_t, ra[2 * i + 1])};
      }
    } else if (AKM && g.vec4) {
#pragma unroll
      for (int i = 0; i < NA / 4; ++i) {
        const int q = tid + NTH * i;
        typedef OT ot4 __attribute__((ext_vector_type(4)));
        ot4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Op16<OT>::cvt(ra[4 * i + j] * a_scale);
        *reinterpret_cast<ot4*>(&Asm[(q / (BM / 4)) * PA + 4 * (q % (BM / 4))]) = v;
      }
    } else if (AKM) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = tid + NTH * i;
        Asm[(e / BM) * PA + e % BM] = Op16<OT>::cvt(ra[i] * a_scale);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = tid + NTH * i;
        Asm[(e / BKB) * LDK + e % BKB] = Op16<OT>::cvt(ra[i] * a_scale);
      }
    }
    if (BKM && g.b16) {
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) {
        const int q = tid + NTH * i;
        *reinterpret_cast<uint2*>(&Bsm[(q / (BN / 4)) * PB + 4 * (q % (BN / 4))]) =
            uint2{__builtin_bit_cast(uint32_t, rb[2 * i]), __builtin_bit_cast(uint32_t, rb[2 * i + 1])};
      }
    } else if (BKM) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + NTH * i;
        Bsm[(e / BN) * PB + e % BN] = Op16<OT>::cvt(rb[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + NTH * i;
        Bsm[(e / BKB) * LDK + e % BKB] = Op16<OT>::cvt(rb[i]);
      }
    }
  };
  // MFMA operands of the wave's T row tiles (first tile-local row row0), k-slots ks + 8 gg .. + 7
  auto operands_km = [&](auto& out, const OT* img, const int pitch, const int row0, const int ks) {
    constexpr int T = sizeof(out) / sizeof(out[0]);
    uint2 lo[T], hi[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const OT* p = img + (ks + 8 * gg + (c >> 2)) * pitch + row0 + 16 * i + 4 * (c & 3);
      lo[i] = lds_tr16(p);
      hi[i] = lds_tr16(p + 4 * pitch);
    }
#pragma unroll
    for (int i = 0; i < T; ++i) out[i] = __builtin_bit_cast(OV, uint4{lo[i].x, lo[i].y, hi[i].x, hi[i].y});
  };
  auto operands_rm = [&](auto& out, const OT* img, const int pitch, const int row0, const int ks) {
    constexpr int T = sizeof(out) / sizeof(out[0]);
#pragma unroll
    for (int i = 0; i < T; ++i) out[i] = *reinterpret_cast<const OV*>(img + (row0 + 16 * i + c) * pitch + ks + 8 * gg);
  };
  float rs = 0.f;
  const bool do_rs = !AFULL && g.rowsum != nullptr && bx == 0 && tid < BM;
  if constexpr (AFULL) {
    __syncthreads();                     // the panel is complete
    const int nks = (g.Kd + 31) / 32;
    const uint4* bp = reinterpret_cast<const uint4*>(g.Bp) + ((z * (KMAX / 32)) * (BN / 16) + TN * wn) * 64 + lane;
    for (int ks = 0; ks < nks; ++ks) {     // (no operand prefetch: 64 registers -> four workgroups per CU hide the round trip)
      OV bcur[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bcur[j] = __builtin_bit_cast(OV, bp[(ks * (BN / 16) + j) * 64]);
      OV a[TM];
      operands_rm(a, Asm, LDA, 16 * TM * wm, 32 * ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(a[i], bcur[j], acc[i][j]);
    }
    __syncthreads();                     // every wave is done with the panel: the epilogue patch may overwrite it
  }
  if (!AFULL && kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; !AFULL && k0 < kend; k0 += BKB) {
    store_tiles();
    __syncthreads();
    if (k0 + BKB < kend) load_tiles(k0 + BKB);
    if (do_rs) {
#pragma unroll
      for (int kk = 0; kk < BKB; ++kk) rs += (float)(AKM ? Asm[kk * PA + tid] : Asm[tid * LDK + kk]);
    }
#pragma unroll
    for (int ks = 0; ks < BKB; ks += 32) {
      OV a[TM], b[TN];
      if (AKM) operands_km(a, Asm, PA, 16 * TM * wm, ks); else operands_rm(a, Asm, LDA, 16 * TM * wm, ks);
      if (BKM) operands_km(b, Bsm, PB, 16 * TN * wn, ks); else operands_rm(b, Bsm, LDK, 16 * TN * wn, ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = Op16<OT>::mfma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (do_rs && m0 + tid < g.M) {
    if (g.rs_part) g.rs_part[(z * sk + slice) * g.M + m0 + tid] = rs * inv_scale;
    else atomicAdd(g.rowsum + z * g.bsrs + m0 + tid, rs * inv_scale);
  }
  // Epilogue through LDS: the accumulator layout gives a lane ONE column of four rows, i.e. 64-byte runs per output row
  // (and the same for the mask / accumulate reads) -- each wave turns a 16-row strip of its tile around in its own LDS
  // patch so that the lanes run ALONG a row: 256-byte (128-byte for the 64-wide tile) coalesced rows for every global
  // access of the epilogue.
  constexpr int EC = 16 * TN, EP = EC + 4, RPI = 64 / EC;            // strip columns, pitch, rows per pass
  __shared__ float Ep_own[AFULL ? 1 : WM * WN][AFULL ? 1 : 16][AFULL ? 1 : EP];
  float (*const Ep)[16][EP] = AFULL ? reinterpret_cast<float (*)[16][EP]>(a_raw) : reinterpret_cast<float (*)[16][EP]>(Ep_own);
  const int ecol = lane % EC, erow0 = lane / EC;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ep[w][4 * gg + r][16 * j + c] = acc[i][j][r] * inv_scale;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int n = n0 + EC * wn + ecol;
    // (element offsets of the strip's first row, advanced by one row pass each: the per-row 64-bit products of the
    // generic strides were most of this kernel's VALU work -- 15 instructions per MFMA on the hidden-256 layer GEMMs)
    const int mfirst = m0 + 16 * TM * wm + 16 * i + erow0;
    long coff = z * g.bsc + (long)mfirst * g.scm + (long)n * g.scn;             // into g.C (float or 16-bit elements)
    long moff = z * g.bsm + (long)mfirst * g.smm + (long)n * g.smn;             // into g.mask
    long roff = z * g.bsbr + (long)mfirst * g.sbr;                              // into g.biasrow
    const long cstep = (long)RPI * g.scm, mstep = (long)RPI * g.smm, rstep = (long)RPI * g.sbr;
    const float bn_ = (g.bias && n < g.N) ? g.bias[z * g.bsbias + n] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; rr += RPI, coff += cstep, moff += mstep, roff += rstep) {
      const int m = mfirst + rr;
      float v = Ep[w][rr + erow0][ecol];
      if (m < g.M && n < g.N) {
        if (sk > 1) {
          if (g.part) g.part[((z * sk + slice) * g.M + m) * g.N + n] = v;
          else atomicAdd(g.C + coff, v);
          continue;
        }
        if (g.accumulate) v += g.c16 ? (float)reinterpret_cast<const OT*>(g.C)[coff] * inv_scale : g.C[coff];
        if (g.bias) v += bn_ * (g.biasrow ? g.biasrow[roff] : 1.0f);
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.mask) {
          const bool on = g.m16 ? (float)reinterpret_cast<const OT*>(g.mask)[moff] > 0.f : g.mask[moff] > 0.f;
          v = on ? v : 0.f;
        }
        // (16-bit outputs are stored as the NEXT GEMM's operand: gradients already scaled by a_scale -- 1 in the forward)
        if (g.c16) reinterpret_cast<OT*>(g.C)[coff] = Op16<OT>::cvt(v * a_scale);
        else g.C[coff] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

template <int TM, int TN, typename OT, int BKB, int WM = 2, int WN = 2, bool AKM = false, bool BKM = false,
          bool AFULL = false, int KMAX = 256>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? ((AFULL && KMAX == 256) ? 8 : 4) : 1) void gemm_bf16_kernel(const Gemm g) {
  // Workgroups are handed to the 8 XCDs round-robin in launch order (x fastest), so the column tiles of one row
  // block -- which read the same A rows -- would land on different L2s.  Re-map: consecutive workgroups OF ONE XCD
  // take the column tiles of one row block (rows beyond the last multiple of 8 row blocks keep the plain order).
  // Split-K weight gradients (few output tiles, hundreds of slices): the tiles of ONE slice read the same operand slabs,
  // so they go to one XCD the same way.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (OBJ_G16_XCD && gridDim.x > 1 && gridDim.y >= 8) {
    const unsigned gx = gridDim.x, L = blockIdx.x + gx * blockIdx.y, full = (gridDim.y / 8) * 8 * gx;
    if (L < full) {
      const unsigned xcd = L % 8, s = L / 8;
      bx = (int)(s % gx);
      by = (int)((s / gx) * 8 + xcd);
    }
  } else if (OBJ_G16_XCD && gridDim.x * gridDim.y > 1 && gridDim.z >= 8) {
    const unsigned gx = gridDim.x, T = gx * gridDim.y, L = blockIdx.x + gx * blockIdx.y + T * blockIdx.z;
    const unsigned full = (gridDim.z / 8) * 8 * T;
    if (L < full) {
      const unsigned xcd = L % 8, s = L / 8, tile = s % T;
      bx = (int)(tile % gx);
      by = (int)(tile / gx);
      bz = (int)((s / T) * 8 + xcd);
    }
  }
  gemm_bf16_body<TM, TN, OT, BKB, WM, WN, AKM, BKM, AFULL, KMAX>(g, bx, by, bz);
}

// The weight-gradient GEMMs of a small-batch step in bf16 mode, in ONE launch (the fp32 twin: gemm_group_kernel): 64 x 64
// tiles on 4 waves, both operands k-major (d_out^T and the activations as they lie in memory), split-K partial tiles.
// OBJ_GROUP16_WIDE: 128 x 128 tiles on 8 waves (every operand slice read once, as the fp32 twin) instead of 64 x 64 on 4
#ifndef OBJ_GROUP16_WIDE
#define OBJ_GROUP16_WIDE 1
#endif
#if OBJ_GROUP16_WIDE
__global__ __launch_bounds__(512) void gemm_group16_kernel(const GemmGroup gr) {
  if ((int)blockIdx.z >= gr.zbeg[gr.count]) return;      // (tasks ride on the fp32 grouped launch only)
  int i = 0;
  while (i + 1 < gr.count && (int)blockIdx.z >= gr.zbeg[i + 1]) ++i;
  const Gemm& g = gr.g[i];
  if ((int)blockIdx.x * 128 >= g.N || (int)blockIdx.y * 128 >= g.M) return;
  gemm_bf16_body<2, 4, __bf16, OBJ_G16_BK_WIDE, 4, 2, true, true>(g, blockIdx.x, blockIdx.y, blockIdx.z - gr.zbeg[i]);
}
#else
__global__ __launch_bounds__(256) void gemm_group16_kernel(const GemmGroup gr) {
  if ((int)blockIdx.z >= gr.zbeg[gr.count]) return;      // (tasks ride on the fp32 grouped launch only)
  int i = 0;
  while (i + 1 < gr.count && (int)blockIdx.z >= gr.zbeg[i + 1]) ++i;
  const Gemm& g = gr.g[i];
  if ((int)blockIdx.x * 64 >= g.N || (int)blockIdx.y * 64 >= g.M) return;
  gemm_bf16_body<2, 2, __bf16, OBJ_G16_BK, 2, 2, true, true>(g, blockIdx.x, blockIdx.y, blockIdx.z - gr.zbeg[i]);
}
#endif

// B (generic strides, fp32) -> the AFULL kernel's operand image: thread (ks, jt, lane) writes the 8 values
// B[32 ks + 8 (lane >> 4) + e][16 jt + (lane & 15)], e = 0..7, rounded to OT; zero beyond Kd / N.
template <typename OT, int KS>
__global__ __launch_bounds__(256) void pack_b_kernel(const Gemm g, OT* __restrict__ out) {
  typedef typename Op16<OT>::V OV;
  const int t = blockIdx.x * 256 + threadIdx.x;              // (ks, jt, lane), KS * 16 * 64 per batch entry
  const int lane = t & 63, jt = (t >> 6) & 15, ks = t >> 10;
  const long z = blockIdx.y;
  const float* B = g.B + z * g.bsb;
  const int n = 16 * jt + (lane & 15);
  OV v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 8 * (lane >> 4) + e;
    v[e] = Op16<OT>::cvt((k < g.Kd && n < g.N) ? B[k * g.sbk + n * g.sbn] : 0.f);
  }
  reinterpret_cast<OV*>(out)[z * (KS * 1024) + t] = v;
}

// selected for the duration of one train_step call (single host thread per device)
struct A2Src { const float* A2; long sam2, bsa2; int k2; };
struct RedGroup;
// What a GEMM call needs beyond its operands: the operand precision of the step, the step's scratch regions and the
// one-shot modifiers of the NEXT call.  One instance per C-ABI call, on that call's stack, passed explicitly -- the
// library keeps no state between (or beside) calls.
struct GemmEnv {
  int operands = 0;                 // 0: fp32 GEMMs, 1: bf16 operands, 2: fp16 operands
  float a_scale = 1.0f;             // Gemm::a_scale of the next gemm(E, ) calls (fp16 backward)
  GemmGroup* group = nullptr;       // non-null: gemm(E, ) collects descriptors instead of launching
  bool group16 = false;             // the collected GEMMs run with bf16 operands (gemm_group16_kernel: small-batch step in bf16 mode)
  const float* biasrow = nullptr;   // Gemm::biasrow / bsbr / sbr of the next calls
  long bsbr = 0;
  int sbr = 1;
  void* packb = nullptr;            // scratch of the AFULL GEMMs' packed B images (176 KB per batch entry)
  long packb_entries = 0;
  // [lo, hi): the step's 16-bit activation buffers (h1 .. hc stored in the operand type) and the same for d_hc,
  // d_h4 .. d_h1 (stored scaled by a_scale); gemm(E, ) marks every operand that lies inside (Gemm::a16 / b16 / m16 / c16)
  const char *act16_lo = nullptr, *act16_hi = nullptr, *grad16_lo = nullptr, *grad16_hi = nullptr;
  A2Src a2 = {nullptr, 0, 0, 0};    // second A source of the NEXT gemm(E, ) call (Gemm::A2), panel path only
  // bump allocator over the caller's workspace region for the partial slabs of ONE step (the weight-gradient GEMMs of
  // a step run side by side, so each has its own slab); exhausted or absent -> atomics
  float* parts = nullptr;
  size_t parts_cap = 0, parts_off = 0;
  float* next_part = nullptr;       // Gemm::part / rs_part of the next gemm(E, ) call
  float* next_rs_part = nullptr;
  RedGroup* red_group = nullptr;    // non-null: reductions are collected (grouped launch)
  int max_slices = 0;               // > 0: wgrad(E, ) cuts the sample axis into at most this many split-K slices
  bool error = false;               // a GEMM of the step was handed operands its kernel does not take: train_step returns
                                    // OBJNERF_EINVAL (a library call never ends the host process)
  bool need_parts = false;          // wgrad(E, ): the slab form is required (the gradient arena is not zero-filled)
  bool parts_failed = false;        // ... and the scratch did not hold it
  bool in_act16(const void* p) const {
    return (act16_lo && (const char*)p >= act16_lo && (const char*)p < act16_hi) ||
           (grad16_lo && (const char*)p >= grad16_lo && (const char*)p < grad16_hi);
  }
  float* parts_alloc(size_t floats) {
    floats = (floats + 63) / 64 * 64;
    if (!parts || parts_off + floats > parts_cap) return nullptr;
    float* p = parts + parts_off;
    parts_off += floats;
    return p;
  }
};
// per-row bias factor of the NEXT gemm(E, ) call (feature_head), reset by the caller

// ---- deterministic split-K: partial-tile slabs + an ordered reduction (instead of float atomics)
#ifndef OBJ_WGRAD_TARGET
#define OBJ_WGRAD_TARGET 512         // 64 x 64 output tiles x slices a weight-gradient GEMM should at least have
#endif
#ifndef OBJ_WGRAD_MAXSLICES
#define OBJ_WGRAD_MAXSLICES 128      // (32 was measured: the background step in bf16 mode 0.92 -> 1.4 ms, too few workgroups)
#endif
// split-K slices of a weight-gradient GEMM over n samples: 512 samples per slice; 256 when that would leave most of the
// chip idle (the background network at the reference's native batch of 16 800 samples: step 0.55 -> 0.46 ms)
static int wgrad_slices(int batch, int M, int N, long n) {
  const long tiles = (long)batch * ((M + 63) / 64) * ((N + 63) / 64);
  long per = 512;
  while (per > OBJ_WGRAD_MINPER && tiles * ((n + per - 1) / per) < OBJ_WGRAD_TARGET) per >>= 1;
  int sk = (int)((n + per - 1) / per);
  if (sk < 2) sk = 2;
  if (sk > 256) sk = 256;
#ifndef OBJ_WGRAD_ATOMICS
  if (sk > OBJ_WGRAD_MAXSLICES) sk = OBJ_WGRAD_MAXSLICES;
#endif
  return sk;
}
struct RedItem {
  const float* part; const float* rs_part; float* C; float* rowsum;
  int M, N, sk, batch;
  long scm, bsc, bsrs;
};
// Optional tail of the step's reduction launch (the one-launch small-batch iteration, objnerf_small_body.h): the block
// after the last item's sums the per-workgroup loss terms in workgroup order and WRITES the status word
// (loss_total_kernel's job), and -- with params set -- every thread that has just summed a gradient element applies
// torch.optim.AdamW to it (adamw_dyn_kernel's arithmetic, objnerf_misc.hip; the flag-dependent skipping and the
// per-group step counters included), so the iteration has no optimiser launch and the gradient is not read back.
struct RedTail {
  const float* loss_part = nullptr; int loss_blocks, K; float* loss_terms; int* status;      // loss_part == NULL: no tail
  float* params = nullptr; const float* grads; float* m; float* v; const int* flags; int* steps; int bank;   // params == NULL: no AdamW
  long p_stride, lo1, lo2, hi2, arena_floats;      // (sums that land outside [grads, grads + arena_floats) are not parameters' gradients)
  double lr, b1, b2, wd; float eps;
};
struct RedGroup {
  static constexpr int MAXG = 16;        // (>= the GEMM group's capacity + head sums + d B: a step's reductions in ONE launch)
  int count;
  int beg[MAXG + 1];      // first block of each item
  RedItem it[MAXG];
  RedTail tail;
};
// one thread per output (m, n) of a batch entry (+ one per row sum): adds the sk slices in slice order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const RedGroup gr) {
  // a block = 64 consecutive outputs x 4 waves; wave w adds slices [w sk / 4, (w + 1) sk / 4) of its outputs in order
  // (coalesced: lanes along the outputs; 16 loads in flight), the four quarter sums meet in LDS and are added in wave
  // order -- a fixed association, so the result is bit-reproducible
  __shared__ float quarter[4][64];
  __shared__ float s_step_size[3], s_bc2_sqrt[3];
  __shared__ int s_active[3];
  const RedTail& tl = gr.tail;
  const bool tail_block = (int)blockIdx.x == gr.beg[gr.count];      // (only launched when the tail is set)
  if (tl.params && threadIdx.x < 3) {
    const int g = threadIdx.x;
    const bool f0 = tl.flags[0] != 0, f1 = tl.flags[1] != 0;
    s_active[g] = g == 0 ? !(f0 && f1) : !f0;
    const int old = tl.steps[3 * tl.bank + g];
    const double st = (double)(old + 1);
    s_step_size[g] = (float)(tl.lr / (1.0 - pow(tl.b1, st)));
    s_bc2_sqrt[g] = (float)sqrt(1.0 - pow(tl.b2, st));
    if (tail_block) tl.steps[3 * (1 - tl.bank) + g] = old + (s_active[g] ? 1 : 0);
  }
  if (tail_block) {
    // the objects' loss terms from the workgroups' partial sums, in workgroup order; status as loss_total_kernel
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int bad = 0;
    // a wave per (object, term): lane l adds partials l, l + 64, .. in order, then the lanes meet in a fixed butterfly --
    // one association for every run (bit-reproducible), 64 loads in flight instead of a chain of `loss_blocks` latencies
    // (1200 workgroups at the benchmark shape: 120 us as one thread per term)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int e = wv; e < 4 * tl.K; e += 4) {
      const float* p = tl.loss_part + (long)(e >> 2) * tl.loss_blocks * 4 + (e & 3);
      float v = 0.f;
      for (int b = lane; b < tl.loss_blocks; b += 64) v += p[4 * b];
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      if (lane == 0) {
        tl.loss_terms[e] = v;
        if (v > 100000.f) bad |= 1;                 // render_rays.py:109-111
        if (!(fabsf(v) <= 3.0e38f)) bad |= 2;       // NaN / Inf
      }
    }
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0 && tl.status) *tl.status = s_bad;
    return;
  }
  int i = 0;
  while (i + 1 < gr.count && (int)blockIdx.x >= gr.beg[i + 1]) ++i;
  const RedItem& r = gr.it[i];
  const long per = (long)r.M * r.N, nrs = r.rs_part ? r.M : 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long e = ((long)blockIdx.x - gr.beg[i]) * 64 + lane;
  const bool live = e < (long)r.batch * (per + nrs);
  const long z = live ? e / (per + nrs) : 0, q = live ? e - z * (per + nrs) : 0;
  const bool is_c = q < per;
  const float* p = is_c ? r.part + z * r.sk * per + q : r.rs_part + z * r.sk * r.M + (q - per);
  const long stride = is_c ? per : r.M;
  const int s0 = (int)((long)r.sk * w / 4), s1 = (int)((long)r.sk * (w + 1) / 4);
  float v = 0.f;
  if (live) {
    int s = s0;
    for (; s + 16 <= s1; s += 16) {
      float t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) t[j] = p[(s + j) * stride];
#pragma unroll
      for (int j = 0; j < 16; ++j) v += t[j];
    }
    for (; s < s1; ++s) v += p[s * stride];
  }
  quarter[w][lane] = v;
  __syncthreads();
  if (w == 0 && live) {
    const float sum = ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
    float* dst;
    if (is_c) {
      const long m = q / r.N, n = q - m * r.N;
      dst = r.C + z * r.bsc + m * r.scm + n;
    } else {
      dst = r.rowsum + z * r.bsrs + (q - per);
    }
    *dst = sum;
    const long idx = dst - tl.grads;                         // position in the gradient arena = in params / moments
    if (tl.params && idx >= 0 && idx < tl.arena_floats) {    // (the 512-d head's moment sums go to the workspace)
      const long pi = idx % tl.p_stride;
      const int g = (pi >= tl.lo1 && pi < tl.lo2) ? 1 : ((pi >= tl.lo2 && pi < tl.hi2) ? 2 : 0);
      if (s_active[g]) {
        const float decay = (float)(1.0 - tl.lr * tl.wd), w1 = (float)(1.0 - tl.b1), w2 = (float)(1.0 - tl.b2);
        const float beta2 = (float)tl.b2;
        float p = tl.params[idx] * decay;
        const float mo = tl.m[idx];
        const float mn = mo + w1 * (sum - mo);
        const float vn = tl.v[idx] * beta2 + (w2 * sum) * sum;
        const float denom = sqrtf(vn) / s_bc2_sqrt[g] + tl.eps;
        p = p + (-s_step_size[g]) * (mn / denom);
        tl.params[idx] = p;
        tl.m[idx] = mn;
        tl.v[idx] = vn;
      }
    }
  }
}
static int red_blocks(const RedItem& r) {
  return (int)(((long)r.batch * ((long)r.M * r.N + (r.rs_part ? r.M : 0)) + 63) / 64);
}
static void red_append(RedGroup& rg, const RedItem& r) {
  if (rg.count == 0) rg.beg[0] = 0;
  rg.it[rg.count] = r;
  rg.beg[rg.count + 1] = rg.beg[rg.count] + red_blocks(r);
  ++rg.count;
}
static void launch_reductions(hipStream_t st, RedGroup& rg) {
  const int tail = rg.tail.loss_part ? 1 : 0;
  if (rg.count) hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)(rg.beg[rg.count] + tail)), dim3(256), 0, st, rg);
  rg.count = 0;
}
// the layer GEMMs the resident-panel kernel takes (16-bit modes, rows k-contiguous, contraction <= 352, N <= 256)
static bool panel_ok(const GemmEnv& E, int M, int N, int Kd, long sak, int splitk, bool rowsum, int nz) {
  return OBJ_G16_AFULL && E.operands && sak == 1 && Kd <= 352 && N <= 256 && splitk <= 1 && M >= 4096 && !rowsum &&
         E.packb && nz <= E.packb_entries;
}

static void gemm(GemmEnv& E, hipStream_t st, int batch, int M, int N, int Kd, const float* A, long sam, long sak, long bsa,
                 const float* B, long sbk, long sbn, long bsb, float* C, long scm, long scn, long bsc,
                 bool accumulate = false, const float* bias = nullptr, long bsbias = 0, bool relu = false,
                 const float* mask = nullptr, long smm = 0, long smn = 0, long bsm = 0, int splitk = 1,
                 float* rowsum = nullptr, long bsrs = 0) {
  Gemm g;
  g.M = M; g.N = N; g.Kd = Kd;
  g.A = A; g.sam = sam; g.sak = sak; g.bsa = bsa;
  g.B = B; g.sbk = sbk; g.sbn = sbn; g.bsb = bsb;
  g.C = C; g.scm = scm; g.scn = scn; g.bsc = bsc;
  g.bias = bias; g.bsbias = bsbias; g.mask = mask; g.smm = smm; g.smn = smn; g.bsm = bsm;
  g.accumulate = accumulate; g.relu = relu; g.splitk = splitk;
  g.rowsum = rowsum; g.bsrs = bsrs;
  g.biasrow = E.biasrow; g.bsbr = E.bsbr; g.sbr = E.sbr;
  g.vec4 = 0; g.Bp = nullptr; g.A2 = nullptr; g.sam2 = g.bsa2 = 0; g.k2 = Kd;
  g.a16 = E.in_act16(A); g.b16 = E.in_act16(B); g.m16 = mask && E.in_act16(mask); g.c16 = E.in_act16(C);
  g.a_scale = E.operands == 2 ? E.a_scale : 1.0f;
  g.part = E.next_part; g.rs_part = E.next_rs_part;
  E.next_part = E.next_rs_part = nullptr;
  const int nz = batch * (splitk > 1 ? splitk : 1);
  if (E.group && !E.operands && !(M >= 256 && N >= 192) && E.group->count < GemmGroup::MAXG) {
    GemmGroup& gr = *E.group;               // collected; launched by flush_group(E, )
    if (E.group16) {
      // gemm_group16_kernel stages both operands k-major: rows contiguous in memory (weight gradients are)
      if (!(sak != 1 && sam == 1 && sbk != 1 && sbn == 1)) { E.error = true; return; }      // (needs k-major operands)
      g.vec4 = (sak % 4 == 0 && bsa % 4 == 0 && M % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0;
    }
    if (gr.count == 0) gr.zbeg[0] = 0;
    gr.g[gr.count] = g;
    gr.zbeg[gr.count + 1] = gr.zbeg[gr.count] + nz;
    ++gr.count;
    return;
  }
  if (E.operands) {
    // operands whose rows are the contiguous dimension are staged k-major (exec is full at the transposing reads:
    // out-of-range elements are zero-filled, never masked)
    const bool akm = sak != 1 && sam == 1, bkm = sbk != 1 && sbn == 1;
    // (split-K weight gradients with a narrow output -- the x1 / x2 columns of the concatenated layers, N = 87 / 42 --
    // take the 128-row tile too: the streamed d_out^T operand is then read once instead of once per 64-row tile)
    const bool wide = M >= 256 && (N >= 192 || (OBJ_G16_WIDE_NARROW && akm && splitk > 1 && N >= 40)), f16 = E.operands == 2;
    const dim3 grid(wide ? (N + 127) / 128 : (N + 63) / 64, wide ? (M + 127) / 128 : (M + 63) / 64, nz);
    // layer GEMMs over the sample axis: resident A panel (64 rows x the whole contraction), 64 x 256 tiles
    if (panel_ok(E, M, N, Kd, sak, splitk, rowsum != nullptr, nz) && (wide || (OBJ_G16_AFULL_NARROW && Kd >= 128))) {
      const int ka = E.a2.A2 ? E.a2.k2 : Kd;
      g.A2 = E.a2.A2 ? E.a2.A2 : A; g.sam2 = E.a2.sam2; g.bsa2 = E.a2.bsa2; g.k2 = ka;
      E.a2.A2 = nullptr;
      g.vec4 = (ka % 8 == 0 && sam % 8 == 0 && bsa % 8 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0;
      if ((g.a16 && !g.vec4) || (g.c16 && g.b16)) { E.error = true; return; }   // (16-bit panel: aligned rows, one 16-bit side)
      g.Bp = E.packb;
      const dim3 pgrid(1, (M + 63) / 64, nz);
#define OBJ_G16_PANEL(OT_, KM_)                                                                                         \
      do {                                                                                                              \
        hipLaunchKernelGGL((pack_b_kernel<OT_, KM_ / 32>), dim3(KM_ / 8, nz), dim3(256), 0, st, g, (OT_*)E.packb);      \
        hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, OT_, 32, 2, 4, false, false, true, KM_>), pgrid, dim3(512), 0, st, g); \
      } while (0)
      if (Kd <= 256) { if (f16) OBJ_G16_PANEL(_Float16, 256); else OBJ_G16_PANEL(__bf16, 256); }
      else { if (f16) OBJ_G16_PANEL(_Float16, 352); else OBJ_G16_PANEL(__bf16, 352); }
#undef OBJ_G16_PANEL
      return;
    }
    E.a2.A2 = nullptr;
    g.vec4 = (akm && sak % 4 == 0 && bsa % 4 == 0 && M % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0;
    if ((g.a16 && !(akm && g.vec4)) || g.c16 || (g.b16 && !(bkm && sbk % 4 == 0 && N % 4 == 0 && bsb % 4 == 0))) {
      E.error = true;                 // (16-bit activations in a GEMM shape that does not take them)
      return;
    }
#define OBJ_G16_LAUNCH(OT_, AK_, BK_)                                                                                   \
    do {                                                                                                                \
      if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, OT_, OBJ_G16_BK_WIDE, 4, 2, AK_, BK_>), grid, dim3(512), 0, st, g); \
      else hipLaunchKernelGGL((gemm_bf16_kernel<2, 2, OT_, OBJ_G16_BK, 2, 2, AK_, BK_>), grid, dim3(256), 0, st, g);    \
    } while (0)
#define OBJ_G16_LAUNCH_T(OT_)                                                                                           \
    do {                                                                                                                \
      if (akm && bkm) OBJ_G16_LAUNCH(OT_, true, true);                                                                  \
      else if (bkm) OBJ_G16_LAUNCH(OT_, false, true);                                                                   \
      else if (akm) OBJ_G16_LAUNCH(OT_, true, false);                                                                   \
      else OBJ_G16_LAUNCH(OT_, false, false);                                                                           \
    } while (0)
    if (f16) OBJ_G16_LAUNCH_T(_Float16); else OBJ_G16_LAUNCH_T(__bf16);
#undef OBJ_G16_LAUNCH_T
#undef OBJ_G16_LAUNCH
    return;
  }
  if (M >= 256 && N >= 192) {         // wide layer GEMMs (hidden 256): 128 x 128 tiles on 8 waves (3.5 % faster than the
                                      // same tile on 4 waves with 64 accumulator registers each).  Up to N = 128 the
                                      // 64 x 64 tiles win (measured, hidden 128): 4x the workgroups, 16 instead
                                      // of 64 accumulator registers -> occupancy hides the operand latency
    dim3 grid((N + 127) / 128, (M + 64 * OBJ_GEMM_WIDE_TM - 1) / (64 * OBJ_GEMM_WIDE_TM), nz);
    hipLaunchKernelGGL(gemm_kernel8, grid, dim3(512), 0, st, g);
  } else {
    dim3 grid((N + 63) / 64, (M + 63) / 64, nz);
    // (64-deep k stages for small grids were measured: slower -- the transposed LDS store conflicts dominate)
    hipLaunchKernelGGL((gemm_kernel<2, 2>), grid, dim3(256), 0, st, g);
  }
}

// weight gradient: C[M][N] += sum over the n samples; few output tiles, long contraction -> split-K + atomics.
// bias_grad (optional, pre-zeroed like C): [M] column sums of the output-gradient operand, same pass.
static void wgrad(GemmEnv& E, hipStream_t st, int batch, int M, int N, long n, const float* A, long sam, long sak, long bsa,
                  const float* B, long sbk, long sbn, long bsb, float* C, long scm, long bsc,
                  float* bias_grad = nullptr) {
  int sk = wgrad_slices(batch, M, N, n);
  if (E.max_slices > 0 && sk > E.max_slices) sk = E.max_slices;
#ifndef OBJ_WGRAD_ATOMICS
  // deterministic form: every slice stores its partial tile; an ordered reduction follows
  const int skd = sk;
  float* part = E.parts_alloc((size_t)batch * skd * M * N);
  float* rs_part = (part && bias_grad) ? E.parts_alloc((size_t)batch * skd * M) : nullptr;
  if (part && (!bias_grad || rs_part)) {
    E.next_part = part; E.next_rs_part = rs_part;
    gemm(E, st, batch, M, N, (int)n, A, sam, sak, bsa, B, sbk, sbn, bsb, C, scm, 1, bsc, false, nullptr, 0, false, nullptr, 0,
         0, 0, skd, bias_grad, bsc);
    RedItem r;
    r.part = part; r.rs_part = rs_part; r.C = C; r.rowsum = bias_grad;
    r.M = M; r.N = N; r.sk = skd; r.batch = batch; r.scm = scm; r.bsc = bsc; r.bsrs = bsc;
    if (E.red_group && E.red_group->count < RedGroup::MAXG) {
      red_append(*E.red_group, r);
    } else {
      RedGroup rg;
      rg.count = 1; rg.beg[0] = 0; rg.beg[1] = red_blocks(r); rg.it[0] = r;
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)rg.beg[1]), dim3(256), 0, st, rg);
    }
    return;
  }
#endif
  if (E.need_parts) { E.parts_failed = true; return; }
  gemm(E, st, batch, M, N, (int)n, A, sam, sak, bsa, B, sbk, sbn, bsb, C, scm, 1, bsc, false, nullptr, 0, false, nullptr, 0,
       0, 0, sk, bias_grad, bsc);
}

// ------------------------------------------------------------------------------------------------
// Small-batch forward for hidden 128 (the background network at the reference's native batch, 1200 rays x 14
// samples): the layer-by-layer GEMMs above are then a chain of dependent ~15 us launches for ~1 us of MFMA work
// each.  Here ONE launch runs the whole network: a workgroup owns 16 RT samples (RT chosen so that the grid fills
// the chip once), keeps their activations in two LDS buffers and streams each layer's weight matrix from L2 into
// a third.  Every activation is also written to HBM, as the backward pass and the weight-gradient GEMMs read them.
// Same arithmetic as the GEMM path (fp32 MFMA, k ascending); the concatenated layers accumulate their two parts
// in one accumulator.  LDS rows have pitch 132 floats: conflict-free operand reads (bank = 4 m + k).
struct FwdSmall {
  long n; int feat;
  const float* params; long ps;
  const float* emb;                    // [K][n][OBJ_EMB]
  float *h1, *h2, *h3, *h4, *hc, *hf;  // [K][n][128]
  float *alpha, *color;                // [K][n], [K][n][3]
  int o_in_w, o_in_b, o_m1_w, o_m1_b, o_cat_w, o_cat_b, o_m2_w, o_m2_b, o_a_w, o_a_b, o_cl_w, o_cl_b, o_oc_w, o_oc_b,
      o_fl_w, o_fl_b;
};
constexpr int FS_H = 128, FS_P = 132;
template <int RT> constexpr size_t fs_lds_bytes() { return (size_t)(FS_H + 2 * 16 * RT) * FS_P * sizeof(float); }

// BF: OBJNERF_TRAIN_BF16 -- the same kernel with the contraction on v_mfma_f32_16x16x32_bf16: operands stay fp32 in
// LDS, a lane reads its 8 consecutive k of a 32-block (two ds_read_b128) and rounds them to bf16 on the fly; one MFMA
// replaces eight (the fp32 kernel spends ~60 % of its time in the MFMA loop).
__device__ __forceinline__ bf16x8 cvt_bf16x8(const f32x4& lo, const f32x4& hi) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
  return o;
}
template <int RT, bool BF>
__global__ __launch_bounds__(512) void mlp_fwd_small_kernel(const FwdSmall a) {
  constexpr int BM = 16 * RT, H = FS_H, PT = FS_P;
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  float* Wb = fs_lds;                   // [128][132] current layer's weights, [out][in]
  float* Xa = Wb + H * PT;              // [BM][132]
  float* Xb = Xa + BM * PT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;     // 8 waves: wave w owns features 16 w .. 16 w + 15
  const int c = lane & 15, gg = lane >> 4;
  const long z = blockIdx.y, n = a.n;
  const long m0 = (long)blockIdx.x * BM;
  const float* P = a.params + z * a.ps;
  const float* emb = a.emb + z * n * OBJ_EMB;
  const int kk = tid & 127, rg = tid >> 7;          // staging: lanes along k (the contiguous source dimension)
  // The workgroup's slice of the embedding (x1 = columns [0, E1), x2 = [E1, E1 + E2)) is fetched ONCE into
  // registers at the start -- its HBM latency hides behind the first weight fetch -- and written to an LDS buffer
  // where a layer needs it (x1 twice): a global load at that point would sit exposed on the critical path.
  constexpr int ER = (BM + 3) / 4;
  float er1[ER], er2[ER];
#pragma unroll
  for (int i = 0; i < ER; ++i) {
    const int m = rg + 4 * i;
    const bool in = m < BM && m0 + m < n;
    er1[i] = (in && kk < OBJ_E1) ? emb[(m0 + m) * OBJ_EMB + kk] : 0.f;
    er2[i] = (in && kk < OBJ_E2) ? emb[(m0 + m) * OBJ_EMB + OBJ_E1 + kk] : 0.f;
  }
  auto put_emb = [&](float* X, const float (&er)[ER], const int KC) {      // zero-padded to a multiple of 16 (32) columns
    const int KC16 = BF ? ((KC + 31) & ~31) : ((KC + 15) & ~15);
    if (kk < KC16) {
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const int m = rg + 4 * i;
        if (m < BM) X[m * PT + kk] = er[i];
      }
    }
  };
  // The next layer's weights W[out][ld], columns [0, KC), are fetched into registers BEFORE the current layer's MFMA
  // loop and stored to Wb after it: their L2 latency hides behind the matrix work.
  float wr[32];
  auto fetch_w = [&](const float* W, const int ld, const int KC) {
#pragma unroll
    for (int i = 0; i < 32; ++i) wr[i] = kk < KC ? W[(rg + 4 * i) * ld + kk] : 0.f;
  };
  auto put_w = [&]() {
#pragma unroll
    for (int i = 0; i < 32; ++i) Wb[(rg + 4 * i) * PT + kk] = wr[i];
  };
  f32x4 acc[RT];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // acc += X[:, :KC] Wb^T for this wave's column tile, all RT row tiles (KC a multiple of 16).  The contraction index
  // is free to permute: MFMA step j of a 16-block takes k = 16 blk + 4 gg + j from lane group gg, so a lane fetches
  // four steps of an operand with ONE ds_read_b128 (conflict-free with the 132-float pitch); the next block's
  // operands are requested before the current block's 4 RT MFMAs are issued.
  auto mma = [&](const float* X, const int KC) {
    if (BF) {          // KC a multiple of 32 (the callers round up; the padding columns hold zeros on both sides)
      const float* bp = Wb + (16 * w + c) * PT + 8 * gg;
      const float* ap = X + c * PT + 8 * gg;
      for (int kb = 0; kb < KC; kb += 32) {
        const bf16x8 b = cvt_bf16x8(*reinterpret_cast<const f32x4*>(bp + kb), *reinterpret_cast<const f32x4*>(bp + kb + 4));
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const float* r = ap + 16 * i * PT + kb;
          const bf16x8 av = cvt_bf16x8(*reinterpret_cast<const f32x4*>(r), *reinterpret_cast<const f32x4*>(r + 4));
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acc[i], 0, 0, 0);
        }
      }
      return;
    }
    const float* bp = Wb + (16 * w + c) * PT + 4 * gg;
    const float* ap = X + c * PT + 4 * gg;
    f32x4 bc = *reinterpret_cast<const f32x4*>(bp), ac[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) ac[i] = *reinterpret_cast<const f32x4*>(ap + 16 * i * PT);
    for (int kb = 0; kb < KC; kb += 16) {
      f32x4 bn = bc, an[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) an[i] = ac[i];
      if (kb + 16 < KC) {
        bn = *reinterpret_cast<const f32x4*>(bp + kb + 16);
#pragma unroll
        for (int i = 0; i < RT; ++i) an[i] = *reinterpret_cast<const f32x4*>(ap + 16 * i * PT + kb + 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[i][j], bc[j], acc[i], 0, 0, 0);
      bc = bn;
#pragma unroll
      for (int i = 0; i < RT; ++i) ac[i] = an[i];
    }
  };
  // relu(acc + bias) -> LDS buffer (next layer's input) and the HBM activation
  auto store = [&](float* X, float* hbm, const float bv) {
    float* out = hbm + z * n * H;
    const int f = 16 * w + c;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        const float v = fmaxf(acc[i][r] + bv, 0.f);
        X[m * PT + f] = v;
        if (m0 + m < n) out[(m0 + m) * H + f] = v;
      }
  };
  constexpr int E1P = BF ? ((OBJ_E1 + 31) & ~31) : ((OBJ_E1 + 15) & ~15), E2P = BF ? ((OBJ_E2 + 31) & ~31) : ((OBJ_E2 + 15) & ~15);
  // biases and head weights of this lane, fetched before anything waits on them
  const int fb = 16 * w + c;
  const float b_in = P[a.o_in_b + fb], b_m1 = P[a.o_m1_b + fb], b_cat = P[a.o_cat_b + fb], b_m2 = P[a.o_m2_b + fb];
  const float b_cl = P[a.o_cl_b + fb], b_fl = a.feat ? P[a.o_fl_b + fb] : 0.f;
  const float wa0 = P[a.o_a_w + lane], wa1 = P[a.o_a_w + lane + 64];
  const float w00 = P[a.o_oc_w + lane], w01 = P[a.o_oc_w + lane + 64];
  const float w10 = P[a.o_oc_w + H + lane], w11 = P[a.o_oc_w + H + lane + 64];
  const float w20 = P[a.o_oc_w + 2 * H + lane], w21 = P[a.o_oc_w + 2 * H + lane + 64];
  const float ba = P[a.o_a_b], bc0 = P[a.o_oc_b], bc1 = P[a.o_oc_b + 1], bc2 = P[a.o_oc_b + 2];
  // ---- h1 = relu(x1 W_in^T + b)                                  (model.py:63-66)
  fetch_w(P + a.o_in_w, OBJ_E1, OBJ_E1);
  put_emb(Xa, er1, OBJ_E1);
  put_w();
  __syncthreads();
  fetch_w(P + a.o_m1_w, H, H);
  zero();
  mma(Xa, E1P);
  __syncthreads();
  store(Xb, a.h1, b_in);
  put_w();
  __syncthreads();
  // ---- h2
  fetch_w(P + a.o_cat_w, H + OBJ_E1, H);
  zero();
  mma(Xb, H);
  __syncthreads();
  store(Xa, a.h2, b_m1);
  put_w();
  put_emb(Xb, er1, OBJ_E1);
  __syncthreads();
  // ---- h3 = relu([h2 | x1] W_cat^T + b)
  fetch_w(P + a.o_cat_w + H, H + OBJ_E1, OBJ_E1);
  zero();
  mma(Xa, H);
  __syncthreads();
  put_w();
  __syncthreads();
  fetch_w(P + a.o_m2_w, H, H);
  mma(Xb, E1P);
  __syncthreads();
  store(Xa, a.h3, b_cat);
  put_w();
  __syncthreads();
  // ---- h4
  fetch_w(P + a.o_cl_w, H + OBJ_E2, H);
  zero();
  mma(Xa, H);
  __syncthreads();
  store(Xb, a.h4, b_m2);
  put_w();
  put_emb(Xa, er2, OBJ_E2);
  __syncthreads();
  // ---- hc = relu([h4 | x2] W_cl^T + b)
  fetch_w(P + a.o_cl_w + H, H + OBJ_E2, OBJ_E2);
  zero();
  mma(Xb, H);
  __syncthreads();
  put_w();
  __syncthreads();
  if (a.feat) fetch_w(P + a.o_fl_w, H + OBJ_E2, H);
  mma(Xa, E2P);
  __syncthreads();
  // hc goes to a THIRD place: Wb is free until the next put_w and Xa (x2) is needed again by the feature layer
  float* Xc = Wb;
  store(Xc, a.hc, b_cl);
  __syncthreads();
  // ---- heads: alpha = 10 (h4 . wa + ba), colour = sigmoid(hc Woc^T + boc)      (model.py:81-96); a wave per row
  {
    for (int m = w; m < BM; m += 8) {
      const float x0 = Xb[m * PT + lane], x1 = Xb[m * PT + lane + 64];
      const float y0 = Xc[m * PT + lane], y1 = Xc[m * PT + lane + 64];
      const float sa = wave_sum64(fmaf(wa1, x1, wa0 * x0));
      const float s0 = wave_sum64(fmaf(w01, y1, w00 * y0));
      const float s1 = wave_sum64(fmaf(w11, y1, w10 * y0));
      const float s2 = wave_sum64(fmaf(w21, y1, w20 * y0));
      if (lane == 0 && m0 + m < n) {
        const long i = z * n + m0 + m;
        a.alpha[i] = (sa + ba) * 10.0f;
        a.color[i * 3] = sigmoid_acc(s0 + bc0);
        a.color[i * 3 + 1] = sigmoid_acc(s1 + bc1);
        a.color[i * 3 + 2] = sigmoid_acc(s2 + bc2);
      }
    }
  }
  if (!a.feat) return;
  // ---- hf = relu([h4 | x2] W_fl^T + b)     (feature layer, same inputs as the colour layer: Xb = h4, Xa = x2)
  __syncthreads();
  put_w();
  __syncthreads();
  fetch_w(P + a.o_fl_w + H, H + OBJ_E2, OBJ_E2);
  zero();
  mma(Xb, H);
  __syncthreads();
  put_w();
  __syncthreads();
  mma(Xa, E2P);
  float* out = a.hf + z * n * H;
  {
    const int f = 16 * w + c;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        if (m0 + m < n) out[(m0 + m) * H + f] = fmaxf(acc[i][r] + b_fl, 0.f);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// The matching backward (input-gradient chain) in one launch: d_hc (+ d_hf) -> d_h4 -> d_h3 -> d_h2 -> d_h1 and the
// embedding gradient, every d_h also written to HBM for the weight-gradient GEMMs.  d_in = d_out W is the same MFMA
// loop with the weight rows read as they lie in memory: Wb[k = out][n = in], pitch 130 (bank = 2 k + n).
struct BwdSmall {
  long n; int feat;
  const float* params; long ps;
  const float *h1, *h2, *h3, *h4;      // relu masks
  const float *d_hc, *d_hf;            // [K][n][128] (d_hf: feature layer, or NULL)
  float *d_h4, *d_h3, *d_h2, *d_h1;    // d_h4 arrives holding the alpha head's contribution (heads_bwd_kernel)
  float* d_emb;                        // [K][n][OBJ_EMB]
  int o_in_w, o_m1_w, o_cat_w, o_m2_w, o_cl_w, o_fl_w;
};
constexpr int BS_PW = 130;
template <int RT> constexpr size_t bs_lds_bytes() { return (size_t)(FS_H * BS_PW + 2 * 16 * RT * FS_P) * sizeof(float); }

template <int RT, bool BF>
__global__ __launch_bounds__(512) void mlp_bwd_small_kernel(const BwdSmall a) {
  constexpr int BM = 16 * RT, H = FS_H, PT = FS_P, PW = BS_PW;
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  float* Wb = fs_lds;                   // [k = out][n = in], pitch 130
  float* Da = Wb + H * PW;              // [BM][132]
  float* Db = Da + BM * PT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;     // wave w owns input features 16 w .. 16 w + 15
  const int c = lane & 15, gg = lane >> 4;
  const long z = blockIdx.y, n = a.n;
  const long m0 = (long)blockIdx.x * BM;
  const float* P = a.params + z * a.ps;
  const int kk = tid & 127, rg = tid >> 7;
  auto load_d = [&](float* X, const float* src) {     // rows [m0, m0 + BM) of a [n][128] tensor
    const float* sp = src + z * n * H;
#pragma unroll
    for (int i = 0; i < (BM + 3) / 4; ++i) {
      const int m = rg + 4 * i;
      if (m < BM) X[m * PT + kk] = (m0 + m < n) ? sp[(m0 + m) * H + kk] : 0.f;
    }
  };
  // W[out][ld], columns [col0, col0 + NC): row `out` -> Wb[out][0 .. NC) (zero up to 128), fetched ahead into registers
  float wr[32];
  auto fetch_w = [&](const float* W, const int ld, const int NC) {
#pragma unroll
    for (int i = 0; i < 32; ++i) wr[i] = kk < NC ? W[(rg + 4 * i) * ld + kk] : 0.f;
  };
  // bf16 mode: the weight rows go to LDS ROUNDED, [k = out][n = in] with a 288-byte pitch (72 dwords = 8 mod 64: the
  // eight rows a transposing read's half-wave touches sit on disjoint bank octets)
  constexpr int PW16 = 288;
  auto put_w = [&]() {
    if (BF) {
      __bf16* W16 = reinterpret_cast<__bf16*>(Wb);
#pragma unroll
      for (int i = 0; i < 32; ++i) W16[(rg + 4 * i) * (PW16 / 2) + kk] = (__bf16)wr[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) Wb[(rg + 4 * i) * PW + kk] = wr[i];
  };
  f32x4 acc[RT], accE[RT];
  auto zero = [&](f32x4 (&v)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // v += D (all 128 output-feature columns) x Wb -> this wave's 16 input features.  Same k permutation as the
  // forward kernel (step j of a 16-block takes k = 16 blk + 4 gg + j): D rows by ds_read_b128, the four weight rows
  // by ds_read_b32 (bank = 8 gg + 2 j + n: two lanes per bank, the minimum for 64 lanes).
  auto mma = [&](const float* D, f32x4 (&v)[RT]) {
    if (BF) {
      // bf16 MFMA: the lane's 8 k-slots of a 32-block are output features 4 gg .. 4 gg + 3 and 16 + 4 gg .. + 3.  The
      // weight operand -- 8 k of ONE input column -- comes out of the row-major bf16 image by two transposing reads
      // (lane 4 q + p of a 16-lane group supplies row q, 8-byte chunk p); it used to be eight ds_read_b32 of fp32 rows
      // and four conversions per MFMA.
      const char* bp = reinterpret_cast<const char*>(Wb) + (4 * gg + (c >> 2)) * PW16 + 32 * w + 8 * (c & 3);
      const float* ap = D + c * PT + 4 * gg;
#pragma unroll
      for (int kb = 0; kb < H; kb += 32) {
        const uint2 b0 = lds_tr16(bp + kb * PW16), b1 = lds_tr16(bp + (kb + 16) * PW16);
        const bf16x8 b = __builtin_bit_cast(bf16x8, uint4{b0.x, b0.y, b1.x, b1.y});
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const float* r = ap + 16 * i * PT + kb;
          const bf16x8 av = cvt_bf16x8(*reinterpret_cast<const f32x4*>(r), *reinterpret_cast<const f32x4*>(r + 16));
          v[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, v[i], 0, 0, 0);
        }
      }
      return;
    }
    const float* bp = Wb + 4 * gg * PW + 16 * w + c;
    const float* ap = D + c * PT + 4 * gg;
    f32x4 bc, ac[RT];
#pragma unroll
    for (int j = 0; j < 4; ++j) bc[j] = bp[j * PW];
#pragma unroll
    for (int i = 0; i < RT; ++i) ac[i] = *reinterpret_cast<const f32x4*>(ap + 16 * i * PT);
    for (int kb = 0; kb < H; kb += 16) {
      f32x4 bn = bc, an[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) an[i] = ac[i];
      if (kb + 16 < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = bp[(kb + 16 + j) * PW];
#pragma unroll
        for (int i = 0; i < RT; ++i) an[i] = *reinterpret_cast<const f32x4*>(ap + 16 * i * PT + kb + 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i) v[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[i][j], bc[j], v[i], 0, 0, 0);
      bc = bn;
#pragma unroll
      for (int i = 0; i < RT; ++i) ac[i] = an[i];
    }
  };
  // masked d_h: (acc [+ prior]) where act > 0 -> LDS (next layer's operand) and HBM
  float mk[RT][4];
  auto fetch_mask = [&](const float* act) {
    const float* hp = act + z * n * H;
    const int f = 16 * w + c;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        mk[i][r] = (m0 + m < n) ? hp[(m0 + m) * H + f] : 0.f;
      }
  };
  float pr[RT][4];                         // d_h4 arrives holding the alpha head's part: fetched with the mask
  auto fetch_prior = [&](const float* src) {
    const float* sp = src + z * n * H;
    const int f = 16 * w + c;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        pr[i][r] = (m0 + m < n) ? sp[(m0 + m) * H + f] : 0.f;
      }
  };
  auto store_dh = [&](float* X, float* hbm, const bool add_prior) {
    float* out = hbm + z * n * H;
    const int f = 16 * w + c;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        const bool in = m0 + m < n;
        float v = acc[i][r];
        if (add_prior) v += pr[i][r];
        v = mk[i][r] > 0.f ? v : 0.f;
        X[m * PT + f] = v;
        if (in) out[(m0 + m) * H + f] = v;
      }
  };
  auto store_emb = [&](const int col0, const int NC) {     // accE -> d_emb[:, col0 + f], f < NC
    float* out = a.d_emb + z * n * OBJ_EMB;
    const int f = 16 * w + c;
    if (f < NC) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * gg + r;
          if (m0 + m < n) out[(m0 + m) * OBJ_EMB + col0 + f] = accE[i][r];
        }
    }
  };
  const bool e1_wave = 16 * w < OBJ_E1, e2_wave = 16 * w < OBJ_E2;      // waves that own embedding columns
  zero(acc);
  zero(accE);
  if (a.feat) {
    // ---- feature layer: d_h4 += d_hf W_fl[:, :H], d_x2 = d_hf W_fl[:, H:]
    fetch_w(P + a.o_fl_w, H + OBJ_E2, H);
    load_d(Da, a.d_hf);
    put_w();
    __syncthreads();
    fetch_w(P + a.o_fl_w + H, H + OBJ_E2, OBJ_E2);
    mma(Da, acc);
    __syncthreads();
    put_w();
    __syncthreads();
    if (e2_wave) mma(Da, accE);
    __syncthreads();
  }
  // ---- colour layer: d_h4 = relu'(h4) (alpha-head part + [d_hf W_fl1] + d_hc W_cl[:, :H]), d_x2 (+)= d_hc W_cl[:, H:]
  fetch_w(P + a.o_cl_w, H + OBJ_E2, H);
  load_d(Da, a.d_hc);
  put_w();
  __syncthreads();
  fetch_w(P + a.o_cl_w + H, H + OBJ_E2, OBJ_E2);
  fetch_mask(a.h4);
  fetch_prior(a.d_h4);
  mma(Da, acc);
  __syncthreads();
  store_dh(Db, a.d_h4, true);                    // Db = d_h4
  put_w();
  __syncthreads();
  fetch_w(P + a.o_m2_w, H, H);
  if (e2_wave) mma(Da, accE);
  store_emb(OBJ_E1, OBJ_E2);
  __syncthreads();
  // ---- mid2: d_h3 = relu'(h3) (d_h4 W_m2)
  put_w();
  __syncthreads();
  fetch_w(P + a.o_cat_w, H + OBJ_E1, H);
  fetch_mask(a.h3);
  zero(acc);
  mma(Db, acc);
  __syncthreads();
  store_dh(Da, a.d_h3, false);                   // Da = d_h3
  put_w();
  __syncthreads();
  // ---- cat layer: d_h2 = relu'(h2) (d_h3 W_cat[:, :H]), d_x1 = d_h3 W_cat[:, H:]
  fetch_w(P + a.o_cat_w + H, H + OBJ_E1, OBJ_E1);
  fetch_mask(a.h2);
  zero(acc);
  mma(Da, acc);
  __syncthreads();
  store_dh(Db, a.d_h2, false);                   // Db = d_h2
  put_w();
  __syncthreads();
  fetch_w(P + a.o_m1_w, H, H);
  zero(accE);
  if (e1_wave) mma(Da, accE);
  __syncthreads();
  // ---- mid1: d_h1 = relu'(h1) (d_h2 W_m1)
  put_w();
  __syncthreads();
  fetch_w(P + a.o_in_w, OBJ_E1, OBJ_E1);
  fetch_mask(a.h1);
  zero(acc);
  mma(Db, acc);
  __syncthreads();
  store_dh(Da, a.d_h1, false);                   // Da = d_h1
  put_w();
  __syncthreads();
  // ---- in layer: d_x1 += d_h1 W_in
  if (e1_wave) mma(Da, accE);
  store_emb(0, OBJ_E1);
}

template <int RT>
static void launch_bwd_small(hipStream_t st, const BwdSmall& f, int K, bool bf) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)mlp_bwd_small_kernel<RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bs_lds_bytes<RT>());
    (void)hipFuncSetAttribute((const void*)mlp_bwd_small_kernel<RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bs_lds_bytes<RT>());
  });
  dim3 grid((unsigned)((f.n + 16 * RT - 1) / (16 * RT)), (unsigned)K);
  if (bf) hipLaunchKernelGGL((mlp_bwd_small_kernel<RT, true>), grid, dim3(512), bs_lds_bytes<RT>(), st, f);
  else hipLaunchKernelGGL((mlp_bwd_small_kernel<RT, false>), grid, dim3(512), bs_lds_bytes<RT>(), st, f);
}

template <int RT>
static void launch_fwd_small(hipStream_t st, const FwdSmall& f, int K, bool bf) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)mlp_fwd_small_kernel<RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)fs_lds_bytes<RT>());
    (void)hipFuncSetAttribute((const void*)mlp_fwd_small_kernel<RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)fs_lds_bytes<RT>());
  });
  dim3 grid((unsigned)((f.n + 16 * RT - 1) / (16 * RT)), (unsigned)K);
  if (bf) hipLaunchKernelGGL((mlp_fwd_small_kernel<RT, true>), grid, dim3(512), fs_lds_bytes<RT>(), st, f);
  else hipLaunchKernelGGL((mlp_fwd_small_kernel<RT, false>), grid, dim3(512), fs_lds_bytes<RT>(), st, f);
}

#include "objnerf_small_body.h"

// row tiles per workgroup: the smallest RT for which K * ceil(n / (16 RT)) workgroups fit the chip in one round;
// 0 = not a small batch
static int small_batch_rt(const GemmEnv& E, int H, long n, int K) {
  if (H != FS_H) return 0;
  for (int rt = 1; rt <= 5; ++rt)
    if ((long)K * ((n + 16 * rt - 1) / (16 * rt)) <= 256) return rt;
  // up to four rounds of 80-sample workgroups the one-launch kernels still win (76 800 samples: 1.02 vs 1.07 ms per
  // step, 38 400: 0.53 vs 0.65); beyond that the GEMM path runs near the MFMA peak
  // (fp32 and bf16 modes: the one-launch kernels exist for both; fp16 keeps the GEMM path from two rounds on)
#ifdef OBJ_NO_SMALL_BF16        // diagnostic: the bf16 mode on the GEMM path from two rounds on (tools/bg_ab.py)
  if (E.operands == 0 && (long)K * ((n + 79) / 80) <= 1024) return 5;
#else
  if (E.operands != 2 && (long)K * ((n + 79) / 80) <= 1024) return 5;
#endif
  return 0;
}

static void flush_group(GemmEnv& E, hipStream_t st, GemmGroup& gr) {
  E.group = nullptr;
  if (gr.count == 0) return;
  int mx = 1, my = 1;
  for (int i = 0; i < gr.count; ++i) {
    mx = std::max(mx, (gr.g[i].N + 63) / 64);
    my = std::max(my, (gr.g[i].M + 63) / 64);
  }
  if (E.group16 && OBJ_GROUP16_WIDE)
    hipLaunchKernelGGL(gemm_group16_kernel, dim3((mx + 1) / 2, (my + 1) / 2, gr.zbeg[gr.count]), dim3(OBJ_GROUP16_WIDE ? 512 : 256), 0, st, gr);
  else if (E.group16) hipLaunchKernelGGL(gemm_group16_kernel, dim3(mx, my, gr.zbeg[gr.count]), dim3(256), 0, st, gr);
  else hipLaunchKernelGGL(gemm_group_kernel, dim3((mx + 1) / 2, (my + 1) / 2, gr.zbeg[gr.count] + gr.task.blocks), dim3(512), 0, st, gr);
  gr.count = 0;
  gr.task.blocks = 0;
  E.group16 = false;
  E.red_group = nullptr;                 // (the caller launches the collected reductions: launch_reductions)
}


// Activation loads of the kernels around the GEMMs: act = 0 fp32 storage, 1 bf16, 2 fp16 (the 16-bit modes keep
// h1 .. hc in the operand type at hidden 256: half the traffic of the HBM-bound layer GEMMs).
__device__ __forceinline__ float act_ld(const float* p, long i, int act) {
  if (act == 0) return p[i];
  if (act == 1) return (float)reinterpret_cast<const __bf16*>(p)[i];
  return (float)reinterpret_cast<const _Float16*>(p)[i];
}
__device__ __forceinline__ float4 act_ld4(const float* p, long i, int act) {       // i % 4 == 0
  if (act == 0) return *reinterpret_cast<const float4*>(p + i);
  const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + i);
  if (act == 1)
    return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                       __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  const h4_t h = __builtin_bit_cast(h4_t, r);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}

__device__ __forceinline__ void act_st4(float* p, long i, float4 v, int act, float scale) {     // i % 4 == 0
  if (act == 0) { *reinterpret_cast<float4*>(p + i) = v; return; }
  uint2 r;
  if (act == 1) {
    typedef __bf16 b4_t __attribute__((ext_vector_type(4)));
    const b4_t b = {(__bf16)(v.x * scale), (__bf16)(v.y * scale), (__bf16)(v.z * scale), (__bf16)(v.w * scale)};
    r = __builtin_bit_cast(uint2, b);
  } else {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    const h4_t h = {Op16<_Float16>::cvt(v.x * scale), Op16<_Float16>::cvt(v.y * scale), Op16<_Float16>::cvt(v.z * scale),
                    Op16<_Float16>::cvt(v.w * scale)};
    r = __builtin_bit_cast(uint2, h);
  }
  *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p) + i) = r;
}

// heads forward: alpha = 10 (h4 . wa + ba), color = sigmoid(hc Woc^T + boc)        (model.py:81-96)
// 16 lanes per sample row (float4 each, coalesced), the four dot products meet by DPP row sums.
__global__ __launch_bounds__(256) void heads_fwd_kernel(int Hh, long n, const float* h4, const float* hc,
                                                        const float* params, long p_stride, int off_wa, int off_ba,
                                                        int off_woc, int off_boc, float* alpha, float* color, int act) {
  extern __shared__ float sw[];            // wa | woc[3]  (4 Hh floats)
  const long z = blockIdx.y;
  const float* P = params + z * p_stride;
  for (int i = threadIdx.x; i < Hh; i += 256) sw[i] = P[off_wa + i];
  for (int i = threadIdx.x; i < 3 * Hh; i += 256) sw[Hh + i] = P[off_woc + i];
  __syncthreads();
  const int l16 = threadIdx.x & 15;
  const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  float sa = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (i < n) {
    const long ro = (z * n + i) * Hh;
    for (int h = 4 * l16; h < Hh; h += 64) {
      const float4 av = act_ld4(h4, ro + h, act);
      const float4 cv = act_ld4(hc, ro + h, act);
      const float* w0 = sw + h;
      sa = fmaf(w0[3], av.w, fmaf(w0[2], av.z, fmaf(w0[1], av.y, fmaf(w0[0], av.x, sa))));
      s0 = fmaf(w0[Hh + 3], cv.w, fmaf(w0[Hh + 2], cv.z, fmaf(w0[Hh + 1], cv.y, fmaf(w0[Hh], cv.x, s0))));
      s1 = fmaf(w0[2 * Hh + 3], cv.w, fmaf(w0[2 * Hh + 2], cv.z, fmaf(w0[2 * Hh + 1], cv.y, fmaf(w0[2 * Hh], cv.x, s1))));
      s2 = fmaf(w0[3 * Hh + 3], cv.w, fmaf(w0[3 * Hh + 2], cv.z, fmaf(w0[3 * Hh + 1], cv.y, fmaf(w0[3 * Hh], cv.x, s2))));
    }
  }
  sa = dpp_rowsum16(sa); s0 = dpp_rowsum16(s0); s1 = dpp_rowsum16(s1); s2 = dpp_rowsum16(s2);
  if (i < n && l16 == 0) {
    alpha[z * n + i] = (sa + P[off_ba]) * 10.0f;
    color[(z * n + i) * 3] = sigmoid_acc(s0 + P[off_boc]);
    color[(z * n + i) * 3 + 1] = sigmoid_acc(s1 + P[off_boc + 1]);
    color[(z * n + i) * 3 + 2] = sigmoid_acc(s2 + P[off_boc + 2]);
  }
}

// heads backward: dhead[n][4] = (10 d_alpha, d_color * color (1 - color)); d_hc = relu'(hc) Woc^T d_craw;
// d_h4 = wa * d_araw  (the colour-layer contribution is accumulated by a GEMM afterwards; d_h4 == NULL: that GEMM adds
// this rank-1 term itself -- bias = wa, per-row factor = dhead[:, 0] -- and the 2 x n x H round trip is saved)
__global__ __launch_bounds__(256) void heads_bwd_kernel(int Hh, long n, const float* hc, const float* color,
                                                        const float* d_alpha, const float* d_color, const float* params,
                                                        long p_stride, int off_wa, int off_woc, float* dhead, float* d_hc,
                                                        float* d_h4, int act, float gscale) {
  extern __shared__ float sw[];            // wa | woc[3]
  const long z = blockIdx.y;
  const float* P = params + z * p_stride;
  for (int i = threadIdx.x; i < Hh; i += 256) sw[i] = P[off_wa + i];
  for (int i = threadIdx.x; i < 3 * Hh; i += 256) sw[Hh + i] = P[off_woc + i];
  __syncthreads();
  const int l16 = threadIdx.x & 15;
  const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= n) return;
  const long o = z * n + i;
  const float da = 10.0f * d_alpha[o];
  float dc[3];
  for (int x = 0; x < 3; ++x) {
    const float cv = color[o * 3 + x];
    dc[x] = d_color[o * 3 + x] * cv * (1.0f - cv);
  }
  if (l16 == 0) *reinterpret_cast<float4*>(dhead + o * 4) = make_float4(da, dc[0], dc[1], dc[2]);
  for (int h = 4 * l16; h < Hh; h += 64) {
    const float4 hv = act_ld4(hc, o * Hh + h, act);
    const float* w0 = sw + h;
    float4 v, u;
    v.x = fmaf(w0[3 * Hh], dc[2], fmaf(w0[2 * Hh], dc[1], w0[Hh] * dc[0]));
    v.y = fmaf(w0[3 * Hh + 1], dc[2], fmaf(w0[2 * Hh + 1], dc[1], w0[Hh + 1] * dc[0]));
    v.z = fmaf(w0[3 * Hh + 2], dc[2], fmaf(w0[2 * Hh + 2], dc[1], w0[Hh + 2] * dc[0]));
    v.w = fmaf(w0[3 * Hh + 3], dc[2], fmaf(w0[2 * Hh + 3], dc[1], w0[Hh + 3] * dc[0]));
    v.x = hv.x > 0.f ? v.x : 0.f; v.y = hv.y > 0.f ? v.y : 0.f; v.z = hv.z > 0.f ? v.z : 0.f; v.w = hv.w > 0.f ? v.w : 0.f;
    u.x = w0[0] * da; u.y = w0[1] * da; u.z = w0[2] * da; u.w = w0[3] * da;
    // (act != 0: the gradient buffers are 16-bit too, stored as the next GEMM's operand -- scaled by gscale)
    act_st4(d_hc, o * Hh + h, v, act, gscale);
    if (d_h4) act_st4(d_h4, o * Hh + h, u, act, gscale);
  }
}

// Head weight gradients: d wa[f] = sum_i dhead[i][0] h4[i][f], d Woc[x][f] = sum_i dhead[i][1+x] hc[i][f], biases =
// column sums of dhead.  A 4 x H output: a GEMM tile would be 94 % padding, so this is a plain reduction over the
// samples (reads h4 and hc once, coalesced along f); one atomic per block and entry into the pre-zeroed gradient.
__global__ __launch_bounds__(256) void head_wgrad_kernel(int Hh, long n, const float* dhead, const float* h4, const float* hc,
                                                         float* grads, long p_stride, int off_wa, int off_ba, int off_woc,
                                                         int off_boc, float* partA, float* partW, float* rsA, float* rsW,
                                                         int act) {
  __shared__ float red[4][256];
  const long z = blockIdx.y;
  const int rows = 256 / Hh > 0 ? 256 / Hh : 1;            // sample rows per pass (H <= 256)
  const int f = threadIdx.x % Hh, row = threadIdx.x / Hh;
  const bool live = row < rows;
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  float a = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  if (live)
    for (long i = i0 + row; i < i1; i += rows) {
      const long o = z * n + i;
      const float4 d = *reinterpret_cast<const float4*>(dhead + o * 4);
      const float hv = act_ld(h4, o * Hh + f, act), cv = act_ld(hc, o * Hh + f, act);
      a = fmaf(d.x, hv, a); c0 = fmaf(d.y, cv, c0); c1 = fmaf(d.z, cv, c1); c2 = fmaf(d.w, cv, c2);
    }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = c0; red[2][threadIdx.x] = c1; red[3][threadIdx.x] = c2;
  __syncthreads();
  float* G = grads + z * p_stride;
  const long zb = z * gridDim.x + blockIdx.x;       // partial slot (reduce_parts_kernel layout: [z][slice][M][N])
  if (row == 0) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rows; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += red[q][r * Hh + f];
    if (partA) {
      partA[zb * Hh + f] = s[0];
#pragma unroll
      for (int x = 0; x < 3; ++x) partW[(zb * 3 + x) * Hh + f] = s[1 + x];
    } else {
      atomicAdd(G + off_wa + f, s[0]);
#pragma unroll
      for (int x = 0; x < 3; ++x) atomicAdd(G + off_woc + x * Hh + f, s[1 + x]);
    }
  }
  // biases: lanes 0..3 of the last wave sum their dhead column over the block's samples
  if (threadIdx.x >= 192) {
    const int lane = threadIdx.x - 192, q = lane & 3;
    float b = 0.f;
    for (long i = i0 + (lane >> 2); i < i1; i += 16) b += dhead[(z * n + i) * 4 + q];
    b += __shfl_xor(b, 4); b += __shfl_xor(b, 8); b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    if (lane < 4) {
      if (partA) { if (q == 0) rsA[zb] = b; else rsW[zb * 3 + q - 1] = b; }
      else atomicAdd(G + (q == 0 ? off_ba : off_boc + q - 1), b);
    }
  }
}

__global__ void relu_mask_kernel(long n, float* d, const float* act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = act[i] > 0.f ? d[i] : 0.f;
}

// PE backward: d B[j][x] = sum_n t[n][x] sum_f d_emb[n][3 + 21 f + j] cos(arg) pi 2^f     (embedding.py:48-52)
// lane = (sample slot 0..2, direction j): 63 lanes per wave, the six d_emb reads of a lane group are contiguous
// over j; three accumulators per thread, LDS reduction per block, one atomic per block and entry.
__global__ __launch_bounds__(256) void pe_bwd_kernel(long n, const float* params, long p_stride, int off_B,
                                                     const float* scale, const float* pts, const float* d_emb,
                                                     float* dB /* [K][63], pre-zeroed */,
                                                     float* part /* NULL or [K][gridDim.x][63] block partials */) {
  __shared__ float red[4][63][3];
  const long z = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % OBJ_NDIR, s3 = lane / OBJ_NDIR;      // lane 63: s3 == 3 -> idle
  const float* B = params + z * p_stride + off_B;
  const float b0 = B[3 * j], b1 = B[3 * j + 1], b2 = B[3 * j + 2];
  const float sc = scale[z];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const long stride = (long)gridDim.x * 12;
  for (long i = ((long)blockIdx.x * 4 + wv) * 3 + s3; i < n && s3 < 3; i += stride) {
    const float* p = pts + (z * n + i) * 3;
    const float t0 = p[0] / sc, t1 = p[1] / sc, t2 = p[2] / sc;
    const float* de = d_emb + (z * n + i) * OBJ_EMB + 3 + j;
    const float pj = fmaf(t2, b2, fmaf(t1, b1, t0 * b0));
    float dp = 0.f;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const float sf = (float)(1 << f);
      float sv, cv;
      sincos_acc((pj * sf) * OBJ_PI_F, sv, cv);
      dp += de[f * OBJ_NDIR] * ((cv * OBJ_PI_F) * sf);
    }
    a0 = fmaf(dp, t0, a0);
    a1 = fmaf(dp, t1, a1);
    a2 = fmaf(dp, t2, a2);
  }
  if (lane < 63) { red[wv][lane][0] = a0; red[wv][lane][1] = a1; red[wv][lane][2] = a2; }
  __syncthreads();
  if (threadIdx.x < 63) {
    const int jj = threadIdx.x / 3, x = threadIdx.x % 3;
    float v = 0.f;
    for (int ww = 0; ww < 4; ++ww)
      for (int ss = 0; ss < 3; ++ss) v += red[ww][ss * OBJ_NDIR + jj][x];
    if (part) part[(z * gridDim.x + blockIdx.x) * 63 + threadIdx.x] = v;
    else atomicAdd(&dB[z * 63 + threadIdx.x], v);
  }
}

__global__ void copy_cols_kernel(long rows, int cols, const float* src, long lds_, float* dst, long ldd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) dst[(i / cols) * ldd + (i % cols)] = src[(i / cols) * lds_ + (i % cols)];
}


// ------------------------------------------------------------------------------------------------
// Feature-distillation branch with the 512-d head hoisted past the compositing, any hidden width Hh
// (the hidden-32 fused kernel has its own copies in objnerf_train.hip).
// beta[r] = b_of . g[r], |g[r]| into rayin[r][Hh], [Hh + 1]  (u = W_of^T g is a GEMM): 16 lanes per ray
__global__ __launch_bounds__(256) void featg_rowstats_kernel(const float* params, long p_stride, int off_b, int C, int R,
                                                             int rin_ld, const float* gt_feat, float* rayin) {
  const int k = blockIdx.y;
  const int l16 = threadIdx.x & 15;
  const long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const float* B = params + (long)k * p_stride + off_b;
  float bs = 0.f, gs = 0.f;
  if (r < R) {
    const float* gp = gt_feat + ((long)k * R + r) * C;
    for (int cc = l16; cc < C; cc += 16) {
      const float gv = gp[cc];
      bs = fmaf(B[cc], gv, bs);
      gs = fmaf(gv, gv, gs);
    }
  }
  bs = dpp_rowsum16(bs);
  gs = dpp_rowsum16(gs);
  if (r < R && l16 == 0) {
    float* o = rayin + ((long)k * R + r) * rin_ld;
    o[rin_ld - 2] = bs;
    o[rin_ld - 1] = sqrtf(gs);
  }
}
// wb = W_of^T b_of, bb = b_of . b_of appended to the Gram matrix: gram[k][Hh * Hh + h], [Hh * Hh + Hh]
__global__ __launch_bounds__(64) void featg_wb_kernel(const float* params, long p_stride, int off_w, int off_b, int C,
                                                      int Hh, float* gram, long gstride) {
  const int k = blockIdx.y, h = blockIdx.x;            // one wave per entry h (h == Hh: bb)
  const float* W = params + (long)k * p_stride + off_w;
  const float* B = params + (long)k * p_stride + off_b;
  float acc = 0.f;
  for (int cc = threadIdx.x; cc < C; cc += 64) acc = fmaf(h < Hh ? W[(long)cc * Hh + h] : B[cc], B[cc], acc);
  acc = wave_sum64(acc);
  if (threadIdx.x == 0) gram[(long)k * gstride + (long)Hh * Hh + h] = acc;
}
// X1 = [a fh | a O], X2 = [c fh | c O] from rayfeat rows (fh[Hh], O, a, c)
__global__ void featg_scale_kernel(long n, int Hh, const float* rayfeat, float* X1, float* X2) {
  const int XC = Hh + 1;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * XC) return;
  const long r = i / XC;
  const int j = (int)(i - r * XC);
  const float* rf = rayfeat + r * (Hh + 3);
  const float v = rf[j];
  X1[i] = rf[Hh + 1] * v;
  X2[i] = rf[Hh + 2] * v;
}
// d W_of[c][h] = T[c][h] + sum_j W_of[c][j] M[j][h] + b_of[c] M[Hh][h];  d b_of[c] = T[c][Hh] + W_of[c] . M[Hh][:] + b_of[c] M[Hh][Hh]
__global__ __launch_bounds__(256) void featg_finish_kernel(const float* params, long p_stride, int off_w, int off_b, int C,
                                                           int Hh, const float* Tm, const float* mom, float* grads) {
  const int XC = Hh + 1;
  const int k = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)C * XC) return;
  const int cc = (int)(i / XC), hh = (int)(i - (long)cc * XC);
  const float* W = params + (long)k * p_stride + off_w + (long)cc * Hh;
  const float bc = params[(long)k * p_stride + off_b + cc];
  const float* M = mom + (long)k * XC * XC;
  float v = Tm[(long)k * C * XC + i];
  // four partial sums: the loads of four steps are in flight together (Hh is a multiple of 32)
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (hh < Hh) {
    for (int j = 0; j < Hh; j += 4) {
      v0 = fmaf(W[j], M[(long)j * XC + hh], v0);
      v1 = fmaf(W[j + 1], M[(long)(j + 1) * XC + hh], v1);
      v2 = fmaf(W[j + 2], M[(long)(j + 2) * XC + hh], v2);
      v3 = fmaf(W[j + 3], M[(long)(j + 3) * XC + hh], v3);
    }
    v += (v0 + v1) + (v2 + v3);
    grads[(long)k * p_stride + off_w + (long)cc * Hh + hh] = fmaf(bc, M[(long)Hh * XC + hh], v);
  } else {
    for (int j = 0; j < Hh; j += 4) {
      v0 = fmaf(W[j], M[(long)Hh * XC + j], v0);
      v1 = fmaf(W[j + 1], M[(long)Hh * XC + j + 1], v1);
      v2 = fmaf(W[j + 2], M[(long)Hh * XC + j + 2], v2);
      v3 = fmaf(W[j + 3], M[(long)Hh * XC + j + 3], v3);
    }
    v += (v0 + v1) + (v2 + v3);
    grads[(long)k * p_stride + off_b + cc] = fmaf(bc, M[(long)Hh * XC + Hh], v);
  }
}

// FeatPrepTask (512 threads per workgroup): featg_wb_kernel's, featg_rowstats_kernel's arithmetic and the [W_of | b_of] copy.
// b = (object, pass-local index)
__device__ void feat_prep_task(const FeatPrepTask& t, int b) {
  const int per = t.nb_wb + t.nb_rs + t.nb_snap;
  const int k = b / per;
  b -= k * per;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* W = t.params + (long)k * t.ps + t.off_w;
  const float* B = t.params + (long)k * t.ps + t.off_b;
  if (b < t.nb_wb) {                                       // wb[h] = W_of[:, h] . b_of (h < Hh), bb = b_of . b_of: a wave per entry
    const int h = 8 * b + wv;
    if (h <= t.Hh) {
      float acc = 0.f;
      for (int cc = lane; cc < t.C; cc += 64) acc = fmaf(h < t.Hh ? W[(long)cc * t.Hh + h] : B[cc], B[cc], acc);
      acc = wave_sum64(acc);
      if (lane == 0) t.gram[(long)k * t.gstride + (long)t.Hh * t.Hh + h] = acc;
    }
    return;
  }
  b -= t.nb_wb;
  if (b < t.nb_rs) {                                       // beta, |g| of 32 rays: 16 lanes per ray
    const int l16 = tid & 15;
    const long r = (long)b * 32 + (tid >> 4);
    float bs = 0.f, gs = 0.f;
    if (r < t.R) {
      const float* gp = t.gt_feat + ((long)k * t.R + r) * t.C;
      for (int cc = l16; cc < t.C; cc += 16) {
        const float gv = gp[cc];
        bs = fmaf(B[cc], gv, bs);
        gs = fmaf(gv, gv, gs);
      }
    }
    bs = dpp_rowsum16(bs);
    gs = dpp_rowsum16(gs);
    if (r < t.R && l16 == 0) {
      float* o = t.rayin + ((long)k * t.R + r) * t.rin_ld;
      o[t.rin_ld - 2] = bs;
      o[t.rin_ld - 1] = sqrtf(gs);
    }
    return;
  }
  b -= t.nb_rs;
  if (t.snap) {                                            // [W_of (C x Hh) | b_of (C)] as they are BEFORE the step
    float* o = t.snap + (long)k * ((long)t.C * (t.Hh + 1));
    const long tot = (long)t.C * (t.Hh + 1);
    for (long i = (long)b * 2048 + tid; i < tot && i < (long)(b + 1) * 2048; i += 512)
      o[i] = i < (long)t.C * t.Hh ? W[i] : B[i - (long)t.C * t.Hh];
  }
}
// featg_finish_kernel + the head's AdamW in one launch (the one-launch background iteration with an optimiser attached):
// the entry's gradient is formed from the COPY of [W_of | b_of] (FeatPrepTask; a gradient needs a whole row of W_of that
// other threads of this launch are stepping) and stepped at once -- adamw_dyn_kernel's arithmetic for the feature group
// [lo2, hi2), whose counter the reduction launch has already advanced in the other bank.
struct FeatFinishOpt {
  float* params; float* m; float* v; const int* flags; const int* steps; int bank;
  double lr, b1, b2, wd; float eps;
};
__global__ __launch_bounds__(256) void featg_finish_adamw_kernel(const float* snap, long p_stride, int off_w, int off_b, int C,
                                                                 int Hh, const float* Tm, const float* mom, float* grads,
                                                                 const FeatFinishOpt o) {
  const int XC = Hh + 1;
  const int k = blockIdx.y;
  __shared__ float s_step, s_bc2;
  __shared__ int s_act;
  if (threadIdx.x == 0) {
    const bool f0 = o.flags[0] != 0;
    s_act = !f0;                                           // (group 2 of adamw_dyn_kernel: skipped when no ray has label 1)
    const double st = (double)(o.steps[3 * o.bank + 2] + 1);
    s_step = (float)(o.lr / (1.0 - pow(o.b1, st)));
    s_bc2 = (float)sqrt(1.0 - pow(o.b2, st));
  }
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)C * XC) return;
  const int cc = (int)(i / XC), hh = (int)(i - (long)cc * XC);
  const float* hs = snap + (long)k * ((long)C * XC);
  const float* W = hs + (long)cc * Hh;
  const float bc = hs[(long)C * Hh + cc];
  const float* M = mom + (long)k * XC * XC;
  float v = Tm[(long)k * C * XC + i];
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, gr;
  long idx;
  if (hh < Hh) {
    for (int j = 0; j < Hh; j += 4) {
      v0 = fmaf(W[j], M[(long)j * XC + hh], v0);
      v1 = fmaf(W[j + 1], M[(long)(j + 1) * XC + hh], v1);
      v2 = fmaf(W[j + 2], M[(long)(j + 2) * XC + hh], v2);
      v3 = fmaf(W[j + 3], M[(long)(j + 3) * XC + hh], v3);
    }
    v += (v0 + v1) + (v2 + v3);
    gr = fmaf(bc, M[(long)Hh * XC + hh], v);
    idx = (long)k * p_stride + off_w + (long)cc * Hh + hh;
  } else {
    for (int j = 0; j < Hh; j += 4) {
      v0 = fmaf(W[j], M[(long)Hh * XC + j], v0);
      v1 = fmaf(W[j + 1], M[(long)Hh * XC + j + 1], v1);
      v2 = fmaf(W[j + 2], M[(long)Hh * XC + j + 2], v2);
      v3 = fmaf(W[j + 3], M[(long)Hh * XC + j + 3], v3);
    }
    v += (v0 + v1) + (v2 + v3);
    gr = fmaf(bc, M[(long)Hh * XC + Hh], v);
    idx = (long)k * p_stride + off_b + cc;
  }
  grads[idx] = gr;
  if (s_act) {
    const float decay = (float)(1.0 - o.lr * o.wd), w1 = (float)(1.0 - o.b1), w2 = (float)(1.0 - o.b2), beta2 = (float)o.b2;
    float p = o.params[idx] * decay;
    const float mo = o.m[idx];
    const float mn = mo + w1 * (gr - mo);
    const float vn = o.v[idx] * beta2 + (w2 * gr) * gr;
    const float denom = sqrtf(vn) / s_bc2 + o.eps;
    p = p + (-s_step) * (mn / denom);
    o.params[idx] = p;
    o.m[idx] = mn;
    o.v[idx] = vn;
  }
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Offs {
  int64_t o[OBJNERF_N_TENSORS + 1];
  int64_t ps;
};

// workspace carve (floats)
struct WS {
  float *emb, *h1, *h2, *h3, *h4, *hc, *hf, *alpha, *color, *d_alpha, *d_color, *dhead;
  float *d_hf, *rayin, *gram, *rayfeat, *X1, *X2, *Tm, *mom;      // feature branch (hoisted head)
  float *dA, *dB_, *dC, *dD, *dE, *d_emb, *dBpe, *pts;
  float* parts; size_t parts_floats;      // split-K partial slabs of the step's weight-gradient GEMMs (wgrad(E, ))
  float* loss_part;                       // [K R][4] block partials of the loss terms
  float* packb;                           // [K][90112] 16-bit: packed B image of the AFULL layer GEMM in flight
  size_t act_floats;                      // room of each of h1 .. hc (floats)
  int* counts;
  size_t bytes;
};

// the 16-bit modes keep h1 .. hc in the operand type when the resident-panel GEMMs serve the shape (hidden 256, >= 4096
// samples per object): those five buffers then take half the room
static bool acts16_shape(int H, long n) { return OBJ_ACT16 && OBJ_G16_AFULL && H == 256 && n >= 4096; }
static WS carve(char* base, int H, int C, long n, long R, int K, bool feat, bool half_acts = false) {
  WS w;
  char* p = base;
  auto take = [&](size_t floats) { float* r = (float*)p; p += al(floats * 4); return r; };
  w.emb = take((size_t)K * n * OBJ_EMB);
  const size_t act = half_acts ? (size_t)K * n * H / 2 : (size_t)K * n * H;
  w.h1 = take(act); w.h2 = take(act); w.h3 = take(act); w.h4 = take(act); w.hc = take(act);
  w.act_floats = act;
  w.hf = feat ? take((size_t)K * n * H) : nullptr;
  w.d_hf = feat ? take((size_t)K * n * H) : nullptr;
  w.rayin = feat ? take((size_t)K * R * (H + 2)) : nullptr;
  w.gram = feat ? take((size_t)K * ((size_t)H * H + H + 1)) : nullptr;
  w.rayfeat = feat ? take((size_t)K * R * (H + 3)) : nullptr;
  w.X1 = feat ? take((size_t)K * R * (H + 1)) : nullptr;
  w.X2 = feat ? take((size_t)K * R * (H + 1)) : nullptr;
  w.Tm = feat ? take((size_t)K * C * (H + 1) + (size_t)K * (H + 1) * (H + 1)) : nullptr;   // Tm | mom, zeroed together
  w.mom = feat ? w.Tm + (size_t)K * C * (H + 1) : nullptr;
  w.alpha = take((size_t)K * n); w.color = take((size_t)K * n * 3);
  w.d_alpha = take((size_t)K * n); w.d_color = take((size_t)K * n * 3);
  w.dhead = take((size_t)K * n * 4);
  w.dA = take(act); w.dB_ = take(act); w.dC = take(act); w.dD = take(act); w.dE = take(act);
  w.d_emb = take((size_t)K * n * OBJ_EMB);
  w.dBpe = take((size_t)K * 64);
  w.pts = take((size_t)K * n * 3);          // sample positions of the origins / directions form of the batch
  {
    // every weight-gradient GEMM of the step with its own slice count (wgrad(E, )): (M, N, samples, has bias)
    const size_t XC = (size_t)H + 1, Hs = (size_t)H;
    size_t tot = 0;
    auto add = [&](int M, int N, long ns, bool bias) {
      tot += (size_t)K * wgrad_slices(K, M, N, ns) * ((size_t)M * N + (bias ? M : 0)) + 128;
    };
    add(H, H, n, true); add(H, OBJ_E2, n, false);          // colour layer
    add(H, H, n, true);                                    // mid2
    add(H, H, n, true); add(H, OBJ_E1, n, false);          // cat layer
    add(H, H, n, true);                                    // mid1
    add(H, OBJ_E1, n, true);                               // in layer
    add(1, H, n, true); add(3, H, n, true);                // heads (H > 256 only)
    if (feat) {
      add(H, H, n, true); add(H, OBJ_E2, n, false);        // feature layer
      add(C, (int)XC, R, false); add((int)XC, (int)XC, R, false);   // 512-d head moments
    }
    // (+ the block partials of the head weight gradients and of d B: at most 2048 + 1024 blocks of 4 H + 4 / 63 floats)
    w.parts_floats = tot + (size_t)(2048 + K) * (4 * Hs + 68) + (size_t)K * 1024 * 64;
    w.parts = take(w.parts_floats);
    w.loss_part = take((size_t)K * R * 4);
    w.packb = H > 128 ? take((size_t)K * 45056) : nullptr;
  }
  w.counts = (int*)take((size_t)2 * K + 2);
  w.bytes = (size_t)(p - base);
  return w;
}

size_t train_workspace_bytes(const objnerf_net* net, int K, int R, int S, int feat, int sixteen) {
  WS w = carve(nullptr, net->hidden, net->feat_dim, (long)R * S, (long)R, K, feat != 0,
               sixteen && acts16_shape(net->hidden, (long)R * S));
  size_t need = w.bytes + 256;
  if (net->hidden == 256 && (S == 32 || S == 64 || S == 128)) {      // the fused hidden-256 path (16-bit modes)
    const size_t n256 = obj256::workspace_bytes(K, R, S, feat, net->feat_dim);
    if (n256 > need) need = n256;
  }
  return need;
}

}  // namespace objgen

// objnerf_context: the helper streams + events of the layer-wise path, owned by the CALLER (objnerf_context_create /
// _destroy are the only entries of the library that create anything).  Without one every launch stays on `stream`.
struct objnerf_context {
  static constexpr int NEV = 16, NS = 3;
  hipStream_t s = nullptr;             // = all[0]
  hipStream_t all[NS];                 // small batches: the independent weight-gradient GEMMs spread over three streams
  hipEvent_t ev[NEV];
  hipEvent_t done, done_all[NS];
  bool own = false;                    // false: single-stream stand-in (all[] = the caller's stream, no events)
};

extern "C" int objnerf_context_create(objnerf_context** out) {
  if (!out) return OBJNERF_EINVAL;
  objnerf_context* c = new objnerf_context();
  bool ok = true;
  for (int i = 0; i < objnerf_context::NS; ++i) {   // non-blocking: never synchronises with the null stream
    ok &= hipStreamCreateWithFlags(&c->all[i], hipStreamNonBlocking) == hipSuccess;
    ok &= hipEventCreateWithFlags(&c->done_all[i], hipEventDisableTiming) == hipSuccess;
  }
  for (int i = 0; i < objnerf_context::NEV; ++i) ok &= hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
  c->s = c->all[0];
  c->own = true;
  if (!ok) { delete c; return OBJNERF_ELAUNCH; }
  *out = c;
  return OBJNERF_OK;
}

extern "C" int objnerf_context_destroy(objnerf_context* c) {
  if (!c) return OBJNERF_OK;
  if (c->own) {
    for (int i = 0; i < objnerf_context::NS; ++i) { (void)hipStreamDestroy(c->all[i]); (void)hipEventDestroy(c->done_all[i]); }
    for (int i = 0; i < objnerf_context::NEV; ++i) (void)hipEventDestroy(c->ev[i]);
    (void)hipEventDestroy(c->done);
  }
  delete c;
  return OBJNERF_OK;
}

namespace objgen {
namespace {
typedef objnerf_context Side;
}  // namespace

// test hook (objnerf_train_args.relu_masks): one thread per (object, sample, byte of 8 features)
__global__ void relu_mask_kernel(long nb, int H, const float* act /* [K n][H] */, uint8_t* masks /* [K n][6][H/8] */,
                                 int layer, int act_mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  const int hb = H / 8;
  const long s = i / hb;
  const int b = (int)(i - s * hb);
  unsigned m = 0;
  for (int j = 0; j < 8; ++j) m |= (act_ld(act, s * H + 8 * b + j, act_mode) > 0.0f ? 1u : 0u) << j;
  masks[(s * 6 + layer) * hb + b] = (uint8_t)m;
}

// pts = (origin + dir * z) - centre, two roundings as vmap.py:548-551 (the fused kernels form it in registers; the
// layer-wise path reads the points three times and keeps them in its workspace)
__global__ void form_points_kernel(long total, int S, const float* origins, const float* dirs, const float* z, float centre,
                                   float* pts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long ray = i / S;
  const float zz = z[i];
#pragma unroll
  for (int x = 0; x < 3; ++x) pts[i * 3 + x] = (origins[ray * 3 + x] + dirs[ray * 3 + x] * zz) - centre;
}

// forward declaration: the one-launch iteration (after the workspace carve below)
static int train_step_small(const objnerf_net* net, const objnerf_train_args* a, hipStream_t st, GemmEnv& E, const WS& w,
                            const int64_t* off, bool bf, int* done);

int train_step(const objnerf_net* net, const objnerf_train_args* a_in, void* stream, int* done) {
  objnerf_train_args a_local = *a_in;
  const objnerf_train_args* a = &a_local;
  GemmEnv E;                                   // this call's GEMM environment (nothing outlives the call)
  E.operands = (a->mode & OBJNERF_TRAIN_FP16) ? 2 : (a->mode & OBJNERF_TRAIN_BF16) ? 1 : 0;
  const int H = net->hidden, C = net->feat_dim, K = a->K;
  if (H % 32 != 0 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  const bool feat = a->gt_feat != nullptr;
  const long n = (long)a->R * a->S;
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  const long ps = a->p_stride;
  hipStream_t st = (hipStream_t)stream;
  const bool half_acts = (a->mode & (OBJNERF_TRAIN_FP16 | OBJNERF_TRAIN_BF16)) != 0 && acts16_shape(H, n);
  WS w = carve((char*)a->workspace, H, C, n, (long)a->R, K, feat, half_acts);
  if (a->workspace_bytes < w.bytes) return OBJNERF_EINVAL;
  E.parts = w.parts; E.parts_cap = w.parts_floats; E.parts_off = 0;
  E.packb = w.packb; E.packb_entries = w.packb ? K : 0;
  // ---- small batches of the hidden-128 network without the feature loss: ONE forward + loss + backward launch, the
  // grouped weight gradients, one reduction launch (objnerf_small_body.h)
  {
    const int S = a->S;
    const int rpw = S > 0 && S <= 64 ? sf_rays_per_wg(S, feat) : 0;
    const long nwg = rpw ? ((long)a->R + rpw - 1) / rpw : 0;
#ifndef OBJ_NO_SMALL_FUSED
    // (the per-workgroup partials are parked in workspace regions of n = R S rows per object: the d B partials, 63 floats
    // per workgroup, in dhead (4 n floats) and the head partials, 4 H + 4 per workgroup, in d_emb (129 n) -- tiny batches
    // such as R <= 3 with S = 4 do not fit and take the general path)
    if (H == FS_H && S >= 4 && S <= 64 && (long)K * nwg <= 1536 && !half_acts && K <= 65535 &&
        nwg * 63 <= n * 4 && nwg * (4L * H + 4) <= n * 129 &&
        !(a->mode & OBJNERF_TRAIN_LAYERWISE && K > 8) && (!feat || (C % 4 == 0 && a->R >= 1)))
      return train_step_small(net, a, st, E, w, off, E.operands == 1, done);
#endif
  }
  if (a->mode & OBJNERF_TRAIN_SELF_COUNTS) {
    const int rc0 = objnerf_label_counts(K, a->R, a->labels, const_cast<int32_t*>(a->counts), const_cast<int32_t*>(a->flags),
                                         stream);
    if (rc0) return rc0;
  }
  RedGroup step_red;                 // small-batch path: every ordered reduction of the step in ONE launch, after the join
  step_red.count = 0;
  if (!a->pts) {
    const long total = (long)K * n;
    hipLaunchKernelGGL(form_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, a->S, a->origins,
                       a->dirs, a->z, a->obj_center, w.pts);
    a_local.pts = w.pts;
  }
  const float* P = a->params;
  float* G = a->grads;
  const int E1 = OBJ_E1, E2 = OBJ_E2, EM = OBJ_EMB;
  const long nH = n * H;

  // helper stream: work that only READS what the main chain produced (or, for the feature preparation below, only
  // parameters and inputs) runs beside it; fork = the side stream waits for everything enqueued on `st` so far
  Side single;                                   // no context: one stream, program order replaces every event
  for (int i = 0; i < Side::NS; ++i) single.all[i] = st;
  single.s = st;
  Side& sd = a->context ? *(Side*)a->context : single;
  const bool multi = sd.own;
  int fk = 0;
  const int n_side = 1;
  auto fork = [&]() {
    if (!multi) return;
    (void)hipEventRecord(sd.ev[fk], st);
    for (int i = 0; i < n_side; ++i) (void)hipStreamWaitEvent(sd.all[i], sd.ev[fk], 0);
    fk = (fk + 1) % Side::NEV;
  };
  hipStream_t ss = sd.s;
  // Weight / bias gradients are accumulated with split-K atomics by GEMMs on the side stream: zero them there, now,
  // off the critical path (feature-branch entries only when they receive a gradient, so "no gradient" stays
  // "untouched").  The previous step's consumer of the gradient arena (AdamW) is ordered before this fork.
  fork();
  for (int k = 0; k < K; ++k) {
    (void)hipMemsetAsync(a->grads + (long)k * ps, 0, (size_t)off[14] * 4, ss);
    if (feat) (void)hipMemsetAsync(a->grads + (long)k * ps + off[14], 0, (size_t)(off[18] - off[14]) * 4, ss);
  }
  if (feat) {
    // the 512-d head is NOT applied per sample: per object G = W_of^T W_of (+ wb, bb), per ray u = W_of^T g, beta, |g|.
    // None of it depends on the forward pass: side stream, joined before the loss.
    const long R = a->R;
    const long gst = (long)H * H + H + 1;
    gemm(E, ss, K, H, H, C, P + off[16], 1, H, ps, P + off[16], H, 1, ps, w.gram, H, 1, gst);
    hipLaunchKernelGGL(featg_wb_kernel, dim3(H + 1, K), dim3(64), 0, ss, P, ps, (int)off[16], (int)off[17], C, H, w.gram, gst);
    gemm(E, ss, K, (int)R, H, C, a->gt_feat, C, 1, R * C, P + off[16], H, 1, ps, w.rayin, H + 2, 1, R * (H + 2));
    hipLaunchKernelGGL(featg_rowstats_kernel, dim3((unsigned)((R + 15) / 16), K), dim3(256), 0, ss, P, ps, (int)off[17], C,
                       (int)R, H + 2, a->gt_feat, w.rayin);
    if (multi) (void)hipEventRecord(sd.done, ss);
  }

  // ---- forward
  int rc = objnerf_embed(net, K, n, P, ps, a->scale, a->pts, w.emb, stream);
  if (rc) return rc;
  dim3 eg((unsigned)((n + 15) / 16), (unsigned)K);      // 16 lanes per sample row
  const size_t head_lds = (size_t)4 * H * sizeof(float);
  const int small_rt = small_batch_rt(E, H, n, K);
  // the small-batch kernels come in fp32 and bf16-MFMA form (fp16 mode: fp32 there); their weight-gradient GEMMs stay
  // fp32 (one grouped launch)
  const bool small_bf = small_rt && E.operands == 1;
  if (small_rt) E.operands = 0;
  // 16-bit modes at hidden 256 (configs[4]): h1 .. hc live in the operand type -- written by the forward GEMMs'
  // epilogues, read as panels / masks / weight-gradient operands and by the head kernels (act_ld).  Their buffers keep
  // the fp32 spacing in the workspace; gemm(E, ) recognises them by address.
  const int act16 = half_acts ? E.operands : 0;
  if (act16 && !panel_ok(E, (int)n, H, H + E1, 1, 1, false, K)) return OBJNERF_EINVAL;      // (cannot happen: same conditions)
  if (act16) {
    E.act16_lo = (const char*)w.h1; E.act16_hi = (const char*)(w.hc + w.act_floats);
    E.grad16_lo = (const char*)w.dA; E.grad16_hi = (const char*)(w.dE + w.act_floats);
  }
  if (small_rt) {
    FwdSmall f;
    f.n = n; f.feat = feat ? 1 : 0; f.params = P; f.ps = ps; f.emb = w.emb;
    f.h1 = w.h1; f.h2 = w.h2; f.h3 = w.h3; f.h4 = w.h4; f.hc = w.hc; f.hf = feat ? w.hf : nullptr;
    f.alpha = w.alpha; f.color = w.color;
    f.o_in_w = (int)off[0]; f.o_in_b = (int)off[1]; f.o_m1_w = (int)off[2]; f.o_m1_b = (int)off[3];
    f.o_cat_w = (int)off[4]; f.o_cat_b = (int)off[5]; f.o_m2_w = (int)off[6]; f.o_m2_b = (int)off[7];
    f.o_a_w = (int)off[8]; f.o_a_b = (int)off[9]; f.o_cl_w = (int)off[10]; f.o_cl_b = (int)off[11];
    f.o_oc_w = (int)off[12]; f.o_oc_b = (int)off[13]; f.o_fl_w = (int)off[14]; f.o_fl_b = (int)off[15];
    switch (small_rt) {
      case 1: launch_fwd_small<1>(st, f, K, small_bf); break;
      case 2: launch_fwd_small<2>(st, f, K, small_bf); break;
      case 3: launch_fwd_small<3>(st, f, K, small_bf); break;
      case 4: launch_fwd_small<4>(st, f, K, small_bf); break;
      default: launch_fwd_small<5>(st, f, K, small_bf); break;
    }
  } else {
  // h1 = relu(x1 W_in^T + b)
  gemm(E, st, K, n, H, E1, w.emb, EM, 1, n * EM, P + off[0], 1, E1, ps, w.h1, H, 1, nH, false, P + off[1], ps, true);
  gemm(E, st, K, n, H, H, w.h1, H, 1, nH, P + off[2], 1, H, ps, w.h2, H, 1, nH, false, P + off[3], ps, true);
  // h3 = relu([h2 | x1] W_cat^T + b)
  // ([h2 | x1] as ONE contraction when the resident-panel kernel takes it: no round trip of the partial result)
  const bool fuse2 = panel_ok(E, (int)n, H, H + E1, 1, 1, false, K) && H == 256;
  if (fuse2) {
    E.a2 = A2Src{w.emb, EM, n * EM, H};
    gemm(E, st, K, n, H, H + E1, w.h2, H, 1, nH, P + off[4], 1, H + E1, ps, w.h3, H, 1, nH, false, P + off[5], ps, true);
  } else {
  gemm(E, st, K, n, H, H, w.h2, H, 1, nH, P + off[4], 1, H + E1, ps, w.h3, H, 1, nH);
  gemm(E, st, K, n, H, E1, w.emb, EM, 1, n * EM, P + off[4] + H, 1, H + E1, ps, w.h3, H, 1, nH, true, P + off[5], ps, true);
  }
  gemm(E, st, K, n, H, H, w.h3, H, 1, nH, P + off[6], 1, H, ps, w.h4, H, 1, nH, false, P + off[7], ps, true);
  // hc = relu([h4 | x2] W_cl^T + b)
  if (fuse2) {
    E.a2 = A2Src{w.emb + E1, EM, n * EM, H};
    gemm(E, st, K, n, H, H + E2, w.h4, H, 1, nH, P + off[10], 1, H + E2, ps, w.hc, H, 1, nH, false, P + off[11], ps, true);
  } else {
  gemm(E, st, K, n, H, H, w.h4, H, 1, nH, P + off[10], 1, H + E2, ps, w.hc, H, 1, nH);
  gemm(E, st, K, n, H, E2, w.emb + E1, EM, 1, n * EM, P + off[10] + H, 1, H + E2, ps, w.hc, H, 1, nH, true, P + off[11], ps,
       true);
  }
  hipLaunchKernelGGL(heads_fwd_kernel, eg, dim3(256), head_lds, st, H, n, w.h4, w.hc, P, ps, (int)off[8], (int)off[9],
                     (int)off[12], (int)off[13], w.alpha, w.color, act16);
  if (feat) {
    gemm(E, st, K, n, H, H, w.h4, H, 1, nH, P + off[14], 1, H + E2, ps, w.hf, H, 1, nH);
    gemm(E, st, K, n, H, E2, w.emb + E1, EM, 1, n * EM, P + off[14] + H, 1, H + E2, ps, w.hf, H, 1, nH, true, P + off[15],
         ps, true);
  }
  }
  if (a->relu_masks) {       // test hook: the ReLU branch bits of this iteration (objnerf_train_args.relu_masks)
    const float* acts[6] = {w.h1, w.h2, w.h3, w.h4, w.hc, feat ? w.hf : nullptr};
    const long nb = (long)K * n * (H / 8);
    for (int l = 0; l < 6; ++l)
      if (acts[l])
        hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, nb, H, acts[l],
                           a->relu_masks, l, l < 5 ? act16 : 0);
  }
  if (feat && multi) (void)hipStreamWaitEvent(st, sd.done, 0);      // the feature preparation (side stream) is needed from here
  // ---- loss + d(alpha, color, clip)      (loss.py:5-103)
  objnerf_loss_args la;
  la.K = K; la.R = a->R; la.S = a->S; la.C = C;
  la.color_scaling = a->color_scaling; la.opacity_scaling = a->opacity_scaling; la.feat_scaling = a->feat_scaling;
  la.reserved = 0;
  la.alpha = w.alpha; la.color = w.color; la.z = a->z; la.gt_depth = a->gt_depth; la.gt_rgb = a->gt_rgb;
  la.labels = a->labels; la.pred_feat = nullptr; la.gt_feat = a->gt_feat; la.flags_in = a->flags;
  la.counts_in = a->counts;
  la.loss_terms = a->loss_terms; la.total = nullptr; la.d_alpha = w.d_alpha; la.d_color = w.d_color;
  la.d_pred_feat = nullptr; la.counts = w.counts; la.status = a->status;
  objmisc::LossHoisted hz;
  hz.Hh = H; hz.hf = w.hf; hz.rayin = w.rayin; hz.gram = w.gram; hz.d_hf = w.d_hf; hz.rayfeat = w.rayfeat;
  rc = objmisc::step_batch_loss_impl(&la, feat ? &hz : nullptr, stream, w.loss_part);
  if (rc) return rc;
  // ---- backward
  // Weight-gradient GEMMs only READ the d-output / activation buffers and write the gradient arena, so they run on a
  // side stream beside the dgrad chain (each of these GEMMs alone leaves most of the chip idle).  Every d_h has its own
  // buffer: nothing a side-stream GEMM reads is overwritten before the join at the end.
  // fp16 operands: every backward GEMM has the back-propagated gradient as its A operand.  Those values are of order
  // (loss scale) / R per ray times a per-sample weight -- mostly below fp16's smallest normal number (6.1e-5) -- so A is
  // scaled by ~8 R (a power of two: exact) before rounding and the product scaled back (Gemm::a_scale).
  const float grad_scale = exp2f(floorf(log2f((float)a->R)) + 3.0f);
  E.a_scale = grad_scale;
  int rr = 0;
  auto side = [&]() -> hipStream_t {        // stream of the next independent weight-gradient GEMM
    hipStream_t r = sd.all[rr];
    rr = (rr + 1) % n_side;
    return r;
  };
  float* d_hc = w.dA;      // [n][H]
  float* d_h4 = w.dB_;
  // (layer-wise chain without the feature branch: the colour layer's input-gradient GEMM adds the alpha head's rank-1
  // term wa x dhead[:, 0] in its epilogue, so d_h4 is written once instead of written, read and written)
  const bool fold_a = !feat && !small_rt;
  hipLaunchKernelGGL(heads_bwd_kernel, eg, dim3(256), head_lds, st, H, n, w.hc, w.color, w.d_alpha, w.d_color, P, ps,
                     (int)off[8], (int)off[12], w.dhead, d_hc, fold_a ? nullptr : d_h4, act16,
                     act16 == 2 ? grad_scale : 1.0f);
  // head weight grads: d wa = dhead[:,0]^T h4, d Woc = dhead[:,1:4]^T hc; biases = column sums of dhead
  // (every bias gradient rides on its layer's weight-gradient GEMM: row sums of the d-output operand tile)
  fork();
  if (H > 256) {
    wgrad(E, ss, K, 1, H, n, w.dhead, 1, 4, n * 4, w.h4, H, 1, nH, G + off[8], H, ps, G + off[9]);
    wgrad(E, ss, K, 3, H, n, w.dhead + 1, 1, 4, n * 4, w.hc, H, 1, nH, G + off[12], H, ps, G + off[13]);
  } else {
    // 64 samples per block (32 iterations of a 2-row pass at H = 128): the loop is a chain of load latencies, so the
    // reduction wants many short blocks (512 samples per block: 183 us for the background batch; now ~25 us)
    int hb = (int)((n + 63) / 64);
    const int cap = (2048 + K - 1) / K;
    if (hb > cap) hb = cap;
    // block partials + an ordered reduction (reduce_parts_kernel: "GEMMs" of M = 1 / 3 rows with hb slices)
    float* pA = E.parts_alloc((size_t)K * hb * H), *pW = E.parts_alloc((size_t)K * hb * 3 * H);
    float* rA = E.parts_alloc((size_t)K * hb), *rW = E.parts_alloc((size_t)K * hb * 3);
    if (!(pA && pW && rA && rW)) pA = pW = rA = rW = nullptr;        // no scratch: float atomics
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)hb, (unsigned)K), dim3(256), 0, ss, H, n, w.dhead, w.h4, w.hc, G, ps,
                       (int)off[8], (int)off[9], (int)off[12], (int)off[13], pA, pW, rA, rW, act16);
    if (pA) {
      RedGroup rg;
      rg.count = 2; rg.beg[0] = 0;
      RedItem& a0 = rg.it[0];
      a0.part = pA; a0.rs_part = rA; a0.C = G + off[8]; a0.rowsum = G + off[9];
      a0.M = 1; a0.N = H; a0.sk = hb; a0.batch = K; a0.scm = H; a0.bsc = ps; a0.bsrs = ps;
      RedItem& a1 = rg.it[1];
      a1.part = pW; a1.rs_part = rW; a1.C = G + off[12]; a1.rowsum = G + off[13];
      a1.M = 3; a1.N = H; a1.sk = hb; a1.batch = K; a1.scm = H; a1.bsc = ps; a1.bsrs = ps;
      rg.beg[1] = red_blocks(a0); rg.beg[2] = rg.beg[1] + red_blocks(a1);
      if (small_rt) { red_append(step_red, a0); red_append(step_red, a1); }
      else hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)rg.beg[2]), dim3(256), 0, ss, rg);
    }
  }
  // d_emb needs no zero fill: the first dgrad into each column block overwrites (x2: feature layer if
  // present, else colour layer; x1: cat layer), later ones accumulate; columns 0..2 (d t) are never read.
  if (feat) {
    // d_hf (pre-activation gradient of the feature layer) came out of the loss kernel; the 512-d head's gradient is
    // d W_of = gt_feat^T [a fh] + W_of M2 + b_of m1^T, d b_of = gt_feat^T [a O] + W_of m1 + b_of s2 with the moments
    // [M2 m1; . s2] = [c fh | c O]^T [fh | O]: two split-K GEMMs over the rays + a small finish (side stream)
    float* d_hf = w.d_hf;
    const long R = a->R;
    const int XC = H + 1;
    const long nr = (long)K * R;
    fork();
    hipLaunchKernelGGL(featg_scale_kernel, dim3((unsigned)((nr * XC + 255) / 256)), dim3(256), 0, ss, nr, H, w.rayfeat, w.X1,
                       w.X2);
    (void)hipMemsetAsync(w.Tm, 0, ((size_t)K * C * XC + (size_t)K * XC * XC) * 4, ss);
    E.a_scale = 1.0f;                       // (targets and moments of the feature head: not gradients)
    wgrad(E, ss, K, C, XC, R, a->gt_feat, 1, C, R * C, w.X1, XC, 1, R * XC, w.Tm, XC, (long)C * XC);
    wgrad(E, ss, K, XC, XC, R, w.X2, 1, XC, R * XC, w.rayfeat, H + 3, 1, R * (H + 3), w.mom, XC, (long)XC * XC);
    E.a_scale = grad_scale;
    hipLaunchKernelGGL(featg_finish_kernel, dim3((unsigned)(((long)C * XC + 255) / 256), K), dim3(256), 0, ss, P, ps,
                       (int)off[16], (int)off[17], C, H, w.Tm, w.mom, G);
    // feature layer: grads + contributions to d_h4 / d_x2
    if (!small_rt) {
      wgrad(E, ss, K, H, H, n, d_hf, 1, H, nH, w.h4, H, 1, nH, G + off[14], H + E2, ps, G + off[15]);
      wgrad(E, ss, K, H, E2, n, d_hf, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[14] + H, H + E2, ps);
      gemm(E, st, K, n, H, H, d_hf, H, 1, nH, P + off[14], H + E2, 1, ps, d_h4, H, 1, nH, true);
      gemm(E, st, K, n, E2, H, d_hf, H, 1, nH, P + off[14] + H, H + E2, 1, ps, w.d_emb + E1, EM, 1, n * EM, false);
    }
  }
  float* d_h3 = w.dC;
  float* d_h2 = w.dD;
  float* d_h1 = w.dE;
  if (small_rt) {
    // small batch: the whole input-gradient chain in one launch, then the (independent) weight-gradient GEMMs
    // side by side on three streams
    BwdSmall b;
    b.n = n; b.feat = feat ? 1 : 0; b.params = P; b.ps = ps;
    b.h1 = w.h1; b.h2 = w.h2; b.h3 = w.h3; b.h4 = w.h4; b.d_hc = d_hc; b.d_hf = feat ? w.d_hf : nullptr;
    b.d_h4 = d_h4; b.d_h3 = d_h3; b.d_h2 = d_h2; b.d_h1 = d_h1; b.d_emb = w.d_emb;
    b.o_in_w = (int)off[0]; b.o_m1_w = (int)off[2]; b.o_cat_w = (int)off[4]; b.o_m2_w = (int)off[6];
    b.o_cl_w = (int)off[10]; b.o_fl_w = (int)off[14];
    switch (small_rt) {
      case 1: launch_bwd_small<1>(st, b, K, small_bf); break;
      case 2: launch_bwd_small<2>(st, b, K, small_bf); break;
      case 3: launch_bwd_small<3>(st, b, K, small_bf); break;
      case 4: launch_bwd_small<4>(st, b, K, small_bf); break;
      default: launch_bwd_small<5>(st, b, K, small_bf); break;
    }
    fork();
    GemmGroup group;
    group.count = 0;
    E.group = &group;
    E.group16 = small_bf;             // bf16 mode: the weight gradients take bf16 operands like every other GEMM of the mode
    E.red_group = &step_red;
    if (feat) {
      wgrad(E, side(), K, H, H, n, w.d_hf, 1, H, nH, w.h4, H, 1, nH, G + off[14], H + E2, ps, G + off[15]);
      wgrad(E, side(), K, H, E2, n, w.d_hf, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[14] + H, H + E2, ps);
    }
    wgrad(E, side(), K, H, H, n, d_hc, 1, H, nH, w.h4, H, 1, nH, G + off[10], H + E2, ps, G + off[11]);
    wgrad(E, side(), K, H, E2, n, d_hc, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[10] + H, H + E2, ps);
    wgrad(E, side(), K, H, H, n, d_h4, 1, H, nH, w.h3, H, 1, nH, G + off[6], H, ps, G + off[7]);
    wgrad(E, side(), K, H, H, n, d_h3, 1, H, nH, w.h2, H, 1, nH, G + off[4], H + E1, ps, G + off[5]);
    wgrad(E, side(), K, H, E1, n, d_h3, 1, H, nH, w.emb, EM, 1, n * EM, G + off[4] + H, H + E1, ps);
    wgrad(E, side(), K, H, H, n, d_h2, 1, H, nH, w.h1, H, 1, nH, G + off[2], H, ps, G + off[3]);
    wgrad(E, side(), K, H, E1, n, d_h1, 1, H, nH, w.emb, EM, 1, n * EM, G + off[0], E1, ps, G + off[1]);
    flush_group(E, ss, group);
  } else {
  // colour layer
  fork();
  wgrad(E, ss, K, H, H, n, d_hc, 1, H, nH, w.h4, H, 1, nH, G + off[10], H + E2, ps, G + off[11]);
  wgrad(E, ss, K, H, E2, n, d_hc, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[10] + H, H + E2, ps);
  if (fold_a) {
    E.biasrow = w.dhead; E.bsbr = n * 4; E.sbr = 4;
    gemm(E, st, K, n, H, H, d_hc, H, 1, nH, P + off[10], H + E2, 1, ps, d_h4, H, 1, nH, false, P + off[8], ps, false, w.h4, H, 1,
         nH);
    E.biasrow = nullptr; E.bsbr = 0; E.sbr = 1;
  } else {
    gemm(E, st, K, n, H, H, d_hc, H, 1, nH, P + off[10], H + E2, 1, ps, d_h4, H, 1, nH, true, nullptr, 0, false, w.h4, H, 1, nH);
  }
  gemm(E, st, K, n, E2, H, d_hc, H, 1, nH, P + off[10] + H, H + E2, 1, ps, w.d_emb + E1, EM, 1, n * EM, feat);
  // mid2:  d_h4 (masked above) -> grads, d_h3
  fork();
  wgrad(E, ss, K, H, H, n, d_h4, 1, H, nH, w.h3, H, 1, nH, G + off[6], H, ps, G + off[7]);
  gemm(E, st, K, n, H, H, d_h4, H, 1, nH, P + off[6], H, 1, ps, d_h3, H, 1, nH, false, nullptr, 0, false, w.h3, H, 1, nH);
  // cat layer
  fork();
  wgrad(E, ss, K, H, H, n, d_h3, 1, H, nH, w.h2, H, 1, nH, G + off[4], H + E1, ps, G + off[5]);
  wgrad(E, ss, K, H, E1, n, d_h3, 1, H, nH, w.emb, EM, 1, n * EM, G + off[4] + H, H + E1, ps);
  gemm(E, st, K, n, H, H, d_h3, H, 1, nH, P + off[4], H + E1, 1, ps, d_h2, H, 1, nH, false, nullptr, 0, false, w.h2, H, 1, nH);
  gemm(E, st, K, n, E1, H, d_h3, H, 1, nH, P + off[4] + H, H + E1, 1, ps, w.d_emb, EM, 1, n * EM, false);
  // mid1
  fork();
  wgrad(E, ss, K, H, H, n, d_h2, 1, H, nH, w.h1, H, 1, nH, G + off[2], H, ps, G + off[3]);
  gemm(E, st, K, n, H, H, d_h2, H, 1, nH, P + off[2], H, 1, ps, d_h1, H, 1, nH, false, nullptr, 0, false, w.h1, H, 1, nH);
  // in layer
  fork();
  wgrad(E, ss, K, H, E1, n, d_h1, 1, H, nH, w.emb, EM, 1, n * EM, G + off[0], E1, ps, G + off[1]);
  gemm(E, st, K, n, E1, H, d_h1, H, 1, nH, P + off[0], E1, 1, ps, w.d_emb, EM, 1, n * EM, true);
  }
  // embedding directions: block partials + ordered reduction straight into the gradient arena (float atomics into dBpe
  // and a copy when the scratch is exhausted)
  int pg = (int)((n + 47) / 48);           // 12 samples per block and pass: at least 4 passes per block
  if (pg > 1024) pg = 1024;
  if (pg < 1) pg = 1;
  float* pe_part = E.parts_alloc((size_t)K * pg * 63);
  if (!pe_part) (void)hipMemsetAsync(w.dBpe, 0, (size_t)K * 64 * 4, st);
  hipLaunchKernelGGL(pe_bwd_kernel, dim3(pg, K), dim3(256), 0, st, n, P, ps, (int)off[18], a->scale, a->pts, w.d_emb,
                     w.dBpe, pe_part);
  if (pe_part) {
    RedGroup rg;
    rg.count = 1; rg.beg[0] = 0;
    RedItem& r = rg.it[0];
    r.part = pe_part; r.rs_part = nullptr; r.C = G + off[18]; r.rowsum = nullptr;
    r.M = 1; r.N = 63; r.sk = pg; r.batch = K; r.scm = 63; r.bsc = ps; r.bsrs = 0;
    rg.beg[1] = red_blocks(r);
    if (small_rt) red_append(step_red, r);
    else hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)rg.beg[1]), dim3(256), 0, st, rg);
  } else {
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((K * 63 + 255) / 256)), dim3(256), 0, st, (long)K, 63, w.dBpe, 63L,
                       G + off[18], ps);
  }
  for (int i = 0; multi && i < n_side; ++i) {   // join: the caller's stream continues after the weight gradients
    (void)hipEventRecord(sd.done_all[i], sd.all[i]);
    (void)hipStreamWaitEvent(st, sd.done_all[i], 0);
  }
  launch_reductions(st, step_red);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  if (E.error) return OBJNERF_EINVAL;
  return OBJNERF_OK;
}

// The one-launch small-batch iteration: train_small_kernel -> the seven weight-gradient GEMMs as ONE grouped launch ->
// ONE reduction launch (weight-gradient slices, head / d B / loss partials, status, optionally AdamW).  Everything on
// the caller's stream: three dependent launches need no helper stream and no events.
static int train_step_small(const objnerf_net* net, const objnerf_train_args* a, hipStream_t st, GemmEnv& E, const WS& w,
                            const int64_t* off, bool bf, int* done) {
  const int H = net->hidden, K = a->K, S = a->S, C = net->feat_dim;
  const long n = (long)a->R * S, nH = n * H, ps = a->p_stride;
  const bool feat = a->gt_feat != nullptr;
  const int rpw = sf_rays_per_wg(S, feat);
  const int nwg = (int)(((long)a->R + rpw - 1) / rpw);
  const int E1 = OBJ_E1, E2 = OBJ_E2, EM = OBJ_EMB;
  const float* P = a->params;
  float* G = a->grads;
  const bool self_counts = (a->mode & OBJNERF_TRAIN_SELF_COUNTS) != 0;
  if (self_counts && K > 1) {        // several objects: the cross-object flags need every object's labels (one small launch)
    const int rc0 = objnerf_label_counts(K, a->R, a->labels, const_cast<int32_t*>(a->counts), const_cast<int32_t*>(a->flags),
                                         (void*)st);
    if (rc0) return rc0;
  }
  const long R = a->R;
  const int XC = H + 1;
  float* head_snap = nullptr;          // [K][C (H + 1)]: [W_of | b_of] before the step (FeatPrepTask), for the fused finish + AdamW
  if (feat) {
    // the 512-d head is NOT applied per sample: per object G = W_of^T W_of (+ wb, bb), per ray
    // u = W_of^T g, beta, |g| -- none of it depends on the forward pass; four small launches ahead of the fused one
    const long gst = (long)H * H + H + 1;
    E.operands = 0;
    // Round 6: ONE launch for the four (the two GEMMs as a group, wb / bb, the rays' beta and |g| and -- with an optimiser
    // attached -- the copy of [W_of | b_of] the head's finish reads, as extra workgroups of the same launch: FeatPrepTask)
    head_snap = a->optim ? E.parts_alloc((size_t)K * C * XC) : nullptr;
    GemmGroup g0;
    g0.count = 0;
    E.group = &g0; E.group16 = false;
    gemm(E, st, K, H, H, C, P + off[16], 1, H, ps, P + off[16], H, 1, ps, w.gram, H, 1, gst);
    gemm(E, st, K, (int)R, H, C, a->gt_feat, C, 1, R * C, P + off[16], H, 1, ps, w.rayin, H + 2, 1, R * (H + 2));
    if (g0.count == 2 && !E.error) {
      FeatPrepTask& t = g0.task;
      t.params = P; t.ps = ps; t.off_w = (int)off[16]; t.off_b = (int)off[17]; t.C = C; t.Hh = H; t.R = (int)R; t.rin_ld = H + 2;
      t.K = K; t.gt_feat = a->gt_feat; t.rayin = w.rayin; t.gram = w.gram; t.gstride = gst; t.snap = head_snap;
      t.nb_wb = (H + 1 + 7) / 8; t.nb_rs = (int)((R + 31) / 32);
      t.nb_snap = head_snap ? (int)(((long)C * XC + 2047) / 2048) : 0;
      t.blocks = K * (t.nb_wb + t.nb_rs + t.nb_snap);
      flush_group(E, st, g0);
    } else {                          // (cannot happen for these shapes; the separate launches remain the fallback)
      if (E.error) return OBJNERF_EINVAL;
      flush_group(E, st, g0);
      head_snap = nullptr;
      hipLaunchKernelGGL(featg_wb_kernel, dim3(H + 1, K), dim3(64), 0, st, P, ps, (int)off[16], (int)off[17], C, H, w.gram, gst);
      hipLaunchKernelGGL(featg_rowstats_kernel, dim3((unsigned)((R + 15) / 16), K), dim3(256), 0, st, P, ps, (int)off[17], C,
                         (int)R, H + 2, a->gt_feat, w.rayin);
    }
    E.group = nullptr;
  }
  SmallFused f;
  f.K = K; f.R = a->R; f.S = S; f.rpw = rpw;
  f.fs = a->feat_scaling; f.o_fl_w = (int)off[14]; f.o_fl_b = (int)off[15];
  f.rayin = w.rayin; f.gram = w.gram; f.hf = w.hf; f.d_hf = w.d_hf; f.rayfeat = w.rayfeat; f.X1 = w.X1; f.X2 = w.X2;
  f.params = P; f.ps = ps; f.scale = a->scale;
  f.pts = a->pts; f.origins = a->origins; f.dirs = a->dirs; f.z = a->z; f.centre = a->obj_center;
  f.gt_depth = a->gt_depth; f.gt_rgb = a->gt_rgb; f.labels = a->labels;
  f.counts = const_cast<int*>(a->counts); f.flags = const_cast<int*>(a->flags); f.self_counts = (self_counts && K == 1) ? 1 : 0;
  f.cs = a->color_scaling; f.os = a->opacity_scaling;
  f.emb = w.emb; f.h1 = w.h1; f.h2 = w.h2; f.h3 = w.h3; f.h4 = w.h4; f.hc = w.hc;
  f.d_hc = w.dA; f.d_h4 = w.dB_; f.d_h3 = w.dC; f.d_h2 = w.dD; f.d_h1 = w.dE;
  // per-workgroup partials live in workspace regions this path does not use otherwise (d_emb: K n 129 floats, dhead:
  // K n 4, loss_part: K R 4): objnerf_train_step's applicability test checked that they fit
  float* hp = w.d_emb;
  f.partA = hp; hp += (size_t)K * nwg * H;
  f.partW = hp; hp += (size_t)K * nwg * 3 * H;
  f.rsA = hp; hp += (size_t)K * nwg;
  f.rsW = hp;
  f.pe_part = w.dhead;
  f.loss_part = w.loss_part;
  f.o_in_w = (int)off[0]; f.o_in_b = (int)off[1]; f.o_m1_w = (int)off[2]; f.o_m1_b = (int)off[3];
  f.o_cat_w = (int)off[4]; f.o_cat_b = (int)off[5]; f.o_m2_w = (int)off[6]; f.o_m2_b = (int)off[7];
  f.o_a_w = (int)off[8]; f.o_a_b = (int)off[9]; f.o_cl_w = (int)off[10]; f.o_cl_b = (int)off[11];
  f.o_oc_w = (int)off[12]; f.o_oc_b = (int)off[13]; f.o_B = (int)off[18];
  if (feat) {
    if (rpw * S <= 64) launch_train_small<4, true>(st, f, nwg, bf);
    else launch_train_small<5, true>(st, f, nwg, bf);
  } else {
    if (rpw * S <= 64) launch_train_small<4, false>(st, f, nwg, bf);
    else launch_train_small<5, false>(st, f, nwg, bf);
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  if (a->relu_masks) {       // test hook: the ReLU branch bits of this iteration, from the stored activations
    const float* acts[6] = {w.h1, w.h2, w.h3, w.h4, w.hc, feat ? w.hf : nullptr};
    const long nb = (long)K * n * (H / 8);
    for (int l = 0; l < 6; ++l)
      if (acts[l])
        hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, nb, H, acts[l], a->relu_masks, l,
                           bf ? 1 : 0);
  }
  if (bf) {
    // bf16 mode: the kernel above stored h1 .. hc (hf, d_hf) and d_hc .. d_h1 as bf16 (objnerf_small_body.h, `hst`); gemm(E, )
    // recognises 16-bit operands by address and the grouped launch passes them through unconverted
    E.act16_lo = (const char*)w.h1;
    E.act16_hi = feat ? (const char*)(w.d_hf + (size_t)K * n * H) : (const char*)(w.hc + w.act_floats);
    E.grad16_lo = (const char*)w.dA; E.grad16_hi = (const char*)(w.dE + w.act_floats);
  }
  // ---- the step's reductions, collected: head partials, d B partials, the weight gradients' split-K slices
  RedGroup red;
  red.count = 0;
  {
    RedItem a0;
    a0.part = f.partA; a0.rs_part = f.rsA; a0.C = G + off[8]; a0.rowsum = G + off[9];
    a0.M = 1; a0.N = H; a0.sk = nwg; a0.batch = K; a0.scm = H; a0.bsc = ps; a0.bsrs = ps;
    RedItem a1;
    a1.part = f.partW; a1.rs_part = f.rsW; a1.C = G + off[12]; a1.rowsum = G + off[13];
    a1.M = 3; a1.N = H; a1.sk = nwg; a1.batch = K; a1.scm = H; a1.bsc = ps; a1.bsrs = ps;
    RedItem r;
    r.part = f.pe_part; r.rs_part = nullptr; r.C = G + off[18]; r.rowsum = nullptr;
    r.M = 1; r.N = 63; r.sk = nwg; r.batch = K; r.scm = 63; r.bsc = ps; r.bsrs = 0;
    red_append(red, a0); red_append(red, a1); red_append(red, r);
  }
  E.operands = 0;                   // (the grouped launch picks its operand type through group16)
  E.a_scale = 1.0f;
  GemmGroup group;
  group.count = 0;
  E.group = &group;
  E.group16 = bf;
  E.red_group = &red;
  E.need_parts = true;              // no zero-filled gradient arena here: the deterministic slab form or an error
  // Split-K slices of the grouped launch: its seven GEMMs are ONE 128 x 128 tile each, so `slices` is also the number of
  // workgroups per GEMM.  ~2 rounds of the chip (7 x 37 = 259 workgroups at least), at most ~1024 samples per slice
  // beyond that: every slice writes its partial tile to HBM and the reduction reads it back -- 128 slices at the
  // benchmark's 76 800 background samples were 48 MB each way for 0.4 MB of gradient (reduction 131 us).
  {
    long want = (n + 1023) / 1024;
    if (want < 36) want = 36;
    // whole rounds of the chip: 7 GEMMs x slices workgroups on num_cu compute units (75 slices = 525 workgroups left a
    // third round for 13 of them: 260 us against 227)
    const int cu = 256, ng = feat ? 9 : 7;
    long rounds = (want * ng + cu / 2) / cu;
    if (rounds < 1) rounds = 1;
    E.max_slices = (int)(rounds * cu / ng);
  }
  wgrad(E, st, K, H, H, n, f.d_hc, 1, H, nH, w.h4, H, 1, nH, G + off[10], H + E2, ps, G + off[11]);
  wgrad(E, st, K, H, E2, n, f.d_hc, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[10] + H, H + E2, ps);
  wgrad(E, st, K, H, H, n, f.d_h4, 1, H, nH, w.h3, H, 1, nH, G + off[6], H, ps, G + off[7]);
  wgrad(E, st, K, H, H, n, f.d_h3, 1, H, nH, w.h2, H, 1, nH, G + off[4], H + E1, ps, G + off[5]);
  wgrad(E, st, K, H, E1, n, f.d_h3, 1, H, nH, w.emb, EM, 1, n * EM, G + off[4] + H, H + E1, ps);
  wgrad(E, st, K, H, H, n, f.d_h2, 1, H, nH, w.h1, H, 1, nH, G + off[2], H, ps, G + off[3]);
  wgrad(E, st, K, H, E1, n, f.d_h1, 1, H, nH, w.emb, EM, 1, n * EM, G + off[0], E1, ps, G + off[1]);
  if (feat) {
    // feature layer
    wgrad(E, st, K, H, H, n, f.d_hf, 1, H, nH, w.h4, H, 1, nH, G + off[14], H + E2, ps, G + off[15]);
    wgrad(E, st, K, H, E2, n, f.d_hf, 1, H, nH, w.emb + E1, EM, 1, n * EM, G + off[14] + H, H + E2, ps);
  }
  if (E.parts_failed || E.error) return OBJNERF_EINVAL;
  flush_group(E, st, group);
  if (feat) {
    // the 512-d head's moments over the RAYS: T = gt_feat^T [a fh | a O], M = [c fh | c O]^T [fh | O] (featg_finish_kernel
    // turns them into d W_of, d b_of after the reduction).  Two launches of their own: inside the grouped launch their
    // 512-row output made it 5x slower (1.82 against 0.35 ms at the benchmark's background batch -- every GEMM of the
    // group is then launched over four row tiles)
    // (round 6: the two as ONE grouped launch of their own)
    GemmGroup g2;
    g2.count = 0;
    E.group = &g2; E.group16 = false; E.operands = 0; E.red_group = &red; E.max_slices = 0;
    wgrad(E, st, K, C, XC, R, a->gt_feat, 1, C, R * C, w.X1, XC, 1, R * XC, w.Tm, XC, (long)C * XC);
    wgrad(E, st, K, XC, XC, R, w.X2, 1, XC, R * XC, w.rayfeat, H + 3, 1, R * (H + 3), w.mom, XC, (long)XC * XC);
    if (E.parts_failed || E.error) return OBJNERF_EINVAL;
    RedGroup* keep = E.red_group;
    flush_group(E, st, g2);              // (launches whatever was collected; GEMMs the group does not take were launched at once)
    E.red_group = keep;
  }
  red.tail.loss_part = f.loss_part; red.tail.loss_blocks = nwg; red.tail.K = K; red.tail.loss_terms = a->loss_terms;
  red.tail.status = a->status;
  if (a->optim) {
    const objnerf_adamw_args* o = a->optim;
    RedTail& t = red.tail;
    t.params = const_cast<float*>(a->params); t.grads = G; t.m = o->exp_avg; t.v = o->exp_avg_sq; t.flags = a->flags;
    t.steps = o->group_steps; t.bank = o->bank; t.p_stride = ps; t.arena_floats = (long)K * ps;
    t.lo1 = off[10]; t.lo2 = off[14]; t.hi2 = off[18];
    t.lr = (double)o->lr; t.b1 = (double)o->beta1; t.b2 = (double)o->beta2; t.wd = (double)o->weight_decay; t.eps = o->eps;
    if (done) *done |= 2;
  }
  launch_reductions(st, red);
  if (feat && a->optim && head_snap) {
    // the head's gradient AND its AdamW step in one launch, from the copy of [W_of | b_of] (round 6)
    const objnerf_adamw_args* o = a->optim;
    FeatFinishOpt fo;
    fo.params = const_cast<float*>(a->params); fo.m = o->exp_avg; fo.v = o->exp_avg_sq; fo.flags = a->flags;
    fo.steps = o->group_steps; fo.bank = o->bank;
    fo.lr = (double)o->lr; fo.b1 = (double)o->beta1; fo.b2 = (double)o->beta2; fo.wd = (double)o->weight_decay; fo.eps = o->eps;
    hipLaunchKernelGGL(featg_finish_adamw_kernel, dim3((unsigned)(((long)C * XC + 255) / 256), K), dim3(256), 0, st, head_snap, ps,
                       (int)off[16], (int)off[17], C, H, w.Tm, w.mom, G, fo);
  } else if (feat) {
    hipLaunchKernelGGL(featg_finish_kernel, dim3((unsigned)(((long)C * XC + 255) / 256), K), dim3(256), 0, st, P, ps,
                       (int)off[16], (int)off[17], C, H, w.Tm, w.mom, G);
    if (a->optim) {               // the head's own entries [of_w, pe_b): their gradient exists only now
      const objnerf_adamw_args* o = a->optim;
      // (entries [off[16], off[18]) only: "P" = off[18] with [0, off[16]) skipped -- B's gradient behind them was stepped
      // by the reduction launch)
      const int rc = objmisc::adamw_flags_range(K, off[18], ps, const_cast<float*>(a->params), G, o->exp_avg,
                                                o->exp_avg_sq, nullptr, a->flags, o->group_steps, o->bank, off[10], off[14],
                                                off[18], 0, off[16], o->lr, o->beta1, o->beta2, o->eps, o->weight_decay,
                                                (void*)st);
      if (rc) return rc;
    }
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Inference (embedding.py:46-55 + model.py:61-103) for any hidden width: the forward half of train_step with
// ping-pong activation buffers.  workspace: emb [K][N][129] | A | B | C  ([K][N][H] each).
size_t eval_workspace_bytes(const objnerf_net* net, int K, long N) {
  return al((size_t)K * N * OBJ_EMB * 4) + 3 * al((size_t)K * N * net->hidden * 4) + 256;
}

int eval_points(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* scale,
                const float* pts, float* out_alpha, float* out_color, float* out_hfeat, float* out_clip,
                void* workspace, size_t workspace_bytes, void* stream, const float* emb_in) {
  const int H = net->hidden, C = net->feat_dim;
  if (H % 32 != 0 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if (!workspace || workspace_bytes < eval_workspace_bytes(net, K, N) - 256) return OBJNERF_EINVAL;
  GemmEnv E;                                   // fp32
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  hipStream_t st = (hipStream_t)stream;
  char* p = (char*)workspace;
  float* emb_ws = (float*)p; p += al((size_t)K * N * OBJ_EMB * 4);
  float* bA = (float*)p;  p += al((size_t)K * N * H * 4);
  float* bB = (float*)p;  p += al((size_t)K * N * H * 4);
  float* bC = (float*)p;
  const float* P = params;
  const long ps = p_stride, n = N, nH = N * H;
  const int E1 = OBJ_E1, E2 = OBJ_E2, EM = OBJ_EMB;
  const float* emb = emb_in;
  if (!emb_in) {
    int rc = objnerf_embed(net, K, N, params, p_stride, scale, pts, emb_ws, stream);
    if (rc) return rc;
    emb = emb_ws;
  }
  float *h1 = bA, *h2 = bB, *h3 = bA, *h4 = bB, *hc = bA;
  gemm(E, st, K, n, H, E1, emb, EM, 1, n * EM, P + off[0], 1, E1, ps, h1, H, 1, nH, false, P + off[1], ps, true);
  gemm(E, st, K, n, H, H, h1, H, 1, nH, P + off[2], 1, H, ps, h2, H, 1, nH, false, P + off[3], ps, true);
  gemm(E, st, K, n, H, H, h2, H, 1, nH, P + off[4], 1, H + E1, ps, h3, H, 1, nH);
  gemm(E, st, K, n, H, E1, emb, EM, 1, n * EM, P + off[4] + H, 1, H + E1, ps, h3, H, 1, nH, true, P + off[5], ps, true);
  gemm(E, st, K, n, H, H, h3, H, 1, nH, P + off[6], 1, H, ps, h4, H, 1, nH, false, P + off[7], ps, true);
  gemm(E, st, K, n, H, H, h4, H, 1, nH, P + off[10], 1, H + E2, ps, hc, H, 1, nH);
  gemm(E, st, K, n, H, E2, emb + E1, EM, 1, n * EM, P + off[10] + H, 1, H + E2, ps, hc, H, 1, nH, true, P + off[11], ps,
       true);
  dim3 eg((unsigned)((n + 15) / 16), (unsigned)K);
  hipLaunchKernelGGL(heads_fwd_kernel, eg, dim3(256), (size_t)4 * H * sizeof(float), st, H, n, h4, hc, P, ps,
                     (int)off[8], (int)off[9], (int)off[12], (int)off[13], out_alpha, out_color, 0);
  if (out_hfeat || out_clip) {
    float* hf = out_hfeat ? out_hfeat : bC;
    gemm(E, st, K, n, H, H, h4, H, 1, nH, P + off[14], 1, H + E2, ps, hf, H, 1, nH);
    gemm(E, st, K, n, H, E2, emb + E1, EM, 1, n * EM, P + off[14] + H, 1, H + E2, ps, hf, H, 1, nH, true, P + off[15], ps,
         true);
    if (out_clip)
      gemm(E, st, K, n, C, H, hf, H, 1, nH, P + off[16], 1, H, ps, out_clip, C, 1, n * C, false, P + off[17], ps, false);
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of model.py:61-103 alone (objnerf_mlp_backward_ws): what loss.backward() runs through the stacked networks
// when a caller keeps the reference's loop body (train.py:424-436) instead of the fused step.  fp32.  The activations
// are recomputed from `emb` (the forward entry keeps none), then the same dgrad / weight-gradient GEMMs as train_step's
// layer-wise chain; with d_clip the 512-d head is differentiated directly (no hoisting: the loss is the caller's).
size_t mlp_backward_workspace_bytes(const objnerf_net* net, int K, long N, int with_clip) {
  WS w = carve(nullptr, net->hidden, net->feat_dim, N, 1, K, with_clip != 0);
  size_t extra = 0;
  if (with_clip) {
    const size_t M = (size_t)net->feat_dim, Hn = (size_t)net->hidden;
    extra = al(((size_t)K * wgrad_slices(K, (int)M, (int)Hn, N) * (M * Hn + M) + 256) * 4);
  }
  return w.bytes + extra + 256;
}

int mlp_backward(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* emb,
                 const float* d_alpha, const float* d_color, const float* d_clip, float* grads, float* d_emb,
                 void* workspace, size_t workspace_bytes, void* stream) {
  const int H = net->hidden, C = net->feat_dim;
  if (H % 32 != 0 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  const bool feat = d_clip != nullptr;
  const long n = N, ps = p_stride, nH = n * H;
  if (!workspace || workspace_bytes + 256 < mlp_backward_workspace_bytes(net, K, N, feat)) return OBJNERF_EINVAL;
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  hipStream_t st = (hipStream_t)stream;
  GemmEnv E;
  WS w = carve((char*)workspace, H, C, n, 1, K, feat);
  E.parts = w.parts; E.parts_cap = w.parts_floats; E.parts_off = 0;
  const float* P = params;
  float* G = grads;
  const int E1 = OBJ_E1, E2 = OBJ_E2, EM = OBJ_EMB;
  for (int k = 0; k < K; ++k)      // feature-branch entries only when they receive a gradient ("no gradient" = untouched)
    (void)hipMemsetAsync(G + (long)k * ps, 0, (size_t)(feat ? off[18] : off[14]) * 4, st);
  // ---- forward (recompute)
  gemm(E, st, K, n, H, E1, emb, EM, 1, n * EM, P + off[0], 1, E1, ps, w.h1, H, 1, nH, false, P + off[1], ps, true);
  gemm(E, st, K, n, H, H, w.h1, H, 1, nH, P + off[2], 1, H, ps, w.h2, H, 1, nH, false, P + off[3], ps, true);
  gemm(E, st, K, n, H, H, w.h2, H, 1, nH, P + off[4], 1, H + E1, ps, w.h3, H, 1, nH);
  gemm(E, st, K, n, H, E1, emb, EM, 1, n * EM, P + off[4] + H, 1, H + E1, ps, w.h3, H, 1, nH, true, P + off[5], ps, true);
  gemm(E, st, K, n, H, H, w.h3, H, 1, nH, P + off[6], 1, H, ps, w.h4, H, 1, nH, false, P + off[7], ps, true);
  gemm(E, st, K, n, H, H, w.h4, H, 1, nH, P + off[10], 1, H + E2, ps, w.hc, H, 1, nH);
  gemm(E, st, K, n, H, E2, emb + E1, EM, 1, n * EM, P + off[10] + H, 1, H + E2, ps, w.hc, H, 1, nH, true, P + off[11], ps, true);
  dim3 eg((unsigned)((n + 15) / 16), (unsigned)K);
  const size_t head_lds = (size_t)4 * H * sizeof(float);
  hipLaunchKernelGGL(heads_fwd_kernel, eg, dim3(256), head_lds, st, H, n, w.h4, w.hc, P, ps, (int)off[8], (int)off[9],
                     (int)off[12], (int)off[13], w.alpha, w.color, 0);
  if (feat) {
    gemm(E, st, K, n, H, H, w.h4, H, 1, nH, P + off[14], 1, H + E2, ps, w.hf, H, 1, nH);
    gemm(E, st, K, n, H, E2, emb + E1, EM, 1, n * EM, P + off[14] + H, 1, H + E2, ps, w.hf, H, 1, nH, true, P + off[15], ps, true);
  }
  // ---- backward
  float *d_hc = w.dA, *d_h4 = w.dB_, *d_h3 = w.dC, *d_h2 = w.dD, *d_h1 = w.dE;
  hipLaunchKernelGGL(heads_bwd_kernel, eg, dim3(256), head_lds, st, H, n, w.hc, w.color, d_alpha, d_color, P, ps, (int)off[8],
                     (int)off[12], w.dhead, d_hc, d_h4, 0, 1.0f);
  wgrad(E, st, K, 1, H, n, w.dhead, 1, 4, n * 4, w.h4, H, 1, nH, G + off[8], H, ps, G + off[9]);
  wgrad(E, st, K, 3, H, n, w.dhead + 1, 1, 4, n * 4, w.hc, H, 1, nH, G + off[12], H, ps, G + off[13]);
  if (feat) {
    // 512-d head: d W_of = d_clip^T hf, d b_of = column sums, d_hf = (d_clip W_of) o [hf > 0]
    float* extra = (float*)((char*)workspace + w.bytes);
    float* keep = E.parts; const size_t keep_cap = E.parts_cap, keep_off = E.parts_off;
    E.parts = extra; E.parts_cap = (size_t)K * wgrad_slices(K, C, H, n) * ((size_t)C * H + C) + 256; E.parts_off = 0;
    wgrad(E, st, K, C, H, n, d_clip, 1, C, n * C, w.hf, H, 1, nH, G + off[16], H, ps, G + off[17]);
    E.parts = keep; E.parts_cap = keep_cap; E.parts_off = keep_off;
    gemm(E, st, K, n, H, C, d_clip, C, 1, n * C, P + off[16], H, 1, ps, w.d_hf, H, 1, nH, false, nullptr, 0, false, w.hf, H, 1, nH);
    wgrad(E, st, K, H, H, n, w.d_hf, 1, H, nH, w.h4, H, 1, nH, G + off[14], H + E2, ps, G + off[15]);
    wgrad(E, st, K, H, E2, n, w.d_hf, 1, H, nH, emb + E1, EM, 1, n * EM, G + off[14] + H, H + E2, ps);
    gemm(E, st, K, n, H, H, w.d_hf, H, 1, nH, P + off[14], H + E2, 1, ps, d_h4, H, 1, nH, true);
    gemm(E, st, K, n, E2, H, w.d_hf, H, 1, nH, P + off[14] + H, H + E2, 1, ps, d_emb + E1, EM, 1, n * EM, false);
  }
  wgrad(E, st, K, H, H, n, d_hc, 1, H, nH, w.h4, H, 1, nH, G + off[10], H + E2, ps, G + off[11]);
  wgrad(E, st, K, H, E2, n, d_hc, 1, H, nH, emb + E1, EM, 1, n * EM, G + off[10] + H, H + E2, ps);
  gemm(E, st, K, n, H, H, d_hc, H, 1, nH, P + off[10], H + E2, 1, ps, d_h4, H, 1, nH, true, nullptr, 0, false, w.h4, H, 1, nH);
  gemm(E, st, K, n, E2, H, d_hc, H, 1, nH, P + off[10] + H, H + E2, 1, ps, d_emb + E1, EM, 1, n * EM, feat);
  wgrad(E, st, K, H, H, n, d_h4, 1, H, nH, w.h3, H, 1, nH, G + off[6], H, ps, G + off[7]);
  gemm(E, st, K, n, H, H, d_h4, H, 1, nH, P + off[6], H, 1, ps, d_h3, H, 1, nH, false, nullptr, 0, false, w.h3, H, 1, nH);
  wgrad(E, st, K, H, H, n, d_h3, 1, H, nH, w.h2, H, 1, nH, G + off[4], H + E1, ps, G + off[5]);
  wgrad(E, st, K, H, E1, n, d_h3, 1, H, nH, emb, EM, 1, n * EM, G + off[4] + H, H + E1, ps);
  gemm(E, st, K, n, H, H, d_h3, H, 1, nH, P + off[4], H + E1, 1, ps, d_h2, H, 1, nH, false, nullptr, 0, false, w.h2, H, 1, nH);
  gemm(E, st, K, n, E1, H, d_h3, H, 1, nH, P + off[4] + H, H + E1, 1, ps, d_emb, EM, 1, n * EM, false);
  wgrad(E, st, K, H, H, n, d_h2, 1, H, nH, w.h1, H, 1, nH, G + off[2], H, ps, G + off[3]);
  gemm(E, st, K, n, H, H, d_h2, H, 1, nH, P + off[2], H, 1, ps, d_h1, H, 1, nH, false, nullptr, 0, false, w.h1, H, 1, nH);
  wgrad(E, st, K, H, E1, n, d_h1, 1, H, nH, emb, EM, 1, n * EM, G + off[0], E1, ps, G + off[1]);
  gemm(E, st, K, n, E1, H, d_h1, H, 1, nH, P + off[0], E1, 1, ps, d_emb, EM, 1, n * EM, true);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

// Backward of embedding.py:46-55 alone (objnerf_embed_bwd): d B [K][21][3] from d_emb [K][N][129]; scratch [K][64] floats
int embed_backward(const objnerf_net* net, int K, long N, const float* params, long p_stride, const float* scale,
                   const float* pts, const float* d_emb, float* d_B, float* scratch, void* stream) {
  if (net->n_freqs != 6) return OBJNERF_ENOTSUP;
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  hipStream_t st = (hipStream_t)stream;
  int pg = (int)((N + 47) / 48);
  if (pg > 1024) pg = 1024;
  if (pg < 1) pg = 1;
  (void)hipMemsetAsync(scratch, 0, (size_t)K * 64 * 4, st);
  hipLaunchKernelGGL(pe_bwd_kernel, dim3(pg, K), dim3(256), 0, st, N, params, p_stride, (int)off[18], scale, pts, d_emb, scratch,
                     (float*)nullptr);
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((K * 63 + 255) / 256)), dim3(256), 0, st, (long)K, 63, scratch, 63L, d_B,
                     63L);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

// G = W_of^T W_of, wb = W_of^T b_of, bb = b_of . b_of of K objects: gram[k][Hh * Hh | Hh | 1], batch stride gstride
// ---- the hoisted 512-d feature head around a fused kernel of another width (objnerf_generic.h)
size_t feat_head_workspace_bytes(int K, int R, int Hh, int C) {
  const size_t XC = (size_t)Hh + 1;
  size_t fl = (size_t)K * ((size_t)Hh * Hh + Hh + 1) + (size_t)K * R * (Hh + 2) + (size_t)K * R * (Hh + 3) + 2 * (size_t)K * R * XC +
              (size_t)K * C * XC + (size_t)K * XC * XC;
  const size_t parts = wgrad_parts_floats(K, C, (int)XC, R) + wgrad_parts_floats(K, (int)XC, (int)XC, R) + 256;
  return (fl + parts) * 4 + 10 * 256;
}
FeatHead feat_head_carve(char* base, int K, int R, int Hh, int C) {
  FeatHead f;
  char* p = base;
  auto take = [&](size_t floats) { float* r = (float*)p; p += al(floats * 4); return r; };
  const size_t XC = (size_t)Hh + 1;
  f.gram = take((size_t)K * ((size_t)Hh * Hh + Hh + 1));
  f.rayin = take((size_t)K * R * (Hh + 2));
  f.rayfeat = take((size_t)K * R * (Hh + 3));
  f.X1 = take((size_t)K * R * XC);
  f.X2 = take((size_t)K * R * XC);
  f.Tm = take((size_t)K * C * XC);
  f.mom = take((size_t)K * XC * XC);
  f.parts_floats = wgrad_parts_floats(K, C, (int)XC, R) + wgrad_parts_floats(K, (int)XC, (int)XC, R) + 256;
  f.parts = take(f.parts_floats);
  return f;
}
int feat_head_prep(void* stream, int K, int R, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                   const float* gt_feat, const FeatHead& f, int operands) {
  GemmEnv E;
  E.operands = operands;
  hipStream_t st = (hipStream_t)stream;
  const long gst = (long)Hh * Hh + Hh + 1;
  const float* P = params;
  gemm(E, st, K, Hh, Hh, C, P + off_w, 1, Hh, p_stride, P + off_w, Hh, 1, p_stride, f.gram, Hh, 1, gst);
  hipLaunchKernelGGL(featg_wb_kernel, dim3(Hh + 1, K), dim3(64), 0, st, P, p_stride, (int)off_w, (int)off_b, C, Hh, f.gram, gst);
  gemm(E, st, K, R, Hh, C, gt_feat, C, 1, (long)R * C, P + off_w, Hh, 1, p_stride, f.rayin, Hh + 2, 1, (long)R * (Hh + 2));
  hipLaunchKernelGGL(featg_rowstats_kernel, dim3((unsigned)((R + 15) / 16), K), dim3(256), 0, st, P, p_stride, (int)off_b, C, R,
                     Hh + 2, gt_feat, f.rayin);
  if (E.error) return OBJNERF_EINVAL;
  return hipGetLastError() == hipSuccess ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
int feat_head_grads(void* stream, int K, int R, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                    const float* gt_feat, const FeatHead& f, float* grads, int operands) {
  GemmEnv E;
  E.operands = operands;
  E.parts = f.parts; E.parts_cap = f.parts_floats; E.parts_off = 0;
  E.need_parts = true;
  hipStream_t st = (hipStream_t)stream;
  const int XC = Hh + 1;
  // T = gt_feat^T [a fh | a O], M = [c fh | c O]^T [fh | O] over the rays (X1 / X2 written by the fused kernel), then
  // d W_of = T + W_of M + b_of m^T, d b_of likewise (featg_finish_kernel)
  wgrad(E, st, K, C, XC, R, gt_feat, 1, C, (long)R * C, f.X1, XC, 1, (long)R * XC, f.Tm, XC, (long)C * XC);
  wgrad(E, st, K, XC, XC, R, f.X2, 1, XC, (long)R * XC, f.rayfeat, Hh + 3, 1, (long)R * (Hh + 3), f.mom, XC, (long)XC * XC);
  if (E.parts_failed || E.error) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(featg_finish_kernel, dim3((unsigned)(((long)C * XC + 255) / 256), K), dim3(256), 0, st, params, p_stride,
                     (int)off_w, (int)off_b, C, Hh, f.Tm, f.mom, grads);
  return hipGetLastError() == hipSuccess ? OBJNERF_OK : OBJNERF_ELAUNCH;
}

void feat_gram(void* stream, int K, const float* params, long p_stride, int off_w, int off_b, int C, int Hh, float* gram,
               long gstride) {
  GemmEnv E;
  gemm(E, (hipStream_t)stream, K, Hh, Hh, C, params + off_w, 1, Hh, p_stride, params + off_w, Hh, 1, p_stride, gram, Hh, 1,
       gstride);
  hipLaunchKernelGGL(featg_wb_kernel, dim3(Hh + 1, K), dim3(64), 0, (hipStream_t)stream, params, p_stride, off_w, off_b, C,
                     Hh, gram, gstride);
}
// out[k][m][:] = W_of[k] hfeat[k][m] + b_of[k] * weight[k][m]   (model.py:101 applied after compositing; weight NULL = 1)
void feature_head(void* stream, int K, long n, int Hh, int C, const float* params, long p_stride, long off_w, long off_b,
                  const float* hfeat, const float* weight, float* out) {
  GemmEnv E;
  E.biasrow = weight; E.bsbr = n;
  gemm(E, (hipStream_t)stream, K, (int)n, C, Hh, hfeat, Hh, 1, n * Hh, params + off_w, 1, Hh, p_stride, out, C, 1, n * C, false,
       params + off_b, p_stride, false);
}
void gemm_f32(void* stream, int batch, int M, int N, int Kd, const float* A, long sam, long sak, long bsa, const float* B,
              long sbk, long sbn, long bsb, float* C, long scm, long scn, long bsc, bool accumulate) {
  GemmEnv E;
  gemm(E, (hipStream_t)stream, batch, M, N, Kd, A, sam, sak, bsa, B, sbk, sbn, bsb, C, scm, scn, bsc, accumulate);
}
size_t wgrad_parts_floats(int batch, int M, int N, long n) {
  return (size_t)batch * wgrad_slices(batch, M, N, n) * M * N + 64;
}
void wgrad_f32(void* stream, int batch, int M, int N, long n, const float* A, long sam, long sak, long bsa, const float* B,
               long sbk, long sbn, long bsb, float* C, long scm, long bsc, float* parts, size_t parts_floats) {
  GemmEnv E;
  E.parts = parts; E.parts_cap = parts ? parts_floats : 0; E.parts_off = 0;
  wgrad(E, (hipStream_t)stream, batch, M, N, n, A, sam, sak, bsa, B, sbk, sbn, bsb, C, scm, bsc);
}

}  // namespace objgen
