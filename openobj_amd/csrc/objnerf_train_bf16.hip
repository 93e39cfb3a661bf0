// bf16-operand variant of the fused training iteration (OBJNERF_TRAIN_BF16): same tile structure as
// objnerf_train.hip (8 waves x 16 samples, whole rays, register-resident activations, LDS transposes for
// the weight gradients) but every contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
//
// One MFMA consumes a whole 32-feature block: lane (c, g) supplies k-slots 8g..8g+7 = features
// phi(g,e) = (e < 4 ? 4g + e : 16 + 4g + e - 4), i.e. exactly the 8 fp32 registers the lane holds of a
// D16-layout activation block, packed to bf16.  Weight images in LDS are stored with that column
// permutation (forward: [out][block][g][e]; transposed: [in][g][e]) so each A operand is ONE ds_read_b128;
// row pitches are 32 B x odd (mod 256 B), which the 16-lane ds_read_b128 groups read conflict-free.
// Master weights, embedding, activations, compositing, losses, gradient accumulators stay fp32; only MFMA
// operands (weights, activations, staged transposes) are rounded to bf16 (round-to-nearest-even).
// The reference is fp32-only (train.py:74 AMP = False): this path is gated by PSNR, not by 1e-4.
#define OBJ_HW_SINCOS 1      // embedding sin / cos on the transcendental unit (see objnerf_device.h)
// Positional encoding: the direction-owner layout of the second-generation fp32 kernel (objnerf_mlp32.h): lane group g
// owns directions 4 i + g in all octaves, so a sample's projection gradients are complete inside the owning lane (no
// cross-group sum on the matrix core, no fp32 table in LDS) and d B accumulates in 18 registers.  Only octave 0 gets
// // its own ... (placeholder)
#include "objnerf_bf16_common.h"
#include "../../include/objnerf_hip.h"

namespace objtrain {
namespace {
using namespace bf16k;

// LDS byte layout (the forward images: objnerf_bf16_common.h)
constexpr int RST = 96;                                            // transposed images: 64 B of outs per input row
constexpr int T_IN = FWD_IMG_END;             // 96 rows
constexpr int T_M1 = T_IN + 96 * RST;         // 32 rows
constexpr int T_CAT = T_M1 + 32 * RST;        // 128 rows: h2 (32) | x1 (96)
constexpr int T_M2 = T_CAT + 128 * RST;       // 32 rows
constexpr int T_CL = T_M2 + 32 * RST;         // 80 rows:  h4 (32) | x2 (48)
constexpr int B_FL = T_CL + 80 * RST;         // feature layer, forward image (32 rows, as B_CL)
constexpr int T_FL = B_FL + 32 * RS_CL;       // feature layer, transposed image (80 rows, as T_CL)
constexpr int B_SMALL = T_FL + 80 * RST;      // fp32: bm1[32] bm2[32] wa[32] woc[96] hb[4] peb[99]
constexpr int B_SM = B_SMALL + 1280;          // s_alpha | s_col[3]  (fp32, 2 KB)
constexpr int STG_PITCH = 288;                // bf16 staging row: 128 samples + pad
constexpr int B_STG = (B_SM + 2048 + 15) / 16 * 16;
constexpr int STG_ROWS_B = 192;
// feature branch, fp32: s_w [128] | gfh [16][32] | gof [32]  (live from the compositing into phase A)
constexpr int B_FEAT = B_STG + STG_ROWS_B * STG_PITCH;
constexpr int LDS_BYTES = B_FEAT + (128 + 16 * 32 + 32) * 4;
// fp32 aliases inside the staging area, live from the forward pass to the end of the compositing (float offsets)
constexpr int FA_HF = 0;                      // hidden feature of the tile [128][33]
constexpr int FA_G = TS * HF_LD;              // Gram matrix [32][33] | wb [32] | bb
constexpr int FA_FHB = FA_G + 32 * 33 + 64;   // fh exchange buffer [NWAVE][64]
static_assert((FA_FHB + 64 * NWAVE) * 4 <= STG_ROWS_B * STG_PITCH, "bf16 feat aliases");
static_assert(B_SMALL % 16 == 0 && B_STG % 16 == 0 && LDS_BYTES <= 163840, "bf16 lds layout");


__device__ __forceinline__ void stage_weights_bf16(char* lds, const float* __restrict__ P, const Layout& L, int tid,
                                                   const bool feat) {
  __bf16* img = reinterpret_cast<__bf16*>(lds);
  stage_forward_bf16(lds, reinterpret_cast<float*>(lds + B_SMALL), P, L, tid, feat, B_FL);
  // transposed images: element (f, g, e) <- W[phi(g, e)][f]
  for (int x = tid; x < 128 * 32; x += NTHR) {
    const int f = x >> 5, ge = x & 31, o = phi(ge >> 3, ge & 7);
    img[(T_CAT + f * RST) / 2 + ge] = (__bf16)w_cat(P, L, o, f);
    if (f < 96) img[(T_IN + f * RST) / 2 + ge] = (__bf16)w_in(P, L, o, f);
    if (f < 80) img[(T_CL + f * RST) / 2 + ge] = (__bf16)w_cl(P, L, o, f);
    if (feat && f < 80) img[(T_FL + f * RST) / 2 + ge] = (__bf16)w_fl(P, L, o, f);
    if (f < 32) {
      img[(T_M1 + f * RST) / 2 + ge] = (__bf16)P[L.m1_w + o * H + f];
      img[(T_M2 + f * RST) / 2 + ge] = (__bf16)P[L.m2_w + o * H + f];
    }
  }
}

// acc (input rows 16 ti .. +15 of the transposed image) += W^T d      timg_lane = lds + T_X + c * RST + 16 g
__device__ __forceinline__ void bwd_tile(f32x4& acc, const char* timg_lane, const int row0, const bf16x8 db) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(timg_lane + row0 * RST);
  acc = MFMA_BF16(a, db, acc);
}

__device__ __forceinline__ void store32_b(char* stg_lane, const int rowbase, const T32& v) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<__bf16*>(stg_lane + (rowbase + 16 * tt + r) * STG_PITCH) = (__bf16)v.t[tt][r];
}
__device__ __forceinline__ void store16_b(char* stg_lane, const int rowbase, const f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<__bf16*>(stg_lane + (rowbase + r) * STG_PITCH) = (__bf16)v[r];
}

// D[out 0..31][in 16 cols] += sum over the 128 staged samples; k-slot (step, g, e) = sample 32 step + 8 g + e
__device__ __forceinline__ void wgrad_pair_b(f32x4& acc0, f32x4& acc1, const char* dT, const char* aT) {
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(aT + 64 * st);
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(dT + 64 * st);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(dT + 16 * STG_PITCH + 64 * st);
    acc0 = MFMA_BF16(a0, b, acc0);
    acc1 = MFMA_BF16(a1, b, acc1);
  }
}

// hid: number of hidden-feature rows in front of the embedding rows of this operand; x2: which embedding half
__device__ __forceinline__ void write_pair_b(float* slab, const f32x4& a0, const f32x4& a1, const int c, const int g,
                                             const int ct, const int w_off, const int ncols, const int b_off,
                                             const int hid = 1 << 20, const bool x2 = false) {
  const int rho = 16 * ct + c;
  int col = rho;
  if (rho >= hid) {
    int t_, g_;
    obj32n::kappa_tg(rho - hid, t_, g_);
    col = x2 ? obj32n::x2_col(t_, g_) : obj32n::x1_col(t_, g_);
    if (col >= 0) col += hid;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o0 = 4 * g + r, o1 = 16 + 4 * g + r;
    if (col >= 0) {
      slab[w_off + o0 * ncols + col] = a0[r];
      slab[w_off + o1 * ncols + col] = a1[r];
    } else if (col == obj32n::BIAS_COL && b_off >= 0) {
      slab[b_off + o0] = a0[r];
      slab[b_off + o1] = a1[r];
    }
  }
}

#ifdef PHASE_TIMING
__device__ unsigned long long g_phase_b[8][24];
#define PT_INIT() unsigned long long pt_acc[18]; for (int i_ = 0; i_ < 18; ++i_) pt_acc[i_] = 0; \
  unsigned long long pt_t0 = __builtin_amdgcn_s_memtime()
#define PT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_acc[i] += t_ - pt_t0; pt_t0 = t_; } while (0)
#define PT_FLUSH() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 18; ++i_) g_phase_b[w][i_] = pt_acc[i_]; } while (0)
#else
#define PT_INIT() do {} while (0)
#define PT(i) do {} while (0)
#define PT_FLUSH() do {} while (0)
#endif

// SS: samples per ray when known at compile time (64 = the metric shape: no integer divisions by S), 0 = any S <= 64
template <bool FEAT, int SS>
__global__ __launch_bounds__(NTHR) void train_fused_bf16_kernel(const TrainDev a_) {
  TrainDev a = a_;
  if (SS) { a.S = SS; a.TR = TS / SS; }
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int k = blockIdx.x / a.G, gi = blockIdx.x % a.G;
  const float* sm = reinterpret_cast<const float*>(ldsb + B_SMALL);
  float* s_alpha = reinterpret_cast<float*>(ldsb + B_SM);
  float* s_col = s_alpha + TS;
  char* stg = ldsb + B_STG;

  for (int i = tid; i < LDS_BYTES / 4; i += NTHR) reinterpret_cast<float*>(ldsb)[i] = 0.0f;
  __syncthreads();
  stage_weights_bf16(ldsb, a.params + (long)k * a.p_stride, a.L, tid, FEAT);
  __syncthreads();

  const float inv_scale = 1.0f / a.scale[k];
  const int S = SS ? SS : a.S, R = a.R, TR = SS ? TS / SS : a.TR;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  int bflag0, bflag1;
  batch_flags(a, bflag0, bflag1);
  const float inv1 = bflag0 ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = bflag1 ? 0.0f : 1.0f / (n2 + 1e-10f);

  f32x4 accA0 = zero4(), accA1 = zero4(), accB0 = zero4(), accB1 = zero4(), accC0 = zero4(), accC1 = zero4();
  // Row sums over the samples (head weights, mid1 / mid2 biases): transposing DPP butterflies into slot registers per
  // tile, or -- where the registers allow (LAZY_*) -- per-lane partial sums reduced once at the end of the sweep
  // (the butterflies are ~150 select / DPP instructions per tile in a VALU-bound kernel).
#ifdef OBJ_BF16_BUTTERFLY
  constexpr bool LAZY_H = false, LAZY_B = false;
#else
  constexpr bool LAZY_H = !FEAT || SS != 0, LAZY_B = !FEAT;     // (the any-S feature build has no registers to spare)
#endif
  float gS0 = 0.f, gS1 = 0.f, gS2 = 0.f;
  float hW[4][8], bS[2][8];
#pragma unroll
  for (int s_ = 0; s_ < 8; ++s_) hW[0][s_] = hW[1][s_] = hW[2][s_] = hW[3][s_] = bS[0][s_] = bS[1][s_] = 0.f;
  float g_ba = 0.f, g_boc0 = 0.f, g_boc1 = 0.f, g_boc2 = 0.f;
  float dB[6][3];                           // d B[4 i + g][x], summed over this lane's samples
#pragma unroll
  for (int i = 0; i < 6; ++i) dB[i][0] = dB[i][1] = dB[i][2] = 0.f;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f, l_f = 0.f;
  f32x4 accF0 = zero4(), accF1 = zero4();
  float* stgf = reinterpret_cast<float*>(stg);                       // fp32 view (feature aliases)
  float* s_w = reinterpret_cast<float*>(ldsb + B_FEAT);
  float* s_gfh = s_w + 128;
  float* s_gof = s_gfh + 16 * 32;


  const bool rows_mode = SS ? true : seg_is_rows(a.S);
  const SegRows seg_rows = SegRows::make(rows_mode ? S : 64, lane);
  // sample position of (tile, slot); issued one tile ahead (phase C), as in objnerf_train.hip
  auto fetch_point = [&](const int tile_, const int slot_, float& x, float& y, float& z_) {
    const int q_ = slot_ / S, si_ = slot_ - q_ * S;
    const int ray_ = tile_ * TR + q_;
    x = 0.f; y = 0.f; z_ = 0.f;
    if (tile_ < a.NT && q_ < TR && ray_ < a.R) {
      const long rr = (long)k * a.R + ray_;
      if (a.pts) {
        const float* p = a.pts + (rr * S + si_) * 3;
        x = p[0]; y = p[1]; z_ = p[2];
      } else {
        const float zz = a.z[rr * S + si_];
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        x = (o[0] + d[0] * zz) - a.obj_center;
        y = (o[1] + d[1] * zz) - a.obj_center;
        z_ = (o[2] + d[2] * zz) - a.obj_center;
      }
    }
  };
  // (c, g) and every per-lane LDS pointer are macros over an opaque copy of the lane id that is re-defined at the
  // phase boundaries: addresses are recomputed next to their use instead of living in (spilled) registers
  int lane_l = lane;
#define RELAUNDER() do { if (FEAT) asm volatile("" : "+v"(lane_l)); } while (0)   // (only the feature build is short of registers)
#define c (lane_l & 15)
#define g (lane_l >> 4)
#define stg_lane (stg + (4 * g) * STG_PITCH + (16 * w + c) * 2)
#define lane_rd (stg + c * STG_PITCH + 16 * g)
#define f_in (ldsb + B_IN + c * RS_IN + 16 * g)
#define f_m1 (ldsb + B_M1 + c * RS_M + 16 * g)
#define f_cat (ldsb + B_CAT + c * RS_CAT + 16 * g)
#define f_m2 (ldsb + B_M2 + c * RS_M + 16 * g)
#define f_cl (ldsb + B_CL + c * RS_CL + 16 * g)
#define t_in (ldsb + T_IN + c * RST + 16 * g)
#define t_m1 (ldsb + T_M1 + c * RST + 16 * g)
#define t_cat (ldsb + T_CAT + c * RST + 16 * g)
#define t_m2 (ldsb + T_M2 + c * RST + 16 * g)
#define t_cl (ldsb + T_CL + c * RST + 16 * g)
#define f_fl (ldsb + B_FL + c * RS_CL + 16 * g)
#define t_fl (ldsb + T_FL + c * RST + 16 * g)
  float nx, ny, nz;
  fetch_point(gi, 16 * w + c, nx, ny, nz);
  PT_INIT();
  for (int tile = gi; tile < a.NT; tile += a.G) {
    asm volatile("" ::: "memory");
    RELAUNDER();
    const int ray0 = tile * TR;
    // ---------------------------------------------------------------- 1. forward
    const int slot = 16 * w + c;
    const int q = slot / S;
    const int ray = ray0 + q;
    const bool valid = (q < TR) && (ray < R);
    const float px = nx, py = ny, pz = nz;       // fetched during the previous tile's phase C
    obj32n::Pe32 pe;
    pe_project_b(sm, g, px, py, pz, inv_scale, pe);
    PT(0);
    T32 h1, h2, h3, h4, hc;
    T32 hf = zero32();           // (feature build) dead after the forward: the backward keeps its sign mask only
    unsigned hf_mask = 0;
    float alpha_v, col_v[3];
    {
      bf16x8 xb1[3], xb2[2];
      {
        obj32n::Emb32 e;               // forward-only: the backward re-creates the embedding tile by tile
        obj32n::embed32(e, pe, g);
#pragma unroll
        for (int b = 0; b < 3; ++b) xb1[b] = pack8(e.x1[2 * b], e.x1[2 * b + 1]);
        xb2[0] = pack8(e.x2[0], e.x2[1]);
        xb2[1] = pack8(e.x2[2], zero4());
      }
      T32 acc = zero32();
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_IN>(acc, f_in, b, xb1[b]);
      h1 = relu32(acc);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc.t[tt][r] = sm[S_BM1 + 16 * tt + 4 * g + r];
      fwd_blk<RS_M>(acc, f_m1, 0, pack32(h1));
      h2 = relu32(acc);
      acc = zero32();
      fwd_blk<RS_CAT>(acc, f_cat, 0, pack32(h2));
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_CAT>(acc, f_cat, 1 + b, xb1[b]);
      h3 = relu32(acc);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc.t[tt][r] = sm[S_BM2 + 16 * tt + 4 * g + r];
      fwd_blk<RS_M>(acc, f_m2, 0, pack32(h3));
      h4 = relu32(acc);
      acc = zero32();
      fwd_blk<RS_CL>(acc, f_cl, 0, pack32(h4));
      fwd_blk<RS_CL>(acc, f_cl, 1, xb2[0]);
      fwd_blk<RS_CL>(acc, f_cl, 2, xb2[1]);
      hc = relu32(acc);
      if (FEAT) {
        acc = zero32();
        fwd_blk<RS_CL>(acc, f_fl, 0, pack32(h4));
        fwd_blk<RS_CL>(acc, f_fl, 1, xb2[0]);
        fwd_blk<RS_CL>(acc, f_fl, 2, xb2[1]);
        hf = relu32(acc);
      }
      float pa = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tt + 4 * g + r;
          pa = fmaf(sm[S_WA + row], h4.t[tt][r], pa);
          pc0 = fmaf(sm[S_WOC + row], hc.t[tt][r], pc0);
          pc1 = fmaf(sm[S_WOC + H + row], hc.t[tt][r], pc1);
          pc2 = fmaf(sm[S_WOC + 2 * H + row], hc.t[tt][r], pc2);
        }
      alpha_v = (xgroup_sum(pa) + sm[S_HB]) * 10.0f;
      col_v[0] = sigmoid_acc(xgroup_sum(pc0) + sm[S_HB + 1]);
      col_v[1] = sigmoid_acc(xgroup_sum(pc1) + sm[S_HB + 2]);
      col_v[2] = sigmoid_acc(xgroup_sum(pc2) + sm[S_HB + 3]);
    }
    if (g == 0) {
      s_alpha[slot] = alpha_v;
      s_col[slot] = col_v[0];
      s_col[TS + slot] = col_v[1];
      s_col[2 * TS + slot] = col_v[2];
    }
    if (FEAT) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          stgf[FA_HF + slot * HF_LD + 16 * tt + 4 * g + r] = hf.t[tt][r];
          hf_mask |= (hf.t[tt][r] > 0.0f ? 1u : 0u) << (4 * tt + r);      // all the backward needs of hf
        }
      // this object's Gram matrix (+ wb, bb) for the tile.  Fixed trip count: a real loop in the middle of the tile
      // body splits every live range around it and costs ~80 spilled registers.
#pragma unroll
      for (int it = 0; it < (32 * 32 + 33 + NTHR - 1) / NTHR; ++it) {
        const int i = tid + NTHR * it;
        if (i < 32 * 32 + 33) {
          const float v = a.gram[(long)k * GRAM + i];
          if (i < 1024) stgf[FA_G + (i >> 5) * 33 + (i & 31)] = v;
          else stgf[FA_G + 32 * 33 + (i - 1024)] = v;
        }
      }
    }
    // ray inputs of this wave's compositing pass, requested BEFORE the barrier so their latency hides behind it
    auto ray_inputs = [&](const int ps_, float& zz_, float& gtd_, float& gr_, float& gg_, float& gb_, int& lab_) {
      const int rpp_ = 64 / S;
      const int ql_ = lane_l / S, pos_ = lane_l - ql_ * S;
      const int qq_ = ps_ * rpp_ + ql_;
      const int rayq_ = ray0 + qq_;
      zz_ = 0.f; gtd_ = 0.f; gr_ = 0.f; gg_ = 0.f; gb_ = 0.f; lab_ = 2;
      if ((ql_ < rpp_) && (qq_ < TR) && (rayq_ < R)) {
        const long rr = (long)k * R + rayq_;
        zz_ = a.z[rr * S + pos_];
        gtd_ = a.gt_depth[rr];
        gr_ = a.gt_rgb[rr * 3]; gg_ = a.gt_rgb[rr * 3 + 1]; gb_ = a.gt_rgb[rr * 3 + 2];
        lab_ = a.labels[rr];
      }
    };
    float pf_zz = 0.f, pf_gtd = 0.f, pf_gr = 0.f, pf_gg = 0.f, pf_gb = 0.f;
    int pf_lab = 2;
    if (w * (64 / S) < TR) ray_inputs(w, pf_zz, pf_gtd, pf_gr, pf_gg, pf_gb, pf_lab);
    auto feat_inputs = [&](const int ps_, const int qb_, float& uh_, float& beta_, float& ngv_, int& lab_) {
      const int rpp_ = 64 / S;
      const int ql2_ = qb_ + (lane_l >> 5);
      const int qq2_ = ps_ * rpp_ + ql2_;
      const int ray2_ = ray0 + qq2_;
      uh_ = 0.f; beta_ = 0.f; ngv_ = 1.f; lab_ = 2;
      if ((ql2_ < rpp_) && (qq2_ < TR) && (ray2_ < R)) {
        const long rr2_ = (long)k * R + ray2_;
        uh_ = a.rayin[rr2_ * RAYIN + (lane_l & 31)];
        beta_ = a.rayin[rr2_ * RAYIN + 32];
        ngv_ = a.rayin[rr2_ * RAYIN + 33];
        lab_ = (int)a.labels[rr2_];
      }
    };
    float pf_uh = 0.f, pf_beta = 0.f, pf_ngv = 1.f;
    int pf_lab2 = 2;
    if (FEAT && S == 64) {
      if (valid) {        // every wave takes part in its ray's feature term (below)
        const long rr2_ = (long)k * R + ray;
        pf_uh = a.rayin[rr2_ * RAYIN + (lane_l & 31)];
        pf_beta = a.rayin[rr2_ * RAYIN + 32];
        pf_ngv = a.rayin[rr2_ * RAYIN + 33];
        pf_lab2 = (int)a.labels[rr2_];
      }
    } else if (FEAT && w * (64 / S) < TR) feat_inputs(w, 0, pf_uh, pf_beta, pf_ngv, pf_lab2);
    PT(2);
    __syncthreads();
    RELAUNDER();
    PT(3);
    // ---------------------------------------------------------------- 2. composite + loss (fp32, as objnerf_train.hip)
    auto composite_passes = [&](const auto& sg) {
      const int rpp = 64 / S;
      const int npass = (TR + rpp - 1) / rpp;
      for (int ps = w; ps < npass; ps += NWAVE) {
        const int ql = lane_l / S, pos = lane_l - ql * S;
        const int qq = ps * rpp + ql;
        const int rayq = ray0 + qq;
        const bool on = (ql < rpp) && (qq < TR) && (rayq < R);
        const int sl = qq * S + pos;
        float al = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, zz = pf_zz;
        float gtd = pf_gtd, gr = pf_gr, gg = pf_gg, gb = pf_gb;
        int lab = pf_lab;
        if (ps != w) ray_inputs(ps, zz, gtd, gr, gg, gb, lab);      // (only when a tile has more than 8 passes)
        if (on) { al = s_alpha[sl]; c0 = s_col[sl]; c1 = s_col[TS + sl]; c2 = s_col[2 * TS + sl]; }
        const float occ = on ? sigmoid_acc(al) : 0.0f;
        const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;
        const float Pinc = sg.scan_mul(fr, pos);
        float T = __shfl_up(Pinc, 1, 64);
        if (pos == 0) T = 1.0f;
        const float wgt = occ * T;
        const float D = sg.total_add(wgt * zz, pos);
        const float O = sg.total_add(wgt, pos);
        const float C0 = sg.total_add(wgt * c0, pos);
        const float C1 = sg.total_add(wgt * c1, pos);
        const float C2 = sg.total_add(wgt * c2, pos);
        const float dz = zz - D;
        const float V = sg.total_add(wgt * (dz * dz), pos);
        const float m1 = (lab == 1) ? 1.0f : 0.0f;
        const float m2 = (lab != 2) ? 1.0f : 0.0f;
        const float tgt = (lab != 0) ? 1.0f : 0.0f;
        const float info = 1.0f / (sqrtf(V) + 1e-4f);
        const float rd = D - gtd, r0 = C0 - gr, r1 = C1 - gg, r2 = C2 - gb, ro = O - tgt;
        auto sgn = [](float x) { return x > 0.f ? 1.0f : (x < 0.f ? -1.0f : 0.0f); };
        const float gD = m1 * sgn(rd) * info * inv1;
        const float gC0 = a.color_scaling * m1 * sgn(r0) * inv1;
        const float gC1 = a.color_scaling * m1 * sgn(r1) * inv1;
        const float gC2 = a.color_scaling * m1 * sgn(r2) * inv1;
        const float gO = a.opacity_scaling * m2 * sgn(ro) * inv2;
        if (on && pos == 0) {
          l_d += m1 * fabsf(rd) * info * inv1;
          l_c += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
          l_o += m2 * fabsf(ro) * inv2;
        }
        float dw = gD * zz + gO + gC0 * c0 + gC1 * c1 + gC2 * c2;
        if (FEAT) {
          // feature-distillation term with the 512-d head hoisted past the compositing (objnerf_train.hip, 4.3 of
          // DESIGN.md): fp32 throughout, only the hidden feature itself came out of bf16 MFMAs
          float* s_fhb = stgf + FA_FHB + 64 * w;
          const float* Gb = stgf + FA_G;
          if (on) s_w[sl] = wgt;
          if (on && pos == 0) s_gof[16 + qq] = O;
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
          const int half = lane_l >> 5, hh = lane_l & 31;
          for (int qb = 0; qb < rpp; qb += 2) {
            const int ql2 = qb + half;
            const int qq2 = ps * rpp + ql2;
            const int ray2 = ray0 + qq2;
            const bool on2 = (ql2 < rpp) && (qq2 < TR) && (ray2 < R);
            const long rr2 = (long)k * R + (on2 ? ray2 : 0);
            float fh = 0.f;
            if (rpp == 1) {
              const bool on1 = (qb < rpp) && (ps * rpp + qb < TR) && (ray0 + ps * rpp + qb < R);
              const int q1 = ps * rpp + qb;
              if (on1) {
                float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
                const float* wp = s_w + q1 * S;
                const float* hp = stgf + FA_HF + (q1 * S) * HF_LD + hh;
                int s2 = half;
                for (; s2 + 6 < S; s2 += 8) {
                  f0 = fmaf(wp[s2], hp[s2 * HF_LD], f0);
                  f1 = fmaf(wp[s2 + 2], hp[(s2 + 2) * HF_LD], f1);
                  f2 = fmaf(wp[s2 + 4], hp[(s2 + 4) * HF_LD], f2);
                  f3 = fmaf(wp[s2 + 6], hp[(s2 + 6) * HF_LD], f3);
                }
                for (; s2 < S; s2 += 2) f0 = fmaf(wp[s2], hp[s2 * HF_LD], f0);
                fh = (f0 + f1) + (f2 + f3);
              }
              fh += __shfl_xor(fh, 32, 64);
            } else if (on2) {
              for (int s2 = 0; s2 < S; ++s2) fh = fmaf(s_w[qq2 * S + s2], stgf[FA_HF + (qq2 * S + s2) * HF_LD + hh], fh);
            }
            s_fhb[half * 32 + hh] = fh;
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            float Gfh = 0.f;
#pragma unroll 8
            for (int h2 = 0; h2 < 32; ++h2) Gfh = fmaf(Gb[hh * 33 + h2], s_fhb[half * 32 + h2], Gfh);
            const float wbh = Gb[32 * 33 + hh], bb = Gb[32 * 33 + 32];
            float uh = pf_uh, beta = pf_beta, ngv = pf_ngv;
            int lab2 = pf_lab2;
            if (ps != w || qb != 0) feat_inputs(ps, qb, uh, beta, ngv, lab2);
            const float O2 = on2 ? s_gof[16 + qq2] : 0.f;
            const float fu = wave_sum32(fh * uh), fGf = wave_sum32(fh * Gfh), fwb = wave_sum32(fh * wbh);
            const float dotFg = fu + O2 * beta;
            const float nF2 = fmaxf(fGf + 2.0f * O2 * fwb + O2 * O2 * bb, 0.0f);
            const float nF = fmaxf(sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
            const float cosv = dotFg / (nF * ngc);
            const float mm1 = (lab2 == 1) ? 1.0f : 0.0f;
            const float gam = -a.feat_scaling * mm1 * inv1;
            const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
            if (on2) {
              if (hh == 0) {
                l_f += mm1 * (1.0f - cosv) * inv1;
                s_gof[qq2] = ar * beta + cr * (fwb + O2 * bb);
                a.rayfeat[rr2 * RAYFEAT + 32] = O2;
                a.rayfeat[rr2 * RAYFEAT + 33] = ar;
                a.rayfeat[rr2 * RAYFEAT + 34] = cr;
              }
              s_gfh[qq2 * 32 + hh] = ar * uh + cr * (Gfh + O2 * wbh);
              a.rayfeat[rr2 * RAYFEAT + hh] = fh;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
          }
          if (on) {
            float dwf = s_gof[qq];
#pragma unroll 8
            for (int h2 = 0; h2 < 32; ++h2) dwf = fmaf(s_gfh[qq * 32 + h2], stgf[FA_HF + sl * HF_LD + h2], dwf);
            dw += dwf;
          }
        }
        const float qv = dw * wgt;
        const float suf = sg.rscan_add(qv, pos) - qv;
        const float docc = dw * T - suf / fr;
        if (on) {
          s_alpha[sl] = 10.0f * (docc * occ * (1.0f - occ));
          s_col[sl] = gC0 * wgt * c0 * (1.0f - c0);
          s_col[TS + sl] = gC1 * wgt * c1 * (1.0f - c1);
          s_col[2 * TS + sl] = gC2 * wgt * c2 * (1.0f - c2);
        }
      }
    };
    if (FEAT && S == 64) {
      // 64 samples per ray (the north-star shape): the feature term's reductions over samples and hidden features
      // are spread over all 8 waves instead of running on the two compositing waves (3 extra barriers, ~5x shorter
      // critical path).  Wave w holds samples 16w..16w+15 of ray w >> 2.
      const SegRows& sg = seg_rows;
      const int pos = lane_l;
      const int sl = w * 64 + pos;
      const bool on = (w < TR) && (ray0 + w < R);
      float occ = 0.f, fr = 1.f, T = 1.f, wgt = 0.f, dw = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, gC0 = 0.f, gC1 = 0.f, gC2 = 0.f;
      if (w < TR) {
        const float zz = pf_zz;
        float al = 0.f;
        if (on) { al = s_alpha[sl]; c0 = s_col[sl]; c1 = s_col[TS + sl]; c2 = s_col[2 * TS + sl]; }
        occ = on ? sigmoid_acc(al) : 0.0f;
        fr = on ? (1.0f - occ) + 1e-10f : 1.0f;
        const float Pinc = sg.scan_mul(fr, pos);
        T = __shfl_up(Pinc, 1, 64);
        if (pos == 0) T = 1.0f;
        wgt = occ * T;
        const float D = sg.total_add(wgt * zz, pos);
        const float O = sg.total_add(wgt, pos);
        const float C0 = sg.total_add(wgt * c0, pos);
        const float C1 = sg.total_add(wgt * c1, pos);
        const float C2 = sg.total_add(wgt * c2, pos);
        const float dz = zz - D;
        const float V = sg.total_add(wgt * (dz * dz), pos);
        const float m1 = (pf_lab == 1) ? 1.0f : 0.0f;
        const float m2 = (pf_lab != 2) ? 1.0f : 0.0f;
        const float tgt = (pf_lab != 0) ? 1.0f : 0.0f;
        const float info = 1.0f / (sqrtf(V) + 1e-4f);
        const float rd = D - pf_gtd, r0 = C0 - pf_gr, r1 = C1 - pf_gg, r2 = C2 - pf_gb, ro = O - tgt;
        auto sgn = [](float x) { return x > 0.f ? 1.0f : (x < 0.f ? -1.0f : 0.0f); };
        const float gD = m1 * sgn(rd) * info * inv1;
        gC0 = a.color_scaling * m1 * sgn(r0) * inv1;
        gC1 = a.color_scaling * m1 * sgn(r1) * inv1;
        gC2 = a.color_scaling * m1 * sgn(r2) * inv1;
        const float gO = a.opacity_scaling * m2 * sgn(ro) * inv2;
        if (on && pos == 0) {
          l_d += m1 * fabsf(rd) * info * inv1;
          l_c += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
          l_o += m2 * fabsf(ro) * inv2;
          s_gof[16 + w] = O;
        }
        dw = gD * zz + gO + gC0 * c0 + gC1 * c1 + gC2 * c2;
        if (on) s_w[sl] = wgt;
      }
      __syncthreads();
      float* s_part = s_gfh + 192;          // [NWAVE][32] partial composited features (s_gfh holds 2 rays here)
      float* s_dwf = s_gfh + 64;            // [128]
      {
        const float wv = valid ? s_w[slot] : 0.0f;
        float v8[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v8[4 * tt + r] = wv * stgf[FA_HF + slot * HF_LD + 16 * tt + 4 * g + r];
        const float psum = slot_sums8(v8, c);
        if (c < 8) s_part[w * 32 + 16 * ((c & 7) >> 2) + 4 * g + (c & 3)] = psum;
      }
      __syncthreads();
      {
        float* s_fhb = stgf + FA_FHB + 64 * w;
        const float* Gb = stgf + FA_G;
        const int half = lane_l >> 5, hh = lane_l & 31;
        const int w0 = w & ~3;
        const float fh = valid ? (s_part[w0 * 32 + hh] + s_part[(w0 + 1) * 32 + hh]) +
                                 (s_part[(w0 + 2) * 32 + hh] + s_part[(w0 + 3) * 32 + hh]) : 0.0f;
        if (half == 0) s_fhb[hh] = fh;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        float Gp = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 16; ++h2) Gp = fmaf(Gb[hh * 33 + 16 * half + h2], s_fhb[16 * half + h2], Gp);
        const float Gfh = Gp + __shfl_xor(Gp, 32, 64);
        const float wbh = Gb[32 * 33 + hh], bb = Gb[32 * 33 + 32];
        const float uh = pf_uh, beta = pf_beta, ngv = pf_ngv;
        const float O2 = valid ? s_gof[16 + q] : 0.f;
        const float fu = wave_sum32(fh * uh), fGf = wave_sum32(fh * Gfh), fwb = wave_sum32(fh * wbh);
        const float dotFg = fu + O2 * beta;
        const float nF2 = fmaxf(fGf + 2.0f * O2 * fwb + O2 * O2 * bb, 0.0f);
        const float nF = fmaxf(__builtin_amdgcn_sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
        const float rnn = __builtin_amdgcn_rcpf(nF * ngc);       // (bf16 mode: 1-ulp hardware reciprocals)
        const float cosv = dotFg * rnn;
        const float mm1 = (pf_lab2 == 1) ? 1.0f : 0.0f;
        const float gam = -a.feat_scaling * mm1 * inv1;
        const float ar = gam * rnn, cr = -gam * cosv * __builtin_amdgcn_rcpf(nF * nF);
        const float gfh = ar * uh + cr * (Gfh + O2 * wbh);
        const float gof = ar * beta + cr * (fwb + O2 * bb);
        if (valid && (w & 3) == 0 && half == 0) {
          const long rr2 = (long)k * R + ray;
          if (hh == 0) {
            l_f += mm1 * (1.0f - cosv) * inv1;
            a.rayfeat[rr2 * RAYFEAT + 32] = O2;
            a.rayfeat[rr2 * RAYFEAT + 33] = ar;
            a.rayfeat[rr2 * RAYFEAT + 34] = cr;
          }
          s_gfh[q * 32 + hh] = gfh;
          a.rayfeat[rr2 * RAYFEAT + hh] = fh;
        }
        if (half == 0) s_fhb[32 + hh] = gfh;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        float dp = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dp = fmaf(s_fhb[32 + 16 * tt + 4 * g + r], stgf[FA_HF + slot * HF_LD + 16 * tt + 4 * g + r], dp);
        const float dwf = xgroup_sum(dp) + gof;
        if (g == 0 && valid) s_dwf[slot] = dwf;
      }
      __syncthreads();
      if (w < TR) {
        if (on) dw += s_dwf[sl];
        const float qv = dw * wgt;
        const float suf = sg.rscan_add(qv, pos) - qv;
        const float docc = dw * T - suf / fr;
        if (on) {
          s_alpha[sl] = 10.0f * (docc * occ * (1.0f - occ));
          s_col[sl] = gC0 * wgt * c0 * (1.0f - c0);
          s_col[TS + sl] = gC1 * wgt * c1 * (1.0f - c1);
          s_col[2 * TS + sl] = gC2 * wgt * c2 * (1.0f - c2);
        }
      }
    } else if (SS || rows_mode) composite_passes(seg_rows); else composite_passes(SegGeneric{S});
    PT(4);
    __syncthreads();
    RELAUNDER();
    PT(5);
    // ---------------------------------------------------------------- 3. backward
    const float da = valid ? s_alpha[slot] : 0.0f;
    const float dc0 = valid ? s_col[slot] : 0.0f;
    const float dc1 = valid ? s_col[TS + slot] : 0.0f;
    const float dc2 = valid ? s_col[2 * TS + slot] : 0.0f;
    if (g == 0) { g_ba += da; g_boc0 += dc0; g_boc1 += dc1; g_boc2 += dc2; }
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(pe.vh[i]));
    float dps[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dps[i] = 0.f;

    // ---- phase A
    T32 d_hf = zero32();
    if (FEAT) {
      const float wv = valid ? s_w[slot] : 0.0f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gv = valid ? s_gfh[q * 32 + 16 * tt + 4 * g + r] : 0.0f;
          d_hf.t[tt][r] = ((hf_mask >> (4 * tt + r)) & 1u) ? wv * gv : 0.0f;
        }
    }
    T32 d_hc, d_h4;
    float pa_[8], pb_[8], pc_[8], pd_[8];       // head-weight gradient products, summed over the samples below
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * g + r;
        const int s = 4 * tt + r;
        const float hv = hc.t[tt][r];
        if (LAZY_H) {
          hW[0][s] = fmaf(da, h4.t[tt][r], hW[0][s]);
          hW[1][s] = fmaf(dc0, hv, hW[1][s]);
          hW[2][s] = fmaf(dc1, hv, hW[2][s]);
          hW[3][s] = fmaf(dc2, hv, hW[3][s]);
        } else {
          pa_[s] = da * h4.t[tt][r];
          pb_[s] = dc0 * hv;
          pc_[s] = dc1 * hv;
          pd_[s] = dc2 * hv;
        }
        const float dv = fmaf(sm[S_WOC + 2 * H + row], dc2, fmaf(sm[S_WOC + H + row], dc1, sm[S_WOC + row] * dc0));
        d_hc.t[tt][r] = hv > 0.0f ? dv : 0.0f;
        d_h4.t[tt][r] = sm[S_WA + row] * da;
      }
    if (!LAZY_H) {
      gS0 += slot_sums16(pa_, pb_, c);
      gS1 += slot_sums16(pc_, pd_, c);
      asm volatile("" : "+v"(gS0), "+v"(gS1));
    }
    store32_b(stg_lane, 0, h4);
    store32_b(stg_lane, 96, h3);
    store32_b(stg_lane, 128, d_hc);
    const bf16x8 d_hc_b = pack32(d_hc);
    bwd_tile(d_h4.t[0], t_cl, 0, d_hc_b);
    bwd_tile(d_h4.t[1], t_cl, 16, d_hc_b);
    const bf16x8 d_hf_b = pack32(d_hf);
    if (FEAT) {
      bwd_tile(d_h4.t[0], t_fl, 0, d_hf_b);
      bwd_tile(d_h4.t[1], t_fl, 16, d_hf_b);
    }
    d_h4 = relu_mask32(d_h4, h4);
    if (LAZY_B) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bS[1][4 * tt + r] += d_h4.t[tt][r];
    } else {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pa_[4 * tt + r] = d_h4.t[tt][r];
      const float sv = slot_sums8(pa_, c);
      gS2 += (c >= 8) ? sv : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
    store32_b(stg_lane, 160, d_h4);
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      f32x4 d_x = zero4();
      bwd_tile(d_x, t_cl, 32 + 16 * T, d_hc_b);
      if (FEAT) bwd_tile(d_x, t_fl, 32 + 16 * T, d_hf_b);
      float o0, o1, o2, o3;
      obj32n::pe32_x2_pair_fb(pe, 2 * T, g, d_x[0], d_x[1], dps[2 * T], o0, o1);
      obj32n::pe32_x2_pair_fb(pe, 2 * T + 1, g, d_x[2], d_x[3], dps[2 * T + 1], o2, o3);
      store16_b(stg_lane, 32 + 16 * T, f32x4{o0, o1, o2, o3});
    }
    const bf16x8 d_h4_b = pack32(d_h4);
    T32 d_h3 = zero32();
    bwd_tile(d_h3.t[0], t_m2, 0, d_h4_b);
    bwd_tile(d_h3.t[1], t_m2, 16, d_h4_b);
    d_h3 = relu_mask32(d_h3, h3);
    PT(6);
    __syncthreads();
    RELAUNDER();
    PT(7);
    if (w < 7) {
      const int dTr = (w < 5) ? 128 : 160;
      const int aTr = (w < 5) ? 16 * w : 96 + 16 * (w - 5);
      wgrad_pair_b(accA0, accA1, lane_rd + dTr * STG_PITCH, lane_rd + aTr * STG_PITCH);
    }
    PT(8);
    __syncthreads();
    RELAUNDER();
    if (FEAT) {             // feature layer weight gradient: same inputs [h4 | x2], d_hf in place of d_hc
      store32_b(stg_lane, 128, d_hf);
      __syncthreads();
      RELAUNDER();
    RELAUNDER();
      if (w < 5) wgrad_pair_b(accF0, accF1, lane_rd + 128 * STG_PITCH, lane_rd + (16 * w) * STG_PITCH);
      __syncthreads();
      RELAUNDER();
    RELAUNDER();
    }
    PT(9);
    // ---- phase B
    store32_b(stg_lane, 0, h2);
    store32_b(stg_lane, 128, d_h3);
    const bf16x8 d_h3_b = pack32(d_h3);
    T32 d_h2 = zero32();
    bwd_tile(d_h2.t[0], t_cat, 0, d_h3_b);
    bwd_tile(d_h2.t[1], t_cat, 16, d_h3_b);
    d_h2 = relu_mask32(d_h2, h2);
    if (LAZY_B) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bS[0][4 * tt + r] += d_h2.t[tt][r];
    } else {
      float pa2_[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pa2_[4 * tt + r] = d_h2.t[tt][r];
      const float sv = slot_sums8(pa2_, c);
      gS2 += (c < 8) ? sv : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
    const bf16x8 d_h2_b = pack32(d_h2);
    T32 d_h1 = zero32();
    bwd_tile(d_h1.t[0], t_m1, 0, d_h2_b);
    bwd_tile(d_h1.t[1], t_m1, 16, d_h2_b);
    d_h1 = relu_mask32(d_h1, h1);
    const bf16x8 d_h1_b = pack32(d_h1);
#pragma unroll
    for (int T = 0; T < 6; ++T) {
      f32x4 d_x = zero4();
      bwd_tile(d_x, t_cat, 32 + 16 * T, d_h3_b);
      bwd_tile(d_x, t_in, 16 * T, d_h1_b);
      store16_b(stg_lane, 32 + 16 * T, obj32n::pe32_x1_tile_fb(pe, T, g, d_x, dps[T]));
    }
    // d B[j][x] += d proj_j * t_x (embedding.py:48); j = 4 i + g lives in this lane only
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dB[i][0] = fmaf(dps[i], pe.t[0], dB[i][0]);
      dB[i][1] = fmaf(dps[i], pe.t[1], dB[i][1]);
      dB[i][2] = fmaf(dps[i], pe.t[2], dB[i][2]);
    }
    PT(11);
    __syncthreads();
    RELAUNDER();
    PT(12);
    wgrad_pair_b(accB0, accB1, lane_rd + 128 * STG_PITCH, lane_rd + (16 * w) * STG_PITCH);
    PT(13);
    __syncthreads();
    RELAUNDER();
    // ---- phase C
    fetch_point(tile + a.G, slot, nx, ny, nz);
    store32_b(stg_lane, 0, h1);
    store32_b(stg_lane, 128, d_h1);
    store32_b(stg_lane, 160, d_h2);
    PT(14);
    __syncthreads();
    RELAUNDER();
    PT(15);
    {
      const int dTr = (w < 6) ? 128 : 160;
      const int aTr = (w < 6) ? 32 + 16 * w : 16 * (w - 6);
      wgrad_pair_b(accC0, accC1, lane_rd + dTr * STG_PITCH, lane_rd + aTr * STG_PITCH);
    }
    PT(16);
    __syncthreads();
    RELAUNDER();
    PT(17);
  }
  PT_FLUSH();
#undef c
#undef g
#undef stg_lane
#undef lane_rd
#undef f_in
#undef f_m1
#undef f_cat
#undef f_m2
#undef f_cl
#undef t_in
#undef t_m1
#undef t_cat
#undef t_m2
#undef t_cl
#undef f_fl
#undef t_fl

  float* slab = a.slab + ((long)k * a.G + gi) * a.slab_stride;
  const Layout& L = a.L;
  if (w < 5) write_pair_b(slab, accA0, accA1, c, g, w, L.cl_w, H + OBJ_E2, L.cl_b, H, true);
  else if (w < 7) write_pair_b(slab, accA0, accA1, c, g, w - 5, L.m2_w, H, -1);
  write_pair_b(slab, accB0, accB1, c, g, w, L.cat_w, H + OBJ_E1, L.cat_b, H, false);
  if (w < 6) write_pair_b(slab, accC0, accC1, c, g, w, L.in_w, OBJ_E1, L.in_b, 0, false);
  else write_pair_b(slab, accC0, accC1, c, g, w - 6, L.m1_w, H, -1);
  if (FEAT && w < 5) write_pair_b(slab, accF0, accF1, c, g, w, L.fl_w, H + OBJ_E2, L.fl_b, H, true);
  float* red = reinterpret_cast<float*>(stg);   // [NWAVE][NRED]
  {
    float* mine = red + w * NRED;
    const int s = c & 7;
    const int row = 16 * (s >> 2) + 4 * g + (s & 3);
    if (LAZY_H) {
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const int rw = 16 * (s_ >> 2) + 4 * g + (s_ & 3);
        const float v2 = dpp_rowsum16(hW[0][s_]), v3 = dpp_rowsum16(hW[1][s_]);
        const float v4 = dpp_rowsum16(hW[2][s_]), v5 = dpp_rowsum16(hW[3][s_]);
        if (c == 0) { mine[64 + rw] = v2; mine[96 + rw] = v3; mine[128 + rw] = v4; mine[160 + rw] = v5; }
      }
    } else {
      if (c < 8) { mine[64 + row] = gS0; mine[128 + row] = gS1; }
      else { mine[96 + row] = gS0; mine[160 + row] = gS1; }
    }
    if (LAZY_B) {
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const int rw = 16 * (s_ >> 2) + 4 * g + (s_ & 3);
        const float v0 = dpp_rowsum16(bS[0][s_]), v1 = dpp_rowsum16(bS[1][s_]);
        if (c == 0) { mine[rw] = v0; mine[32 + rw] = v1; }
      }
    } else {
      mine[(c < 8 ? 0 : 32) + row] = gS2;
    }
    const float s0 = wave_sum64(g_ba), s1 = wave_sum64(g_boc0), s2 = wave_sum64(g_boc1), s3 = wave_sum64(g_boc2);
    if (lane == 0) { mine[192] = s0; mine[193] = s1; mine[194] = s2; mine[195] = s3; }
    const float e0 = wave_sum64(l_d), e1 = wave_sum64(l_c), e2 = wave_sum64(l_o);
    const float e3 = wave_sum64(l_f);
    if (lane == 0) { mine[196] = e0; mine[197] = e1; mine[198] = e2; mine[199] = e3; }
    float* dbw = red + NWAVE * NRED + w * 72;            // [NWAVE][slot i][g][3] = B's own row-major order
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const float v = dpp_rowsum16(dB[i][x]);
        if (c == 0) dbw[12 * i + 3 * g + x] = v;
      }
  }
  __syncthreads();
  if (tid < 3 * OBJ_NDIR) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[NWAVE * NRED + ww * 72 + tid];
    slab[L.pe_b + tid] = v;
  }
  for (int i = tid; i < NRED; i += NTHR) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[ww * NRED + i];
    if (i < 32) slab[L.m1_b + i] = v;
    else if (i < 64) slab[L.m2_b + i - 32] = v;
    else if (i < 96) slab[L.a_w + i - 64] = v;
    else if (i < 192) slab[L.oc_w + i - 96] = v;
    else if (i == 192) slab[L.a_b] = v;
    else if (i < 196) slab[L.oc_b + i - 193] = v;
    else a.loss_part[((long)k * a.G + gi) * 4 + (i - 196)] = v;
  }
}

}  // namespace

size_t bf16_lds_bytes() { return LDS_BYTES; }

#ifdef PHASE_TIMING
extern "C" int objnerf_debug_phase_bf16(unsigned long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return OBJNERF_ELAUNCH;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase_b), sizeof(unsigned long long) * 8 * 24) == hipSuccess
             ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
#endif

// OBJNERF_BF16_V1=1 (read once): keep the first-generation kernel for the 64-sample no-feature shape too (A/B runs)
static bool bf16_first_generation() {
  static const bool v1 = [] { const char* e = getenv("OBJNERF_BF16_V1"); return e && e[0] == '1'; }();
  return v1;
}

void launch_train_bf16(const TrainDev& d, void* stream, bool feat) {
  objnerf_once_per_device([] {
    const auto at = hipFuncAttributeMaxDynamicSharedMemorySize;
    (void)hipFuncSetAttribute((const void*)train_fused_bf16_kernel<false, 0>, at, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)train_fused_bf16_kernel<false, 64>, at, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)train_fused_bf16_kernel<true, 0>, at, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)train_fused_bf16_kernel<true, 64>, at, LDS_BYTES);
  });
  if (d.S == 64 && !bf16_first_generation()) {
    if (feat) launch_train_bf16_v2f(d, stream); else launch_train_bf16_v2(d, stream);
    return;
  }
  const dim3 grid(d.K * d.G), blk(NTHR);
  hipStream_t st = (hipStream_t)stream;
  if (feat && d.S == 64) hipLaunchKernelGGL((train_fused_bf16_kernel<true, 64>), grid, blk, LDS_BYTES, st, d);
  else if (feat) hipLaunchKernelGGL((train_fused_bf16_kernel<true, 0>), grid, blk, LDS_BYTES, st, d);
  else if (d.S == 64) hipLaunchKernelGGL((train_fused_bf16_kernel<false, 64>), grid, blk, LDS_BYTES, st, d);
  else hipLaunchKernelGGL((train_fused_bf16_kernel<false, 0>), grid, blk, LDS_BYTES, st, d);
}

}  // namespace objtrain
