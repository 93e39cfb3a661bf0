// Fused fp32 training iteration for K stacked hidden-32 object networks, second generation (gfx950).
//
// Same contract and tile structure as the first generation (objnerf_train.hip: its forward-only / eval kernels and
// the host entry points): one launch = forward, compositing, losses, backward (dgrad + wgrad) of train.py:424-472; a 512-thread
// workgroup owns one object's weights in LDS and sweeps its rays in tiles of 128 samples; partial gradients leave as
// one slab per workgroup (no global atomics, bit-reproducible).  What is new is in objnerf_mlp32.h: the
// direction-owner positional encoding (no cross-group sums, d B in registers, one range reduction per direction for
// all octaves) and the in-major conflict-free weight image (one ds_read_b64 per forward k-step, two ds_read_b128 per
// transposed tile).  Per tile and wave: 560 MFMAs (602 before) and ~40 % fewer VALU instructions.
#include <mutex>
#include "objnerf_mlp32.h"
#include "objnerf_train_common.h"
#include "../../include/objnerf_hip.h"

using namespace obj32n;
using namespace objtrain;

#if !defined(OBJ_LAZY_HEADS) && !defined(OBJ_BUTTERFLY_HEADS)
#define OBJ_LAZY_HEADS
#endif

// Ceiling measurements (tools/f32_ablation.sh -> profiles/r06_f32_ablation.txt): -DOBJ32_ABL=<bits> REMOVES parts of the
// tile loop so that the kernel's time can be attributed.  Every such build computes wrong gradients by design and is
// only ever built as a variant library (never by the Makefile, whose OBJ32_ABL is 0: all of this folds away).
//   1  the compositing / loss pass (two of eight waves)      2  the tile loop's barriers
//   4  the staging stores of the weight-gradient operands     8  the weight-gradient MFMA loops
//  16  v_sin / v_cos of the encoding (identity instead)       32 the encoding's arithmetic altogether (fwd + chain rule)
//  64  the forward / input-gradient MFMAs
#ifndef OBJ32_ABL
#define OBJ32_ABL 0
#endif
#define TILE_SYNC() do { if (!((OBJ32_ABL) & 2)) __syncthreads(); } while (0)
// k-step pairs of a weight-gradient loop unrolled together (of 16): 16 in the headline instantiation (measured with the
// one-chain input gradients: 2 -> 9.25 ms, 4 -> 8.91, 8 -> 8.89, 12 / 16 -> 8.82), 8 elsewhere (the feature instantiation
// spills at 16: 12.0 -> 19.7 ms)
#ifndef OBJ32_WG_UNROLL_FAST
#define OBJ32_WG_UNROLL_FAST 16
#endif

namespace {

__device__ __forceinline__ void st_T32(float* stg_lane, const int rowbase, const T32& v) {
  if ((OBJ32_ABL) & 4) { asm volatile("" :: "v"(v.t[0]), "v"(v.t[1])); return; }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) stg_lane[(rowbase + 16 * tt + r) * STG_LD] = v.t[tt][r];
}
__device__ __forceinline__ void st_T16(float* stg_lane, const int rowbase, const f32x4& v) {
  if ((OBJ32_ABL) & 4) { asm volatile("" :: "v"(v)); return; }
#pragma unroll
  for (int r = 0; r < 4; ++r) stg_lane[(rowbase + r) * STG_LD] = v[r];
}

// D[out 0..31][in 16 cols] += sum_s dT[out][s] * aT[in][s] over the 128 staged samples (MFMA k-slot (step st, lane
// group g) = sample 32 g + st: a lane walks consecutive samples, two steps per ds_read_b64, conflict-free with the
// 130-float row pitch).  dT / aT point at &stg[(row0 + c) * LD + 32 g].
template <int UNR = 8>
__device__ __forceinline__ void wg_pair(f32x4& acc0, f32x4& acc1, const float* dT, const float* aT) {
  if ((OBJ32_ABL) & 8) return;
  dT = (const float*)__builtin_assume_aligned(dT, 8);
  aT = (const float*)__builtin_assume_aligned(aT, 8);
#pragma unroll UNR
  for (int st = 0; st < 32; st += 2) {
    const f32x2 b = *reinterpret_cast<const f32x2*>(aT + st);
    const f32x2 a0 = *reinterpret_cast<const f32x2*>(dT + st);
    const f32x2 a1 = *reinterpret_cast<const f32x2*>(dT + 16 * STG_LD + st);
    acc0 = OBJ_MFMA(a0[0], b[0], acc0);
    acc1 = OBJ_MFMA(a1[0], b[0], acc1);
    acc0 = OBJ_MFMA(a0[1], b[1], acc0);
    acc1 = OBJ_MFMA(a1[1], b[1], acc1);
  }
}

// one 16-output half of a tile pair (the feature variant balances its 28 tile pairs over the waves in halves)
template <int UNR = 8>
__device__ __forceinline__ void wg_half(f32x4& acc0, const float* dT, const float* aT) {
  if ((OBJ32_ABL) & 8) return;
  dT = (const float*)__builtin_assume_aligned(dT, 8);
  aT = (const float*)__builtin_assume_aligned(aT, 8);
#pragma unroll UNR
  for (int st = 0; st < 32; st += 2) {
    const f32x2 b = *reinterpret_cast<const f32x2*>(aT + st);
    const f32x2 a0 = *reinterpret_cast<const f32x2*>(dT + st);
    acc0 = OBJ_MFMA(a0[0], b[0], acc0);
    acc0 = OBJ_MFMA(a0[1], b[1], acc0);
  }
}

// a quarter: one 16-output half over half of the staged samples (steps st0 .. st0 + 15 of every lane group's 32)
template <int UNR = 8>
__device__ __forceinline__ void wg_quarter(f32x4& acc0, const float* dT, const float* aT, const int st0) {
  if ((OBJ32_ABL) & 8) return;
  dT = (const float*)__builtin_assume_aligned(dT, 8);
  aT = (const float*)__builtin_assume_aligned(aT, 8);
#pragma unroll UNR
  for (int st = st0; st < st0 + 16; st += 2) {
    const f32x2 b = *reinterpret_cast<const f32x2*>(aT + st);
    const f32x2 a0 = *reinterpret_cast<const f32x2*>(dT + st);
    acc0 = OBJ_MFMA(a0[0], b[0], acc0);
    acc0 = OBJ_MFMA(a0[1], b[1], acc0);
  }
}

// one weight-gradient tile pair -> slab.  col: reference column of this lane's staged input row (>= 0), BIAS_COL
// (-> b_off) or ZERO_COL (padding: nothing to write)
__device__ __forceinline__ void wr_pair(float* slab, const f32x4& a0, const f32x4& a1, const int g, const int col,
                                        const int w_off, const int ncols, const int b_off) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o0 = 4 * g + r, o1 = 16 + 4 * g + r;
    if (col >= 0) {
      slab[w_off + o0 * ncols + col] = a0[r];
      slab[w_off + o1 * ncols + col] = a1[r];
    } else if (col == BIAS_COL && b_off >= 0) {
      slab[b_off + o0] = a0[r];
      slab[b_off + o1] = a1[r];
    }
  }
}

// outputs 16 half + 4 g + r of one half pair
__device__ __forceinline__ void wr_half(float* slab, const f32x4& a0, const int half, const int g, const int col,
                                        const int w_off, const int ncols, const int b_off) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 16 * half + 4 * g + r;
    if (col >= 0) slab[w_off + o * ncols + col] = a0[r];
    else if (col == BIAS_COL && b_off >= 0) slab[b_off + o] = a0[r];
  }
}

#ifdef PHASE_TIMING
__device__ unsigned long long g_phase32[8][24];
#define PT_INIT() unsigned long long pt_acc[18]; for (int i_ = 0; i_ < 18; ++i_) pt_acc[i_] = 0; \
  unsigned long long pt_t0 = __builtin_amdgcn_s_memtime()
#define PT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_acc[i] += t_ - pt_t0; pt_t0 = t_; } while (0)
#define PT_FLUSH() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 18; ++i_) g_phase32[w][i_] = pt_acc[i_]; } while (0)
#else
#define PT_INIT() do {} while (0)
#define PT(i) do {} while (0)
#define PT_FLUSH() do {} while (0)
#endif

// LDS: weight image | sigma / rgb strip | staging area.  With the feature layer the image is 72.8 KB, so the staging area
// has 160 rows instead of 192 and the weight-gradient rounds are laid out differently (rows of the (d_out | input)
// operands per round; a round = one barrier-separated set of tile (half) pairs, one per wave):
//   no feature loss  A  [h4 | x2] 0..79, h3 96.., d_hc 128.., d_h4 160..      colour 5 + mid2 2 pairs
//                    B  [h2 | x1] 0..127, d_h3 128..                           cat 8
//                    C  h1 0.., x1 32..127 (stays), d_h1 128.., d_h2 160..     in 6 + mid1 2
//   feature loss     A  [h4 | x2] 0..79, d_hc 80.., d_hf 112..                 colour 5 + feature 3 pairs
//                    A2 h3 0.., x2 48..79 (stays), d_h4 80.., d_hf (stays)     feature 2 + mid2 2 pairs, as 8 halves
//                    B  as above                                               cat 8
//                    C  x1 (stays), d_h1 128..                                 in 6
//                    C2 h1 0.., d_h2 128..                                     mid1 2 pairs as 8 quarters (halves x sample halves)
template <bool FEAT>
struct Lay {
  static constexpr int IMG = img_floats(FEAT);
  static constexpr int ROWS = FEAT ? 160 : STG_ROWS;
  static constexpr int LDS_FLOATS = IMG + SM_FLOATS + ROWS * STG_LD;
  static constexpr int A_H3 = FEAT ? 0 : 96, A_DHC = FEAT ? 80 : 128, A_DHF = 112, A_DH4 = FEAT ? 80 : 160;
  static constexpr int C_DH2 = FEAT ? 128 : 160;
  static_assert(LDS_FLOATS * 4 <= 163840, "LDS budget");
  static_assert((IMG * 4) % 16 == 0, "staging area alignment");
};
// feature branch: aliases inside the staging area (float offsets), live from the forward pass to the start of phase A.
// hidden-feature buffer [128][33] at 0, Gram matrix + wb + bb, per-wave exchange buffers (objnerf_train_common.h
// OFF_GBUF / OFF_FHB: rows 0..45); compositing weights, d loss / d fh, opacity terms in rows 144..159, which no round
// of the feature layout stages.
constexpr int F_SW = 144 * STG_LD;          // s_w [128]
constexpr int F_GFH = F_SW + TS;            // gfh [16][32]
constexpr int F_GOF = F_GFH + 16 * 32;      // gO_feat [16], O [16]
static_assert(F_GOF + 32 <= 160 * STG_LD && OFF_FHB + 64 * NWAVE <= 80 * STG_LD, "feature aliases");

// SS: samples per ray when known at compile time (64 = the metric shape: no integer divisions by S, only the row-scan
// compositing is compiled in), 0 = any S <= 64 at run time.
template <bool FEAT, bool MASKS, int SS>
__global__ __launch_bounds__(NTHR) void train_fused32_kernel(const TrainDev a) {
  typedef Lay<FEAT> LY;
  constexpr int IMG = LY::IMG;
  // without the feature loss: one-chain input gradients (objnerf_mlp32.h, mma_t16) in EVERY instantiation -- the test-hook
  // and any-S builds must round like the headline one (test_headline_config_full_size_vs_anchor_and_additivity holds the
  // hook launch bit-equal to the production launch); the headline instantiation also unrolls its weight-gradient loops fully
  constexpr bool FAST = !FEAT;            // (measured in the feature instantiation too: 12.06 against 12.0 - 12.2 ms, no gain)
  constexpr int WGU = (!FEAT && !MASKS && SS == 64) ? OBJ32_WG_UNROLL_FAST : 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  float* sv = lds + sv_base(FEAT);
  float* s_alpha = lds + IMG;            // [4][TS]: 10 * raw alpha | colour[3]; overwritten in place by their gradients
  float* s_col = s_alpha + TS;
  float* stg = lds + IMG + SM_FLOATS;

  // this workgroup's segments (objnerf_train_common.h): one (k, gi, NT, step G) in strided mode; in flat mode the
  // pieces of its share of the flat (object, tile) space, each with its own weight staging and partial slab
  const long T_flat = (long)a.K * a.NT;
  long cur = a.flat_nwg ? T_flat * blockIdx.x / a.flat_nwg : 0;
  const long cur_end = a.flat_nwg ? T_flat * (blockIdx.x + 1) / a.flat_nwg : 1;
  bool first_seg = true;
  while (cur < cur_end) {
  int k, t0, t1, tstep, sslot;
  if (a.flat_nwg) {
    k = (int)(cur / a.NT);
    t0 = (int)(cur - (long)k * a.NT);
    const long rem = cur_end - cur;
    t1 = (long)t0 + rem < (long)a.NT ? (int)(t0 + rem) : a.NT;
    tstep = 1;
    sslot = (int)blockIdx.x - flat_wg_of(T_flat, a.flat_nwg, (long)k * a.NT);
    cur += t1 - t0;
  } else {
    k = blockIdx.x / a.G; t0 = blockIdx.x % a.G; t1 = a.NT; tstep = a.G; sslot = t0;
    cur = cur_end;
  }
  const int n_slots = a.flat_nwg ? a.Gs : a.G;
  if (!first_seg) __syncthreads();       // (the previous segment's reduction has read the staging area)
  first_seg = false;

  stage_weights32(lds, a.params + (long)k * a.p_stride, a.L, FEAT, tid, NTHR);
  for (int i = tid; i < LY::ROWS * STG_LD; i += NTHR) stg[i] = 0.0f;
  __syncthreads();

  const float scale = a.scale[k];
  const int S = SS ? SS : a.S, R = a.R, TR = SS ? TS / SS : a.TR;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  int bflag0, bflag1;
  batch_flags(a, bflag0, bflag1);
  const float inv1 = bflag0 ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = bflag1 ? 0.0f : 1.0f / (n2 + 1e-10f);

  // per-lane LDS bases (objnerf_mlp32.h)
  const float* wf = (const float*)__builtin_assume_aligned(lds + 4 * g * WROW + out_pos(c), 8);
  const float* wt0 = (const float*)__builtin_assume_aligned(lds + c * WROW + 8 * g + 4 * (g & 1), 16);
  const float* wt1 = (const float*)__builtin_assume_aligned(lds + c * WROW + 8 * g + 4 * (1 - (g & 1)), 16);
  float* stg_lane = stg + (4 * g) * STG_LD + 16 * w + c;
  const float* lane_rd = stg + c * STG_LD + 32 * g;

  // persistent gradient accumulators
  f32x4 accA0 = zero4(), accA1 = zero4(), accB0 = zero4(), accB1 = zero4(), accC0 = zero4(), accC1 = zero4();
  f32x4 accA2 = zero4(), accC2 = zero4();    // feature layout: the half pairs of rounds A2 and C2
  // row-wise sums over samples: slot s of a register = lane s of each lane group; feature of slot s (< 8) is
  // 16 (s >> 2) + 4 g + (s & 3); lanes 8..15 carry a second quantity
  // Row sums over the samples (head weights, mid1 / mid2 biases).  Two forms: transposing DPP butterflies into "slot"
  // registers (slot s of a register = lane s of each lane group; feature of slot s (< 8) is 16 (s >> 2) + 4 g + (s & 3);
  // lanes 8..15 carry a second quantity) -- or, OBJ_LAZY_HEADS / OBJ_LAZY_BIAS, per-lane partial sums over this
  // lane's samples, reduced over the 16 lanes of the group once at the end (fewer instructions, 32 / 16 more
  // registers).  The head weights are lazy by default (10.08 -> 9.78 ms on the 50 x 4096 x 64 step); both together
  // do not fit the 256 registers of a 512-thread workgroup (OBJ_LAZY_BIAS alone: 9.91 ms).
  // (the feature variant needs the 30 registers: butterfly form there; whichever form is unused is dead code)
#ifdef OBJ_LAZY_HEADS
  constexpr bool LAZY_H = !FEAT;
#else
  constexpr bool LAZY_H = false;
#endif
  float hW[4][8];     // d W_alpha, d W_oc[0..2];  [.][4 tt + r] <-> feature 16 tt + 4 g + r
#pragma unroll
  for (int s_ = 0; s_ < 8; ++s_) hW[0][s_] = hW[1][s_] = hW[2][s_] = hW[3][s_] = 0.f;
  float gS0 = 0.f;   // [0..7] d W_alpha   | [8..15] d W_oc[0]
  float gS1 = 0.f;   // [0..7] d W_oc[1]   | [8..15] d W_oc[2]
#ifdef OBJ_LAZY_BIAS
  float bS[2][8];     // d b_mid1, d b_mid2
#pragma unroll
  for (int s_ = 0; s_ < 8; ++s_) bS[0][s_] = bS[1][s_] = 0.f;
#else
  float gS2 = 0.f;   // [0..7] d b_mid1    | [8..15] d b_mid2
#endif
  float g_hb = 0.f;  // d (alpha bias | colour bias g - 1) of this lane's head output, summed over its samples
  float dB[6][3];    // d B[4 i + g][x], summed over this lane's samples
#pragma unroll
  for (int i = 0; i < 6; ++i) dB[i][0] = dB[i][1] = dB[i][2] = 0.f;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f, l_f = 0.f;

  // sample position of (tile, slot); issued one tile ahead (phase C) so that the HBM latency is off the tile's
  // critical path
  auto fetch_point = [&](const int tile_, const int slot_, float& x, float& y, float& z_) {
    const int q_ = slot_ / S, si_ = slot_ - q_ * S;
    const int ray_ = tile_ * TR + q_;
    x = 0.f; y = 0.f; z_ = 0.f;
    if (tile_ < t1 && q_ < TR && ray_ < a.R) {
      const long rr = (long)k * a.R + ray_;
      if (a.pts) {
        const float* p = a.pts + (rr * S + si_) * 3;
        x = p[0]; y = p[1]; z_ = p[2];
      } else {
        const float zz = a.z[rr * S + si_];
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        x = (o[0] + d[0] * zz) - a.obj_center;   // vmap.py:548-551 (two roundings: -ffp-contract=off)
        y = (o[1] + d[1] * zz) - a.obj_center;
        z_ = (o[2] + d[2] * zz) - a.obj_center;
      }
    }
  };
  const bool rows_mode = SS ? true : seg_is_rows(S);
  const SegRows seg_rows = SegRows::make(rows_mode ? S : 64, lane);
  const int slot = 16 * w + c;
  float nx, ny, nz;
  fetch_point(t0, slot, nx, ny, nz);
  PT_INIT();
  for (int tile = t0; tile < t1; tile += tstep) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    const int ray0 = tile * TR;
    // ---------------------------------------------------------------- 1. forward
    const int q = slot / S;
    const int ray = ray0 + q;
    const bool valid = (q < TR) && (ray < R);
    Pe32 pe;
    pe32_project(sv, g, nx, ny, nz, scale, pe);     // (nx, ny, nz) fetched during the previous tile's phase C
    PT(0);
    Acts act;
    {
      Emb32 e;                        // forward-only: the backward re-creates the embedding tile by tile
      embed32(e, pe, g);
      PT(1);
      if (MASKS && a.emb_debug && valid) {     // test hook: this lane's share of its sample's embedding row
        float* er = a.emb_debug + (((long)k * R + ray) * S + (slot - q * S)) * (OBJ_E1 + OBJ_E2);
#pragma unroll
        for (int T_ = 0; T_ < 6; ++T_)
#pragma unroll
          for (int r_ = 0; r_ < 4; ++r_) {
            const int col = x1_col(4 * T_ + r_, g);
            if (col >= 0) er[col] = e.x1[T_][r_];
          }
#pragma unroll
        for (int T_ = 0; T_ < 3; ++T_)
#pragma unroll
          for (int r_ = 0; r_ < 4; ++r_) {
            const int col = x2_col(4 * T_ + r_, g);
            if (col >= 0) er[OBJ_E1 + col] = e.x2[T_][r_];
          }
      }
      s_alpha[g * TS + slot] = mlp32_forward<FEAT>(wf, sv, g, e, act);
    }
    if (MASKS && a.relu_masks) {      // test hook: ReLU branch bits of this lane's sample
      uint8_t* dst = a.relu_masks + (((long)k * R + (valid ? ray : 0)) * S + (slot - q * S)) * 24;
      write_relu_mask(dst, 0, g, act.h1, valid);
      write_relu_mask(dst, 1, g, act.h2, valid);
      write_relu_mask(dst, 2, g, act.h3, valid);
      write_relu_mask(dst, 3, g, act.h4, valid);
      write_relu_mask(dst, 4, g, act.hc, valid);
      if (FEAT) write_relu_mask(dst, 5, g, act.hf, valid);
    }
    unsigned hf_pos = 0;              // ReLU branch bits of this lane's 8 hidden-feature entries (all the backward needs
                                      // of them: one register instead of eight across the compositing)
    if (FEAT) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hf_pos |= (act.hf.t[tt][r] > 0.0f ? 1u : 0u) << (4 * tt + r);
      // hidden feature of the tile -> [128][33] buffer; this object's Gram matrix (+ wb, bb) beside it (section 4.3 of
      // the 512-d head is hoisted past the compositing)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[slot * HF_LD + 16 * tt + 4 * g + r] = act.hf.t[tt][r];
      for (int i = tid; i < 32 * 32 + 33; i += NTHR) {
        const float v = a.gram[(long)k * GRAM + i];
        if (i < 1024) stg[OFF_GBUF + (i >> 5) * 33 + (i & 31)] = v;
        else stg[OFF_GBUF + 32 * 33 + (i - 1024)] = v;
      }
    }
    PT(2);
    // ray inputs of this wave's compositing pass, requested BEFORE the barrier so their latency hides behind it
    auto ray_inputs = [&](const int ps_, float& zz_, float& gtd_, float& gr_, float& gg_, float& gb_, int& lab_) {
      const int rpp_ = 64 / S;
      const int ql_ = lane / S, pos_ = lane - ql_ * S;
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
    // feature term: (u[hh], beta, |g|, label) of ray pair (qb, half) -- same idea, first pair of the pass
    auto feat_inputs = [&](const int ps_, const int qb_, float& uh_, float& beta_, float& ngv_, int& lab_) {
      const int rpp_ = 64 / S;
      const int ql2_ = qb_ + (lane >> 5);
      const int qq2_ = ps_ * rpp_ + ql2_;
      const int ray2_ = ray0 + qq2_;
      uh_ = 0.f; beta_ = 0.f; ngv_ = 1.f; lab_ = 2;
      if ((ql2_ < rpp_) && (qq2_ < TR) && (ray2_ < R)) {
        const long rr2_ = (long)k * R + ray2_;
        uh_ = a.rayin[rr2_ * RAYIN + (lane & 31)];
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
        pf_uh = a.rayin[rr2_ * RAYIN + (lane & 31)];
        pf_beta = a.rayin[rr2_ * RAYIN + 32];
        pf_ngv = a.rayin[rr2_ * RAYIN + 33];
        pf_lab2 = (int)a.labels[rr2_];
      }
    } else if (FEAT && w * (64 / S) < TR) feat_inputs(w, 0, pf_uh, pf_beta, pf_ngv, pf_lab2);
    TILE_SYNC();
    PT(3);
    // ---------------------------------------------------------------- 2. composite + loss (loss.py:27-101)
    auto composite_passes = [&](const auto& sg) {
      const int rpp = 64 / S;                       // rays per wave pass
      const int npass = (TR + rpp - 1) / rpp;
      for (int ps = w; ps < npass; ps += NWAVE) {
        const int ql = lane / S, pos = lane - ql * S;
        const int qq = ps * rpp + ql;
        const int rayq = ray0 + qq;
        const bool on = (ql < rpp) && (qq < TR) && (rayq < R);
        const int sl = qq * S + pos;
        float al = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, zz = pf_zz;
        float gtd = pf_gtd, gr = pf_gr, gg = pf_gg, gb = pf_gb;
        int lab = pf_lab;
        if (ps != w) ray_inputs(ps, zz, gtd, gr, gg, gb, lab);      // (only when a tile has more than 8 passes)
        if (on) { al = s_alpha[sl]; c0 = s_col[sl]; c1 = s_col[TS + sl]; c2 = s_col[2 * TS + sl]; }
        const float occ = on ? sigmoid_acc(al) : 0.0f;               // render_rays.py:13
        const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;          // render_rays.py:38
        const float Pinc = sg.scan_mul(fr, pos);
        float T = __shfl_up(Pinc, 1, 64);
        if (pos == 0) T = 1.0f;
        const float wgt = occ * T;                                   // render_rays.py:43
        const float D = sg.total_add(wgt * zz, pos);       // loss.py:31
        const float O = sg.total_add(wgt, pos);            // loss.py:35
        const float C0 = sg.total_add(wgt * c0, pos);      // loss.py:34
        const float C1 = sg.total_add(wgt * c1, pos);
        const float C2 = sg.total_add(wgt * c2, pos);
        const float dz = zz - D;
        const float V = sg.total_add(wgt * (dz * dz), pos);  // loss.py:32-33
        const float m1 = (lab == 1) ? 1.0f : 0.0f;                   // mask_sem & mask_obj
        const float m2 = (lab != 2) ? 1.0f : 0.0f;                   // mask_sem
        const float tgt = (lab != 0) ? 1.0f : 0.0f;                  // mask_obj.float()
        const float info = 1.0f / (sqrtf(V) + 1e-4f);                // render_rays.py:96-100
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
          // ---- feature-distillation term (loss.py:82-99) with the linear 512-d head hoisted past the
          // compositing: F = W_of fh + b_of O,  fh = sum_s w_s hf_s.  cos(F, g) only needs
          //   F.g = fh.u + O beta,   |F|^2 = fh^T G fh + 2 O wb.fh + O^2 bb     (u, beta, G, wb, bb precomputed)
          float* s_w = stg + F_SW;
          float* s_gfh = stg + F_GFH;
          float* s_gof = stg + F_GOF;
          float* s_fhb = stg + OFF_FHB + 64 * w;
          const float* Gb = stg + OFF_GBUF;
          if (on) s_w[sl] = wgt;
          if (on && pos == 0) s_gof[16 + qq] = O;
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
          const int half = lane >> 5, hh = lane & 31;
          for (int qb = 0; qb < rpp; qb += 2) {
            const int ql2 = qb + half;
            const int qq2 = ps * rpp + ql2;
            const int ray2 = ray0 + qq2;
            const bool on2 = (ql2 < rpp) && (qq2 < TR) && (ray2 < R);
            const long rr2 = (long)k * R + (on2 ? ray2 : 0);
            float fh = 0.f;
            if (rpp == 1) {
              // one ray per pass (S = 33..64): both 32-lane halves work on it, half the samples each
              const bool on1 = (qb < rpp) && (ps * rpp + qb < TR) && (ray0 + ps * rpp + qb < R);
              const int q1 = ps * rpp + qb;
              if (on1) {
                float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
                const float* wp = s_w + q1 * S;
                const float* hp = stg + (q1 * S) * HF_LD + hh;
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
              for (int s2 = 0; s2 < S; ++s2) fh = fmaf(s_w[qq2 * S + s2], stg[(qq2 * S + s2) * HF_LD + hh], fh);
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
            const float gam = -a.feat_scaling * mm1 * inv1;         // d total / d cos
            const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
            if (on2) {
              if (hh == 0) {
                l_f += mm1 * (1.0f - cosv) * inv1;
                s_gof[qq2] = ar * beta + cr * (fwb + O2 * bb);       // d total / d opacity (feature part)
                a.rayfeat[rr2 * RAYFEAT + 32] = O2;      // layout (fh[32], O, a, c): [fh | O] is a GEMM operand
                a.rayfeat[rr2 * RAYFEAT + 33] = ar;
                a.rayfeat[rr2 * RAYFEAT + 34] = cr;
              }
              s_gfh[qq2 * 32 + hh] = ar * uh + cr * (Gfh + O2 * wbh);   // d total / d fh
              a.rayfeat[rr2 * RAYFEAT + hh] = fh;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
          }
          if (on) {
            float dwf = s_gof[qq];
#pragma unroll 8
            for (int h2 = 0; h2 < 32; ++h2) dwf = fmaf(s_gfh[qq * 32 + h2], stg[sl * HF_LD + h2], dwf);
            dw += dwf;
          }
        }
        const float qv = dw * wgt;
        const float suf = sg.rscan_add(qv, pos) - qv;            // sum_{j>i} dL/dw_j * w_j
        const float docc = dw * T - suf / fr;
        if (on) {                                                    // in place: this lane owns slot sl
          s_alpha[sl] = 10.0f * (docc * occ * (1.0f - occ));         // d / d raw alpha (model.py:88)
          s_col[sl] = gC0 * wgt * c0 * (1.0f - c0);                  // d / d raw colour (pre-sigmoid)
          s_col[TS + sl] = gC1 * wgt * c1 * (1.0f - c1);
          s_col[2 * TS + sl] = gC2 * wgt * c2 * (1.0f - c2);
        }
      }
    };
    if (FEAT && S == 64) {
      // 64 samples per ray (the north-star shape): the feature term's reductions over samples and hidden features
      // are spread over all 8 waves instead of running on the two compositing waves (3 extra barriers, a much
      // shorter critical path).  Wave w holds samples 16w..16w+15 of ray w >> 2.  Same arithmetic as the
      // general path above.
      const SegRows& sg = seg_rows;
      float* s_w = stg + F_SW;
      float* s_gfh = stg + F_GFH;
      float* s_gof = stg + F_GOF;
      float* s_part = s_gfh + 192;          // [NWAVE][32] partial composited features (s_gfh holds 2 rays here)
      float* s_dwf = s_gfh + 64;            // [128]
      const int pos = lane;
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
      TILE_SYNC();
      {
        const float wv = valid ? s_w[slot] : 0.0f;
        float v8[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v8[4 * tt + r] = wv * stg[slot * HF_LD + 16 * tt + 4 * g + r];
        const float psum = slot_sums8(v8, c);
        if (c < 8) s_part[w * 32 + 16 * ((c & 7) >> 2) + 4 * g + (c & 3)] = psum;
      }
      TILE_SYNC();
      {
        float* s_fhb = stg + OFF_FHB + 64 * w;
        const float* Gb = stg + OFF_GBUF;
        const int half = lane >> 5, hh = lane & 31;
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
        const float nF = fmaxf(sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
        const float cosv = dotFg / (nF * ngc);
        const float mm1 = (pf_lab2 == 1) ? 1.0f : 0.0f;
        const float gam = -a.feat_scaling * mm1 * inv1;
        const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
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
            dp = fmaf(s_fhb[32 + 16 * tt + 4 * g + r], stg[slot * HF_LD + 16 * tt + 4 * g + r], dp);
        const float dwf = xgroup_sum(dp) + gof;
        if (g == 0 && valid) s_dwf[slot] = dwf;
      }
      TILE_SYNC();
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
    } else if ((OBJ32_ABL) & 1) { /* ablation: no compositing */
    } else if (SS || rows_mode) composite_passes(seg_rows); else composite_passes(SegGeneric{S});
    PT(4);
    TILE_SYNC();
    PT(5);
    // ---------------------------------------------------------------- 3. backward
    const float da = valid ? s_alpha[slot] : 0.0f;
    const float dc0 = valid ? s_col[slot] : 0.0f;
    const float dc1 = valid ? s_col[TS + slot] : 0.0f;
    const float dc2 = valid ? s_col[2 * TS + slot] : 0.0f;
    g_hb += (g == 0) ? da : ((g == 1) ? dc0 : ((g == 2) ? dc1 : dc2));
    float dps[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dps[i] = 0.f;
    // The sincos of the embedding is RE-computed below: hide the angles from CSE, otherwise the compiler keeps every
    // forward value live across the whole backward pass.
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(pe.vh[i]), "+v"(pe.vl[i]));

    // ---- phase A: heads, colour layer, (feature layer,) mid2
    T32 d_hf = zero32();
    if (FEAT) {           // d loss / d hf_s = w_s * (d loss / d fh of the sample's ray), through the ReLU
      const float wv = valid ? stg[F_SW + slot] : 0.0f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gv = valid ? stg[F_GFH + q * 32 + 16 * tt + 4 * g + r] : 0.0f;
          d_hf.t[tt][r] = ((hf_pos >> (4 * tt + r)) & 1u) ? wv * gv : 0.0f;
        }
    }
    T32 d_hc, d_h4;
    float pa_[8];
    float pb_[8], pc_[8], pd_[8];       // head-weight gradient products, summed over the samples below (butterfly form)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * g + r;
        const int s = 4 * tt + r;
        const float hv = act.hc.t[tt][r];
        if (LAZY_H) {
          hW[0][s] = fmaf(da, act.h4.t[tt][r], hW[0][s]);
          hW[1][s] = fmaf(dc0, hv, hW[1][s]);
          hW[2][s] = fmaf(dc1, hv, hW[2][s]);
          hW[3][s] = fmaf(dc2, hv, hW[3][s]);
        } else {
          pa_[s] = da * act.h4.t[tt][r];
          pb_[s] = dc0 * hv;
          pc_[s] = dc1 * hv;
          pd_[s] = dc2 * hv;
        }
        const float dv = fmaf(sv[SV_WOC + 2 * H + row], dc2, fmaf(sv[SV_WOC + H + row], dc1, sv[SV_WOC + row] * dc0));
        d_hc.t[tt][r] = hv > 0.0f ? dv : 0.0f;
        d_h4.t[tt][r] = sv[SV_WA + row] * da;
      }
    // round A staging (row maps: Lay)
    if (!LAZY_H) {
      gS0 += slot_sums16(pa_, pb_, c);
      gS1 += slot_sums16(pc_, pd_, c);
      asm volatile("" : "+v"(gS0), "+v"(gS1));
    }
    st_T32(stg_lane, 0, act.h4);
    if (!FEAT) st_T32(stg_lane, LY::A_H3, act.h3);
    st_T32(stg_lane, LY::A_DHC, d_hc);
    if (FEAT) st_T32(stg_lane, LY::A_DHF, d_hf);
    mma_t32<FAST>(d_h4, wt0, wt1, R_CL, d_hc);
    if (FEAT) mma_t32<FAST>(d_h4, wt0, wt1, R_FL, d_hf);
    d_h4 = relu_mask32(d_h4, act.h4);
#ifdef OBJ_LAZY_BIAS
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bS[1][4 * tt + r] += d_h4.t[tt][r];
#else
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pa_[4 * tt + r] = d_h4.t[tt][r];
    {
      const float sv8 = slot_sums8(pa_, c);
      gS2 += (c >= 8) ? sv8 : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
#endif
    if (!FEAT) st_T32(stg_lane, LY::A_DH4, d_h4);
    // PE backward, x2 part (octaves 4, 5): a 16-row tile = two direction slots
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      f32x4 d_x = zero4();
      mma_t16<FAST>(d_x, wt0, wt1, R_CL + 32 + 16 * T, d_hc);
      if (FEAT) mma_t16<FAST>(d_x, wt0, wt1, R_FL + 32 + 16 * T, d_hf);
      float o0, o1, o2, o3;
      pe32_x2_pair_fb(pe, 2 * T, g, d_x[0], d_x[1], dps[2 * T], o0, o1);
      pe32_x2_pair_fb(pe, 2 * T + 1, g, d_x[2], d_x[3], dps[2 * T + 1], o2, o3);
      st_T16(stg_lane, 32 + 16 * T, f32x4{o0, o1, o2, o3});
    }
    T32 d_h3 = zero32();
    mma_t32<FAST>(d_h3, wt0, wt1, R_M2, d_h4);
    d_h3 = relu_mask32(d_h3, act.h3);
    PT(6);
    TILE_SYNC();
    PT(7);
    if (FEAT) {
      // round A: colour tiles 0..4 (waves 0..4), feature tiles 0..2 (waves 5..7)
      const int dTr = (w < 5) ? LY::A_DHC : LY::A_DHF;
      const int aTr = (w < 5) ? 16 * w : 16 * (w - 5);
      wg_pair<WGU>(accA0, accA1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
      TILE_SYNC();
      // round A2: h3 over the h4 rows, d_h4 over the d_hc rows; feature tiles 3, 4 (x2 rows 48..79, d_hf still in
      // place) and the two mid2 tiles, one 16-output half per wave
      st_T32(stg_lane, LY::A_H3, act.h3);
      st_T32(stg_lane, LY::A_DH4, d_h4);
      TILE_SYNC();
      {
        const int half = w & 1, t2 = (w & 3) >> 1;
        const int dTr = ((w < 4) ? LY::A_DHF : LY::A_DH4) + 16 * half;
        const int aTr = (w < 4) ? 48 + 16 * t2 : LY::A_H3 + 16 * t2;
        wg_half<WGU>(accA2, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
      }
    } else if (w < 7) {
      const int dTr = (w < 5) ? 128 : 160;
      const int aTr = (w < 5) ? 16 * w : 96 + 16 * (w - 5);
      wg_pair<WGU>(accA0, accA1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(8);
    TILE_SYNC();
    PT(9);
    // ---- phase B: cat layer.  [h2 | x1] rows 0..127, d_h3pre rows 128..
    st_T32(stg_lane, 0, act.h2);
    st_T32(stg_lane, 128, d_h3);
    T32 d_h2 = zero32();
    mma_t32<FAST>(d_h2, wt0, wt1, R_CAT, d_h3);
    d_h2 = relu_mask32(d_h2, act.h2);
#ifdef OBJ_LAZY_BIAS
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bS[0][4 * tt + r] += d_h2.t[tt][r];
#else
    float pa2_[8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pa2_[4 * tt + r] = d_h2.t[tt][r];
    {
      const float sv8 = slot_sums8(pa2_, c);
      gS2 += (c < 8) ? sv8 : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
#endif
    T32 d_h1 = zero32();
    mma_t32<FAST>(d_h1, wt0, wt1, R_M1, d_h2);
    d_h1 = relu_mask32(d_h1, act.h1);
    // PE backward, x1 part (octaves 0..3): d x1 tile = cat^T d_h3 + in^T d_h1; a tile = one direction slot
#pragma unroll
    for (int T = 0; T < 6; ++T) {
      f32x4 d_x = zero4();
      mma_t16<FAST>(d_x, wt0, wt1, R_CAT + 32 + 16 * T, d_h3);
      mma_t16<FAST>(d_x, wt0, wt1, R_IN + 16 * T, d_h1);
      st_T16(stg_lane, 32 + 16 * T, pe32_x1_tile_fb(pe, T, g, d_x, dps[T]));
    }
    // d B[j][x] += d proj_j * t_x (embedding.py:48); j = 4 i + g lives in this lane only
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dB[i][0] = fmaf(dps[i], pe.t[0], dB[i][0]);
      dB[i][1] = fmaf(dps[i], pe.t[1], dB[i][1]);
      dB[i][2] = fmaf(dps[i], pe.t[2], dB[i][2]);
    }
    PT(10);
    PT(11);
    TILE_SYNC();
    PT(12);
    wg_pair<WGU>(accB0, accB1, lane_rd + 128 * STG_LD, lane_rd + (16 * w) * STG_LD);
    PT(13);
    TILE_SYNC();
    // ---- phase C: in layer (x1 stays at rows 32..127) + mid1.  h1 rows 0.., d_h1pre 128.., d_h2pre 160..
    fetch_point(tile + tstep, slot, nx, ny, nz);
    st_T32(stg_lane, 0, act.h1);
    st_T32(stg_lane, 128, d_h1);
    if (!FEAT) st_T32(stg_lane, LY::C_DH2, d_h2);
    PT(14);
    TILE_SYNC();
    PT(15);
    if (FEAT) {
      // round C: the six in-layer tiles; round C2: d_h2 over the d_h1 rows, the two mid1 tiles as eight quarters
      if (w < 6) wg_pair<WGU>(accC0, accC1, lane_rd + 128 * STG_LD, lane_rd + (32 + 16 * w) * STG_LD);
      TILE_SYNC();
      st_T32(stg_lane, LY::C_DH2, d_h2);
      TILE_SYNC();
      // all eight waves: wave w takes half (w >> 1) & 1 of tile w >> 2 over the sample half w & 1; the two partial
      // sums of a half meet once, after the sweep
      wg_quarter<WGU>(accC2, lane_rd + (LY::C_DH2 + 16 * ((w >> 1) & 1)) * STG_LD, lane_rd + (16 * (w >> 2)) * STG_LD,
                 16 * (w & 1));
      // the next tile's forward pass writes its hidden-feature buffer into rows this round is reading
      TILE_SYNC();
    } else {
      const int dTr = (w < 6) ? 128 : 160;
      const int aTr = (w < 6) ? 32 + 16 * w : 16 * (w - 6);
      wg_pair<WGU>(accC0, accC1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(16);
    // (no feature loss: the staging area is next written in phase A of the following tile, two barriers from here)
    PT(17);
  }
  PT_FLUSH();

  // ------------------------------------------------------------------ write this workgroup's slab
  float* slab = a.slab + ((long)k * n_slots + sslot) * a.slab_stride;
  const Layout& L = a.L;
  {
    // reference column of the staged input row this lane's accumulator column stands for
    int t_, g_;
    // reference column of row rho of a [h4 | x2] operand (colour and feature layers)
    auto cl_col = [&](const int rho) {
      int col = rho;
      if (rho >= H) { kappa_tg(rho - H, t_, g_); col = x2_col(t_, g_); if (col >= 0) col += H; }
      return col;
    };
    if (FEAT) {
      if (w < 5) wr_pair(slab, accA0, accA1, g, cl_col(16 * w + c), L.cl_w, H + OBJ_E2, L.cl_b);
      else wr_pair(slab, accA0, accA1, g, cl_col(16 * (w - 5) + c), L.fl_w, H + OBJ_E2, L.fl_b);
      const int half = w & 1, t2 = (w & 3) >> 1;
      if (w < 4) wr_half(slab, accA2, half, g, cl_col(48 + 16 * t2 + c), L.fl_w, H + OBJ_E2, L.fl_b);
      else wr_half(slab, accA2, half, g, 16 * t2 + c, L.m2_w, H, -1);
    } else if (w < 5) {                           // colour layer: [h4 | x2]
      wr_pair(slab, accA0, accA1, g, cl_col(16 * w + c), L.cl_w, H + OBJ_E2, L.cl_b);
    } else if (w < 7) {
      wr_pair(slab, accA0, accA1, g, 16 * (w - 5) + c, L.m2_w, H, -1);
    }
    {                                             // cat layer: [h2 | x1]
      const int rho = 16 * w + c;
      int col = rho;
      if (rho >= H) { kappa_tg(rho - H, t_, g_); col = x1_col(t_, g_); if (col >= 0) col += H; }
      wr_pair(slab, accB0, accB1, g, col, L.cat_w, H + OBJ_E1, L.cat_b);
    }
    if (w < 6) {                                  // in layer: x1
      kappa_tg(16 * w + c, t_, g_);
      wr_pair(slab, accC0, accC1, g, x1_col(t_, g_), L.in_w, OBJ_E1, L.in_b);
    } else if (!FEAT) {
      wr_pair(slab, accC0, accC1, g, 16 * (w - 6) + c, L.m1_w, H, -1);
    }
  }
  if (FEAT) {        // round C2's sample halves: odd waves hand their partial sums to their even partners
    __syncthreads();                                  // (the last round's staging reads are done)
    float* xch = stg + (w >> 1) * 256 + 4 * lane;
    if (w & 1) *reinterpret_cast<f32x4*>(xch) = accC2;
    __syncthreads();
    if (!(w & 1)) {
      accC2 += *reinterpret_cast<const f32x4*>(xch);
      wr_half(slab, accC2, (w >> 1) & 1, g, 16 * (w >> 2) + c, L.m1_w, H, -1);
    }
  }
  // slot registers -> LDS (per wave), then sum the 8 waves
  __syncthreads();    // the last tile's weight-gradient reads of the staging area are done
  float* red = stg;   // [NWAVE][NRED32]
  constexpr int NRED32 = 6 * 32 + 4 + 4 + 72;     // row sums | head biases | loss terms | d B [slot][g][3]
  {
    float* mine = red + w * NRED32;
    {
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
        if (c < 8) { mine[64 + row] = gS0; mine[128 + row] = gS1; }          // wa, woc1
        else { mine[96 + row] = gS0; mine[160 + row] = gS1; }                // woc0, woc2
      }
#ifdef OBJ_LAZY_BIAS
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const int rw = 16 * (s_ >> 2) + 4 * g + (s_ & 3);
        const float v0 = dpp_rowsum16(bS[0][s_]), v1 = dpp_rowsum16(bS[1][s_]);
        if (c == 0) { mine[rw] = v0; mine[32 + rw] = v1; }
      }
#else
      mine[(c < 8 ? 0 : 32) + row] = gS2;                                  // bm1 | bm2
#endif
    }
    const float hb = dpp_rowsum16(g_hb);                 // this lane group's head-bias gradient
    if (c == 0) mine[192 + g] = hb;
    const float e0 = wave_sum64(l_d), e1 = wave_sum64(l_c), e2 = wave_sum64(l_o);
    const float e3 = FEAT ? wave_sum64(l_f) : 0.0f;
    if (lane == 0) { mine[196] = e0; mine[197] = e1; mine[198] = e2; mine[199] = e3; }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const float v = dpp_rowsum16(dB[i][x]);
        if (c == 0) mine[200 + 12 * i + 3 * g + x] = v;
      }
  }
  __syncthreads();
  for (int i = tid; i < NRED32; i += NTHR) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[ww * NRED32 + i];
    if (i < 32) slab[L.m1_b + i] = v;
    else if (i < 64) slab[L.m2_b + i - 32] = v;
    else if (i < 96) slab[L.a_w + i - 64] = v;
    else if (i < 192) slab[L.oc_w + i - 96] = v;
    else if (i == 192) slab[L.a_b] = v;
    else if (i < 196) slab[L.oc_b + i - 193] = v;
    else if (i < 200) a.loss_part[((long)k * n_slots + sslot) * 4 + (i - 196)] = v;
    else if (i - 200 < 3 * OBJ_NDIR) slab[L.pe_b + (i - 200)] = v;       // [4 i + g][x] = B's own row-major order
  }
  }      // segments
}

}  // namespace

namespace objtrain {

size_t fused32_lds_bytes() { return (size_t)Lay<false>::LDS_FLOATS * 4; }

void launch_train32(const TrainDev& d, void* stream, bool feat) {
  constexpr int n0 = Lay<false>::LDS_FLOATS * 4, n1 = Lay<true>::LDS_FLOATS * 4;
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<false, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, n0);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<false, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n0);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<false, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n0);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<true, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n1);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<true, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, n1);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n1);
  });
  const dim3 grid(d.flat_nwg ? d.flat_nwg : d.K * d.G), blk(NTHR);
  hipStream_t st = (hipStream_t)stream;
  if (feat) {
    if (d.relu_masks || d.emb_debug) hipLaunchKernelGGL((train_fused32_kernel<true, true, 0>), grid, blk, n1, st, d);
    else if (d.S == 64) hipLaunchKernelGGL((train_fused32_kernel<true, false, 64>), grid, blk, n1, st, d);
    else hipLaunchKernelGGL((train_fused32_kernel<true, false, 0>), grid, blk, n1, st, d);
  } else if (d.relu_masks || d.emb_debug) hipLaunchKernelGGL((train_fused32_kernel<false, true, 0>), grid, blk, n0, st, d);
  else if (d.S == 64) hipLaunchKernelGGL((train_fused32_kernel<false, false, 64>), grid, blk, n0, st, d);
  else hipLaunchKernelGGL((train_fused32_kernel<false, false, 0>), grid, blk, n0, st, d);
}

}  // namespace objtrain

#ifdef PHASE_TIMING
extern "C" int objnerf_debug_phase32(unsigned long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase32), sizeof(unsigned long long) * 8 * 24) == hipSuccess ? 0 : -1;
}
#endif
