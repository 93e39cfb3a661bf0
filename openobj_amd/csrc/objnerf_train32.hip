// Fused fp32 training iteration for K stacked hidden-32 object networks, second generation (gfx950).
//
// Same contract and tile structure as the first generation (objnerf_train.hip, which still serves the 512-d feature
// loss): one launch = forward, compositing, losses, backward (dgrad + wgrad) of train.py:424-472; a 512-thread
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

namespace {

__device__ __forceinline__ void st_T32(float* stg_lane, const int rowbase, const T32& v) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) stg_lane[(rowbase + 16 * tt + r) * STG_LD] = v.t[tt][r];
}
__device__ __forceinline__ void st_T16(float* stg_lane, const int rowbase, const f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) stg_lane[(rowbase + r) * STG_LD] = v[r];
}

// D[out 0..31][in 16 cols] += sum_s dT[out][s] * aT[in][s] over the 128 staged samples (MFMA k-slot (step st, lane
// group g) = sample 32 g + st: a lane walks consecutive samples, two steps per ds_read_b64, conflict-free with the
// 130-float row pitch).  dT / aT point at &stg[(row0 + c) * LD + 32 g].
__device__ __forceinline__ void wg_pair(f32x4& acc0, f32x4& acc1, const float* dT, const float* aT) {
  dT = (const float*)__builtin_assume_aligned(dT, 8);
  aT = (const float*)__builtin_assume_aligned(aT, 8);
#pragma unroll 8
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

constexpr int IMG = img_floats(false);
constexpr int LDS_FLOATS32 = IMG + SM_FLOATS + STG_ROWS * STG_LD;
static_assert(LDS_FLOATS32 * 4 <= 163840, "LDS budget");
static_assert((IMG * 4) % 16 == 0, "staging area alignment");

// SS: samples per ray when known at compile time (64 = the metric shape: no integer divisions by S, only the row-scan
// compositing is compiled in), 0 = any S <= 64 at run time.
template <bool MASKS, int SS>
__global__ __launch_bounds__(NTHR) void train_fused32_kernel(const TrainDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int k = blockIdx.x / a.G, gi = blockIdx.x % a.G;
  float* sv = lds + sv_base(false);
  float* s_alpha = lds + IMG;            // [4][TS]: 10 * raw alpha | colour[3]; overwritten in place by their gradients
  float* s_col = s_alpha + TS;
  float* stg = lds + IMG + SM_FLOATS;

  stage_weights32(lds, a.params + (long)k * a.p_stride, a.L, false, tid, NTHR);
  for (int i = tid; i < STG_ROWS * STG_LD; i += NTHR) stg[i] = 0.0f;
  __syncthreads();

  const float scale = a.scale[k];
  const int S = SS ? SS : a.S, R = a.R, TR = SS ? TS / SS : a.TR;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  const float inv1 = a.flags[0] ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = a.flags[1] ? 0.0f : 1.0f / (n2 + 1e-10f);

  // per-lane LDS bases (objnerf_mlp32.h)
  const float* wf = (const float*)__builtin_assume_aligned(lds + 4 * g * WROW + out_pos(c), 8);
  const float* wt0 = (const float*)__builtin_assume_aligned(lds + c * WROW + 8 * g + 4 * (g & 1), 16);
  const float* wt1 = (const float*)__builtin_assume_aligned(lds + c * WROW + 8 * g + 4 * (1 - (g & 1)), 16);
  float* stg_lane = stg + (4 * g) * STG_LD + 16 * w + c;
  const float* lane_rd = stg + c * STG_LD + 32 * g;

  // persistent gradient accumulators
  f32x4 accA0 = zero4(), accA1 = zero4(), accB0 = zero4(), accB1 = zero4(), accC0 = zero4(), accC1 = zero4();
  // row-wise sums over samples: slot s of a register = lane s of each lane group; feature of slot s (< 8) is
  // 16 (s >> 2) + 4 g + (s & 3); lanes 8..15 carry a second quantity
  // Row sums over the samples (head weights, mid1 / mid2 biases).  Two forms: transposing DPP butterflies into "slot"
  // registers (slot s of a register = lane s of each lane group; feature of slot s (< 8) is 16 (s >> 2) + 4 g + (s & 3);
  // lanes 8..15 carry a second quantity) -- or, OBJ_LAZY_HEADS / OBJ_LAZY_BIAS, per-lane partial sums over this
  // lane's samples, reduced over the 16 lanes of the group once at the end (fewer instructions, 32 / 16 more
  // registers).  The head weights are lazy by default (10.08 -> 9.78 ms on the 50 x 4096 x 64 step); both together
  // do not fit the 256 registers of a 512-thread workgroup (OBJ_LAZY_BIAS alone: 9.91 ms).
#ifdef OBJ_LAZY_HEADS
  float hW[4][8];     // d W_alpha, d W_oc[0..2];  [.][4 tt + r] <-> feature 16 tt + 4 g + r
#pragma unroll
  for (int s_ = 0; s_ < 8; ++s_) hW[0][s_] = hW[1][s_] = hW[2][s_] = hW[3][s_] = 0.f;
#else
  float gS0 = 0.f;   // [0..7] d W_alpha   | [8..15] d W_oc[0]
  float gS1 = 0.f;   // [0..7] d W_oc[1]   | [8..15] d W_oc[2]
#endif
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
  float l_d = 0.f, l_c = 0.f, l_o = 0.f;

  // sample position of (tile, slot); issued one tile ahead (phase C) so that the HBM latency is off the tile's
  // critical path
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
  fetch_point(gi, slot, nx, ny, nz);
  PT_INIT();
  for (int tile = gi; tile < a.NT; tile += a.G) {
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
      s_alpha[g * TS + slot] = mlp32_forward<false>(wf, sv, g, e, act);
    }
    if (MASKS) {                      // test hook: ReLU branch bits of this lane's sample
      uint8_t* dst = a.relu_masks + (((long)k * R + (valid ? ray : 0)) * S + (slot - q * S)) * 24;
      write_relu_mask(dst, 0, g, act.h1, valid);
      write_relu_mask(dst, 1, g, act.h2, valid);
      write_relu_mask(dst, 2, g, act.h3, valid);
      write_relu_mask(dst, 3, g, act.h4, valid);
      write_relu_mask(dst, 4, g, act.hc, valid);
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
    __syncthreads();
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
        const float dw = gD * zz + gO + gC0 * c0 + gC1 * c1 + gC2 * c2;
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
    if (SS || rows_mode) composite_passes(seg_rows); else composite_passes(SegGeneric{S});
    PT(4);
    __syncthreads();
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

    // ---- phase A: heads, colour layer, mid2
    T32 d_hc, d_h4;
    float pa_[8];
#ifndef OBJ_LAZY_HEADS
    float pb_[8], pc_[8], pd_[8];       // head-weight gradient products, summed over the samples below
#endif
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * g + r;
        const int s = 4 * tt + r;
        const float hv = act.hc.t[tt][r];
#ifdef OBJ_LAZY_HEADS
        hW[0][s] = fmaf(da, act.h4.t[tt][r], hW[0][s]);
        hW[1][s] = fmaf(dc0, hv, hW[1][s]);
        hW[2][s] = fmaf(dc1, hv, hW[2][s]);
        hW[3][s] = fmaf(dc2, hv, hW[3][s]);
#else
        pa_[s] = da * act.h4.t[tt][r];
        pb_[s] = dc0 * hv;
        pc_[s] = dc1 * hv;
        pd_[s] = dc2 * hv;
#endif
        const float dv = fmaf(sv[SV_WOC + 2 * H + row], dc2, fmaf(sv[SV_WOC + H + row], dc1, sv[SV_WOC + row] * dc0));
        d_hc.t[tt][r] = hv > 0.0f ? dv : 0.0f;
        d_h4.t[tt][r] = sv[SV_WA + row] * da;
      }
    // group A staging: [h4 | x2] rows 0..79, h3 rows 96..127, d_hc rows 128.., d_h4pre rows 160..
#ifndef OBJ_LAZY_HEADS
    gS0 += slot_sums16(pa_, pb_, c);
    gS1 += slot_sums16(pc_, pd_, c);
    asm volatile("" : "+v"(gS0), "+v"(gS1));
#endif
    st_T32(stg_lane, 0, act.h4);
    st_T32(stg_lane, 96, act.h3);
    st_T32(stg_lane, 128, d_hc);
    mma_t32(d_h4, wt0, wt1, R_CL, d_hc);
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
    st_T32(stg_lane, 160, d_h4);
    // PE backward, x2 part (octaves 4, 5): a 16-row tile = two direction slots
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      f32x4 d_x = zero4();
      mma_t16(d_x, wt0, wt1, R_CL + 32 + 16 * T, d_hc);
      float o0, o1, o2, o3;
      pe32_x2_pair_fb(pe, 2 * T, g, d_x[0], d_x[1], dps[2 * T], o0, o1);
      pe32_x2_pair_fb(pe, 2 * T + 1, g, d_x[2], d_x[3], dps[2 * T + 1], o2, o3);
      st_T16(stg_lane, 32 + 16 * T, f32x4{o0, o1, o2, o3});
    }
    T32 d_h3 = zero32();
    mma_t32(d_h3, wt0, wt1, R_M2, d_h4);
    d_h3 = relu_mask32(d_h3, act.h3);
    PT(6);
    __syncthreads();
    PT(7);
    if (w < 7) {
      const int dTr = (w < 5) ? 128 : 160;
      const int aTr = (w < 5) ? 16 * w : 96 + 16 * (w - 5);
      wg_pair(accA0, accA1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(8);
    __syncthreads();
    PT(9);
    // ---- phase B: cat layer.  [h2 | x1] rows 0..127, d_h3pre rows 128..
    st_T32(stg_lane, 0, act.h2);
    st_T32(stg_lane, 128, d_h3);
    T32 d_h2 = zero32();
    mma_t32(d_h2, wt0, wt1, R_CAT, d_h3);
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
    mma_t32(d_h1, wt0, wt1, R_M1, d_h2);
    d_h1 = relu_mask32(d_h1, act.h1);
    // PE backward, x1 part (octaves 0..3): d x1 tile = cat^T d_h3 + in^T d_h1; a tile = one direction slot
#pragma unroll
    for (int T = 0; T < 6; ++T) {
      f32x4 d_x = zero4();
      mma_t16(d_x, wt0, wt1, R_CAT + 32 + 16 * T, d_h3);
      mma_t16(d_x, wt0, wt1, R_IN + 16 * T, d_h1);
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
    __syncthreads();
    PT(12);
    wg_pair(accB0, accB1, lane_rd + 128 * STG_LD, lane_rd + (16 * w) * STG_LD);
    PT(13);
    __syncthreads();
    // ---- phase C: in layer (x1 stays at rows 32..127) + mid1.  h1 rows 0.., d_h1pre 128.., d_h2pre 160..
    fetch_point(tile + a.G, slot, nx, ny, nz);
    st_T32(stg_lane, 0, act.h1);
    st_T32(stg_lane, 128, d_h1);
    st_T32(stg_lane, 160, d_h2);
    PT(14);
    __syncthreads();
    PT(15);
    {
      const int dTr = (w < 6) ? 128 : 160;
      const int aTr = (w < 6) ? 32 + 16 * w : 16 * (w - 6);
      wg_pair(accC0, accC1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(16);
    // The staging area is next written in phase A of the following tile, two barriers from here.
    PT(17);
  }
  PT_FLUSH();

  // ------------------------------------------------------------------ write this workgroup's slab
  float* slab = a.slab + ((long)k * a.G + gi) * a.slab_stride;
  const Layout& L = a.L;
  {
    // reference column of the staged input row this lane's accumulator column stands for
    int t_, g_;
    if (w < 5) {                                  // colour layer: [h4 | x2]
      const int rho = 16 * w + c;
      int col = rho;
      if (rho >= H) { kappa_tg(rho - H, t_, g_); col = x2_col(t_, g_); if (col >= 0) col += H; }
      wr_pair(slab, accA0, accA1, g, col, L.cl_w, H + OBJ_E2, L.cl_b);
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
    } else {
      wr_pair(slab, accC0, accC1, g, 16 * (w - 6) + c, L.m1_w, H, -1);
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
#ifdef OBJ_LAZY_HEADS
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const int rw = 16 * (s_ >> 2) + 4 * g + (s_ & 3);
        const float v2 = dpp_rowsum16(hW[0][s_]), v3 = dpp_rowsum16(hW[1][s_]);
        const float v4 = dpp_rowsum16(hW[2][s_]), v5 = dpp_rowsum16(hW[3][s_]);
        if (c == 0) { mine[64 + rw] = v2; mine[96 + rw] = v3; mine[128 + rw] = v4; mine[160 + rw] = v5; }
      }
#else
      if (c < 8) { mine[64 + row] = gS0; mine[128 + row] = gS1; }          // wa, woc1
      else { mine[96 + row] = gS0; mine[160 + row] = gS1; }                // woc0, woc2
#endif
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
    if (lane == 0) { mine[196] = e0; mine[197] = e1; mine[198] = e2; mine[199] = 0.0f; }
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
    else if (i < 200) a.loss_part[((long)k * a.G + gi) * 4 + (i - 196)] = v;
    else if (i - 200 < 3 * OBJ_NDIR) slab[L.pe_b + (i - 200)] = v;       // [4 i + g][x] = B's own row-major order
  }
}

}  // namespace

namespace objtrain {

size_t fused32_lds_bytes() { return (size_t)LDS_FLOATS32 * 4; }

void launch_train32(const TrainDev& d, void* stream) {
  objnerf_once_per_device([] {
    const int n = (int)fused32_lds_bytes();
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
    (void)hipFuncSetAttribute((const void*)train_fused32_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
  });
  const dim3 grid(d.K * d.G), blk(NTHR);
  hipStream_t st = (hipStream_t)stream;
  if (d.relu_masks) hipLaunchKernelGGL((train_fused32_kernel<true, 0>), grid, blk, fused32_lds_bytes(), st, d);
  else if (d.S == 64) hipLaunchKernelGGL((train_fused32_kernel<false, 64>), grid, blk, fused32_lds_bytes(), st, d);
  else hipLaunchKernelGGL((train_fused32_kernel<false, 0>), grid, blk, fused32_lds_bytes(), st, d);
}

}  // namespace objtrain

#ifdef PHASE_TIMING
extern "C" int objnerf_debug_phase32(unsigned long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase32), sizeof(unsigned long long) * 8 * 24) == hipSuccess ? 0 : -1;
}
#endif
