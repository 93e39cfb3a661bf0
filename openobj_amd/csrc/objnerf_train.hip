// Fused training iteration + point evaluation for K stacked hidden-32 object networks (gfx950).
//
// Replaces, for cfg.training_strategy == "hip", the reference's per-iteration op sequence
//   vmap(pe_model) -> vmap(fc_model) -> loss.step_batch_loss -> backward          (train.py:424-472)
// with ONE kernel.  A 512-thread workgroup (8 waves, two per SIMD so one wave's VALU work overlaps the
// other's MFMAs) owns one object's weights in LDS and sweeps that object's rays in tiles of 128 samples
// (whole rays).  Per tile:
//   1. every wave embeds 16 samples and runs the MLP forward chain on MFMA, activations in registers;
//   2. sigma / rgb go to LDS, a wave per ray-group composites (segmented wave scans), evaluates the
//      masked losses of loss.py and writes d(loss)/d(sigma,rgb) back in place;
//   3. every wave back-propagates its 16 samples in registers (dgrad), transposes (d_out, input) pairs
//      through LDS and the eight waves share the weight-gradient MFMAs (two 16x16 tiles each per layer
//      group), accumulating in registers across the whole sweep.
// Each workgroup then writes one partial-gradient slab; finalize_kernel sums the slabs (no atomics on
// global memory, bit-reproducible run to run).
#include <mutex>
#include "objnerf_mlp32.h"
#include "objnerf_train_common.h"
#include "objnerf_generic.h"
#include "../../include/objnerf_hip.h"

using namespace obj32;
using namespace objtrain;

namespace {

struct EvalDev {
  int K, G; long N;
  const float* params; long p_stride; const float* scale; const float* pts; const float* emb;
  float* alpha; float* color; float* hfeat;
  Layout L;
};

// stg[(rowbase + 16 tt + 4 g + r)][16 w + c] = v  for a 32-feature block / a 16-feature tile
__device__ __forceinline__ void store_T32(float* stg_lane, const int rowbase, const T32& v) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) stg_lane[(rowbase + 16 * tt + r) * STG_LD] = v.t[tt][r];
}
__device__ __forceinline__ void store_T16(float* stg_lane, const int rowbase, const f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) stg_lane[(rowbase + r) * STG_LD] = v[r];
}

// D[out 0..31][in 16 cols] += sum_s dT[out][s] * aT[in][s] over the 128 staged samples.
// The contraction index is free to permute: MFMA k-slot (step st, lane group g) takes sample 32 g + st, so a
// lane walks CONSECUTIVE samples and fetches two steps per ds_read_b64 (8-byte aligned with the 130-float
// row stride; 2c + 32g + st covers all 64 banks -> conflict-free).  dT / aT point at &stg[(row0 + c) * LD + 32 g].
__device__ __forceinline__ void wgrad_pair(f32x4& acc0, f32x4& acc1, const float* dT, const float* aT) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  dT = (const float*)__builtin_assume_aligned(dT, 8);     // row pitch 520 B, 32 g and st even: 8-byte aligned
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

__device__ __forceinline__ void write_pair(float* slab, const f32x4& a0, const f32x4& a1, const int c, const int g,
                                           const int ct, const int w_off, const int ncols, const int b_off) {
  const int col = 16 * ct + c;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o0 = 4 * g + r, o1 = 16 + 4 * g + r;
    if (col < ncols) {
      slab[w_off + o0 * ncols + col] = a0[r];
      slab[w_off + o1 * ncols + col] = a1[r];
    } else if (col == ncols && b_off >= 0) {
      slab[b_off + o0] = a0[r];
      slab[b_off + o1] = a1[r];
    }
  }
}

// Diagnostic build (-DPHASE_TIMING): per-wave s_memtime deltas of each phase of workgroup 0, read back with
// objnerf_debug_phase() (tools/phase_timing.py).  Not part of the product library.
#ifdef PHASE_TIMING
__device__ unsigned long long g_phase[8][24];
#define PT_INIT() unsigned long long pt_acc[18]; for (int i_ = 0; i_ < 18; ++i_) pt_acc[i_] = 0; \
  unsigned long long pt_t0 = __builtin_amdgcn_s_memtime()
#define PT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_acc[i] += t_ - pt_t0; pt_t0 = t_; } while (0)
#define PT_FLUSH() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 18; ++i_) g_phase[w][i_] = pt_acc[i_]; } while (0)
#else
#define PT_INIT() do {} while (0)
#define PT(i) do {} while (0)
#define PT_FLUSH() do {} while (0)
#endif

// keeps the instruction scheduler from pulling the sincos of later embedding tiles ahead of the current one
// (which overlaps nicely but needs ~50 more live registers and spills the persistent accumulators)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

#define TILE_SYNC() __syncthreads()

// ------------------------------------------------------------------------------------------------
constexpr int X1N = 6, X2N = 3;       // 16-wide embedding tiles of the two input blocks (96 and 48 padded entries)
template <bool FEAT, bool MASKS>
__global__ __launch_bounds__(NTHR) void train_fused_kernel(const TrainDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int k = blockIdx.x / a.G, gi = blockIdx.x % a.G;
  constexpr int WF = FEAT ? W_FLOATS_FEAT : W_FLOATS_NOFEAT;
  float* s_alpha = lds + WF;
  float* s_col = s_alpha + TS;
  float* stg = lds + WF + SM_FLOATS;

  stage_weights(lds, a.params + (long)k * a.p_stride, a.L, FEAT, tid, NTHR);
  for (int i = tid; i < STG_ROWS * STG_LD; i += NTHR) stg[i] = 0.0f;
  __syncthreads();

  const float scale = a.scale[k];
  const int S = a.S, R = a.R, TR = a.TR;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  int bflag0, bflag1;
  batch_flags(a, bflag0, bflag1);
  const float inv1 = bflag0 ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = bflag1 ? 0.0f : 1.0f / (n2 + 1e-10f);

  // persistent gradient accumulators
  f32x4 accA0 = zero4(), accA1 = zero4(), accB0 = zero4(), accB1 = zero4(), accC0 = zero4(), accC1 = zero4();
  // row-wise sums over samples: slot s of a register = lane s of each lane group; feature of slot s (< 8) is
  // 16 (s >> 2) + 4 g + (s & 3); lanes 8..15 carry a second quantity
  float gS0 = 0.f;   // [0..7] d W_alpha   | [8..15] d W_oc[0]
  float gS1 = 0.f;   // [0..7] d W_oc[1]   | [8..15] d W_oc[2]
  float gS2 = 0.f;   // [0..7] d b_mid1    | [8..15] d b_mid2
  float g_ba = 0.f, g_boc0 = 0.f, g_boc1 = 0.f, g_boc2 = 0.f;
  f32x4 accT0 = zero4(), accT1 = zero4();   // d B: rows j = 4g + r (accT1: 16 + 4g + r), column x = c < 3
  float l_d = 0.f, l_c = 0.f, l_o = 0.f, l_f = 0.f;
  f32x4 accF0 = zero4(), accF1 = zero4();



  // Loop-invariant per-lane LDS addresses must NOT be hoisted out of the tile loop: there are dozens of them
  // and they end up spilled to scratch, each reload a serialised ~500-cycle stall.  Inside the loop the lane
  // coordinates (c, g) and every per-lane LDS pointer are macros over an opaque copy of the lane id that is
  // re-defined at each phase boundary, so addresses are recomputed (a few VALU ops) next to their use.
  int lane_l = lane;
#define RELAUNDER() asm volatile("" : "+v"(lane_l))
#define c (lane_l & 15)
#define g (lane_l >> 4)
#define wt_fl (lds + OFF_FL + (4 * g) * ST_CL + c)
#define stg_lane (stg + (4 * g) * STG_LD + 16 * w + c)
#define lane_rd (stg + c * STG_LD + 32 * g)
#define wt_in (lds + OFF_IN + (4 * g) * ST_IN + c)
#define wt_m1 (lds + OFF_M1 + (4 * g) * ST_M + c)
#define wt_cat (lds + OFF_CAT + (4 * g) * ST_CAT + c)
#define wt_m2 (lds + OFF_M2 + (4 * g) * ST_M + c)
#define wt_cl (lds + OFF_CL + (4 * g) * ST_CL + c)
  // sample position of (tile, slot); issued one tile ahead (phase C) so that the HBM latency is off the tile's
  // critical path
  auto fetch_point = [&](const int tile_, const int slot_, float& x, float& y, float& z_) {
    const int q_ = slot_ / a.S, si_ = slot_ - q_ * a.S;
    const int ray_ = tile_ * a.TR + q_;
    x = 0.f; y = 0.f; z_ = 0.f;
    if (tile_ < a.NT && q_ < a.TR && ray_ < a.R) {
      const long rr = (long)k * a.R + ray_;
      if (a.pts) {
        const float* p = a.pts + (rr * a.S + si_) * 3;
        x = p[0]; y = p[1]; z_ = p[2];
      } else {
        const float zz = a.z[rr * a.S + si_];
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        x = (o[0] + d[0] * zz) - a.obj_center;   // vmap.py:548-551 (two roundings: -ffp-contract=off)
        y = (o[1] + d[1] * zz) - a.obj_center;
        z_ = (o[2] + d[2] * zz) - a.obj_center;
      }
    }
  };
  const bool rows_mode = seg_is_rows(a.S);
  const SegRows seg_rows = SegRows::make(rows_mode ? a.S : 64, lane);
  float nx, ny, nz;
  fetch_point(gi, 16 * w + lane_l % 16, nx, ny, nz);
  PT_INIT();
  for (int tile = gi; tile < a.NT; tile += a.G) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    RELAUNDER();
    const int ray0 = tile * TR;
    // ---------------------------------------------------------------- 1. forward
    const int slot = 16 * w + c;
    const int q = slot / S;
    const int ray = ray0 + q;
    const bool valid = (q < TR) && (ray < R);
    const float px = nx, py = ny, pz = nz;       // fetched during the previous tile's phase C
    Pe pe;
    pe_project(lds, g, px, py, pz, scale, pe);
    PT(0);
    Acts act;
    Heads hd;
    {
      Emb e;                          // forward-only: the backward re-creates the embedding tile by tile
      embed(e, pe, g);
      PT(1);
      mlp_forward<FEAT>(lds, c, g, e, act, hd);
    }
    PT(2);
    if (MASKS) {                      // test hook: ReLU branch bits of this lane's sample
      uint8_t* dst = a.relu_masks + (((long)k * R + (valid ? ray : 0)) * S + (slot - q * S)) * 24;
      write_relu_mask(dst, 0, lane >> 4, act.h1, valid);
      write_relu_mask(dst, 1, lane >> 4, act.h2, valid);
      write_relu_mask(dst, 2, lane >> 4, act.h3, valid);
      write_relu_mask(dst, 3, lane >> 4, act.h4, valid);
      write_relu_mask(dst, 4, lane >> 4, act.hc, valid);
      if (FEAT) write_relu_mask(dst, 5, lane >> 4, act.hf, valid);
    }
    if (g == 0) {
      s_alpha[slot] = hd.alpha;
      s_col[slot] = hd.col[0];
      s_col[TS + slot] = hd.col[1];
      s_col[2 * TS + slot] = hd.col[2];
    }
    if (FEAT) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[slot * HF_LD + 16 * tt + 4 * g + r] = act.hf.t[tt][r];
      for (int i = tid; i < 32 * 32 + 33; i += NTHR) {       // this object's Gram matrix (+ wb, bb) for the tile
        const float v = a.gram[(long)k * GRAM + i];
        if (i < 1024) stg[OFF_GBUF + (i >> 5) * 33 + (i & 31)] = v;
        else stg[OFF_GBUF + 32 * 33 + (i - 1024)] = v;
      }
    }
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
        pf_uh = a.rayin[rr2_ * RAYIN + (lane_l & 31)];
        pf_beta = a.rayin[rr2_ * RAYIN + 32];
        pf_ngv = a.rayin[rr2_ * RAYIN + 33];
        pf_lab2 = (int)a.labels[rr2_];
      }
    } else if (FEAT && w * (64 / S) < TR) feat_inputs(w, 0, pf_uh, pf_beta, pf_ngv, pf_lab2);
    TILE_SYNC();
    RELAUNDER();
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
          float* s_w = stg + OFF_SW;
          float* s_gfh = stg + OFF_GFH;
          float* s_gof = stg + OFF_GOF;
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
                // four independent partial sums: the LDS reads of several steps are in flight together
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
      float* s_w = stg + OFF_SW;
      float* s_gfh = stg + OFF_GFH;
      float* s_gof = stg + OFF_GOF;
      float* s_part = s_gfh + 192;          // [NWAVE][32] partial composited features (s_gfh holds 2 rays here)
      float* s_dwf = s_gfh + 64;            // [128]
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
      __syncthreads();
      {
        float* s_fhb = stg + OFF_FHB + 64 * w;
        const float* Gb = stg + OFF_GBUF;
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
    } else if (rows_mode) composite_passes(seg_rows); else composite_passes(SegGeneric{S});
    PT(4);
    TILE_SYNC();
    RELAUNDER();
    PT(5);
    // ---------------------------------------------------------------- 3. backward
    const float da = valid ? s_alpha[slot] : 0.0f;
    const float dc0 = valid ? s_col[slot] : 0.0f;
    const float dc1 = valid ? s_col[TS + slot] : 0.0f;
    const float dc2 = valid ? s_col[2 * TS + slot] : 0.0f;
    if (g == 0) { g_ba += da; g_boc0 += dc0; g_boc1 += dc1; g_boc2 += dc2; }
    // The sincos of the embedding is RE-computed below: hide ps from CSE, otherwise the compiler keeps every
    // forward cos value live across the whole backward pass.
#pragma unroll
    for (int j = 0; j < OBJ_NDIR; ++j) asm volatile("" : "+v"(pe.ps[j]));
    float dps[OBJ_NDIR];
#pragma unroll
    for (int j = 0; j < OBJ_NDIR; ++j) dps[j] = 0.f;

    // ---- phase A: heads, colour layer, (feature layer,) mid2
    T32 d_hf = zero32();
    if (FEAT) {
      const float wv = valid ? stg[OFF_SW + slot] : 0.0f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gv = valid ? stg[OFF_GFH + q * 32 + 16 * tt + 4 * g + r] : 0.0f;
          d_hf.t[tt][r] = act.hf.t[tt][r] > 0.0f ? wv * gv : 0.0f;
        }
      // (rows 80..95, where s_w / gfh live, are not touched by the phase-A staging below)
    }
    T32 d_hc, d_h4;
    float pa_[8], pb_[8], pc_[8], pd_[8];       // head-weight gradient products, summed over the samples below
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * g + r;
        const int s = 4 * tt + r;
        const float hv = act.hc.t[tt][r];
        pa_[s] = da * act.h4.t[tt][r];
        pb_[s] = dc0 * hv;
        pc_[s] = dc1 * hv;
        pd_[s] = dc2 * hv;
        const float dv = fmaf(lds[OFF_WOC + 2 * H + row], dc2, fmaf(lds[OFF_WOC + H + row], dc1, lds[OFF_WOC + row] * dc0));
        d_hc.t[tt][r] = hv > 0.0f ? dv : 0.0f;
        d_h4.t[tt][r] = lds[OFF_WA + row] * da;
      }
    // group A staging: [h4 | x2] rows 0..79, h3 rows 96..127, d_hc rows 128.., d_h4pre rows 160..
    gS0 += slot_sums16(pa_, pb_, c);
    gS1 += slot_sums16(pc_, pd_, c);
    asm volatile("" : "+v"(gS0), "+v"(gS1));
    store_T32(stg_lane, 0, act.h4);
    store_T32(stg_lane, 96, act.h3);
    store_T32(stg_lane, 128, d_hc);
    mma_bwd32<ST_CL>(d_h4, wt_cl, 0, d_hc);
    if (FEAT) mma_bwd32<ST_CL>(d_h4, wt_fl, 0, d_hf);
    d_h4 = relu_mask32(d_h4, act.h4);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pa_[4 * tt + r] = d_h4.t[tt][r];
    {
      const float sv = slot_sums8(pa_, c);
      gS2 += (c >= 8) ? sv : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
    store_T32(stg_lane, 160, d_h4);
    // PE backward, x2 part, one 16-row tile at a time
#pragma unroll
    for (int T = 0; T < X2N; ++T) {
      f32x4 d_x = zero4();
      mma_bwd16<ST_CL>(d_x, wt_cl, 32 + 16 * T, d_hc);
      if (FEAT) mma_bwd16<ST_CL>(d_x, wt_fl, 32 + 16 * T, d_hf);
      store_T16(stg_lane, 32 + 16 * T, pe_x2_tile_fb(pe, T, g, d_x, dps));
    }
    T32 d_h3 = zero32();
    mma_bwd32<ST_M>(d_h3, wt_m2, 0, d_h4);
    d_h3 = relu_mask32(d_h3, act.h3);
    PT(6);
    TILE_SYNC();
    RELAUNDER();
    PT(7);
    if (w < 7) {
      const int dTr = (w < 5) ? 128 : 160;
      const int aTr = (w < 5) ? 16 * w : 96 + 16 * (w - 5);
      wgrad_pair(accA0, accA1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(8);
    TILE_SYNC();
    RELAUNDER();
    PT(9);
    if (FEAT) {             // feature layer weight gradient: same inputs [h4 | x2], d_hf in place of d_hc
      store_T32(stg_lane, 128, d_hf);
      TILE_SYNC();
    RELAUNDER();
      if (w < 5) wgrad_pair(accF0, accF1, lane_rd + 128 * STG_LD, lane_rd + (16 * w) * STG_LD);
      TILE_SYNC();
    RELAUNDER();
    }
    // ---- phase B: cat layer.  [h2 | x1] rows 0..127, d_h3pre rows 128..
    store_T32(stg_lane, 0, act.h2);
    store_T32(stg_lane, 128, d_h3);
    T32 d_h2 = zero32();
    mma_bwd32<ST_CAT>(d_h2, wt_cat, 0, d_h3);
    d_h2 = relu_mask32(d_h2, act.h2);
    float pa2_[8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pa2_[4 * tt + r] = d_h2.t[tt][r];
    {
      const float sv = slot_sums8(pa2_, c);
      gS2 += (c < 8) ? sv : 0.0f;
      asm volatile("" : "+v"(gS2));
    }
    T32 d_h1 = zero32();
    mma_bwd32<ST_M>(d_h1, wt_m1, 0, d_h2);
    d_h1 = relu_mask32(d_h1, act.h1);
    // PE backward, x1 part: d x1 tile = cat^T d_h3 + in^T d_h1, consumed tile by tile
#pragma unroll
    for (int T = 0; T < X1N; ++T) {
      f32x4 d_x = zero4();
      mma_bwd16<ST_CAT>(d_x, wt_cat, 32 + 16 * T, d_h3);
      mma_bwd16<ST_IN>(d_x, wt_in, 16 * T, d_h1);
      store_T16(stg_lane, 32 + 16 * T, pe_x1_tile_fb(pe, T, g, d_x, dps));
    }
    PT(10);
    // d ps[i] of lane group g belongs to direction j = (i + 4g) mod 21, doubled when it wrapped into the next
    // octave.  The four groups of a sample are summed ON THE MATRIX CORE: one 16x16x4 MFMA per register i with
    // B = d ps[i] (k = lane group, n = sample) and a 0/1/2 selection matrix A[row][k] = [row == j(i, k)] * factor
    // scatters the four values into rows j of a 32-row d-projection tile (exact: products by 0, 1, 2).  No LDS
    // atomics (ds_add_f32 retires ~1 lane/clk for the whole CU: 8 waves x 21 of them cost ~10 % of the kernel).
    {
      T32 dpj = zero32();
#pragma unroll
      for (int i = 0; i < OBJ_NDIR; ++i) {
        bool need[2] = {false, false};
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int mm = 4 * gg + i;
          need[((i > 8 && mm >= OBJ_NDIR) ? mm - OBJ_NDIR : mm) >> 4] = true;
        }
        const int m = 4 * g + i;
        const bool wrap = (i > 8) && (m >= OBJ_NDIR);
        const int j = wrap ? m - OBJ_NDIR : m;
        const float f = wrap ? 2.0f : 1.0f;
        if (need[0]) dpj.t[0] = OBJ_MFMA((j == c) ? f : 0.0f, dps[i], dpj.t[0]);
        if (need[1]) dpj.t[1] = OBJ_MFMA((j - 16 == c) ? f : 0.0f, dps[i], dpj.t[1]);
      }
      store_T32(stg_lane, 160, dpj);      // rows 160..180 (181..191: zeros), free in phase B
      // d B[j][x] += sum over THIS wave's 16 samples of dproj[j][s] t[x][s] (embedding.py:48): the wave re-reads
      // its own columns as MFMA operands (A = table rows, B = the t rows 32..34 of the x1 staging)
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      const float* ta = stg + (160 + c) * STG_LD + 16 * w + g;
      const float* tb = stg + (32 + (c < 3 ? c : 2)) * STG_LD + 16 * w + g;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const float bv = (c < 3) ? tb[4 * st] : 0.0f;
        accT0 = OBJ_MFMA(ta[4 * st], bv, accT0);
        accT1 = OBJ_MFMA(ta[16 * STG_LD + 4 * st], bv, accT1);
      }
    }
    PT(11);
    TILE_SYNC();
    RELAUNDER();
    PT(12);
    wgrad_pair(accB0, accB1, lane_rd + 128 * STG_LD, lane_rd + (16 * w) * STG_LD);
    PT(13);
    TILE_SYNC();
    RELAUNDER();
    // ---- phase C: in layer (x1 stays at rows 32..127) + mid1.  h1 rows 0.., d_h1pre 128.., d_h2pre 160..
    fetch_point(tile + a.G, slot, nx, ny, nz);
    store_T32(stg_lane, 0, act.h1);
    store_T32(stg_lane, 128, d_h1);
    store_T32(stg_lane, 160, d_h2);
    PT(14);
    TILE_SYNC();
    RELAUNDER();
    PT(15);
    {
      const int dTr = (w < 6) ? 128 : 160;
      const int aTr = (w < 6) ? 32 + 16 * w : 16 * (w - 6);
      wgrad_pair(accC0, accC1, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    PT(16);
    // The staging area is next written in phase A of the following tile, two barriers from here; only the feature
    // build writes it earlier (its hidden-feature buffer after the forward pass) and needs this barrier.
    if (FEAT) TILE_SYNC();
    RELAUNDER();
    PT(17);
  }
  PT_FLUSH();
#undef c
#undef g
#undef wt_fl
#undef stg_lane
#undef lane_rd
#undef wt_in
#undef wt_m1
#undef wt_cat
#undef wt_m2
#undef wt_cl

  // ------------------------------------------------------------------ write this workgroup's slab
  float* slab = a.slab + ((long)k * a.G + gi) * a.slab_stride;
  const Layout& L = a.L;
  if (w < 5) write_pair(slab, accA0, accA1, c, g, w, L.cl_w, H + OBJ_E2, L.cl_b);
  else if (w < 7) write_pair(slab, accA0, accA1, c, g, w - 5, L.m2_w, H, -1);
  write_pair(slab, accB0, accB1, c, g, w, L.cat_w, H + OBJ_E1, L.cat_b);
  if (w < 6) write_pair(slab, accC0, accC1, c, g, w, L.in_w, OBJ_E1, L.in_b);
  else write_pair(slab, accC0, accC1, c, g, w - 6, L.m1_w, H, -1);
  if (FEAT && w < 5) write_pair(slab, accF0, accF1, c, g, w, L.fl_w, H + OBJ_E2, L.fl_b);
  // slot registers -> LDS (per wave), then sum the 8 waves
  __syncthreads();    // the last tile's weight-gradient reads of the staging area are done
  float* red = stg;   // [NWAVE][NRED]
  {
    float* mine = red + w * NRED;
    const int s = c & 7;
    const int row = 16 * (s >> 2) + 4 * g + (s & 3);
    if (c < 8) { mine[64 + row] = gS0; mine[128 + row] = gS1; mine[row] = gS2; }          // wa, woc1, bm1
    else { mine[96 + row] = gS0; mine[160 + row] = gS1; mine[32 + row] = gS2; }           // woc0, woc2, bm2
    const float s0 = wave_sum64(g_ba), s1 = wave_sum64(g_boc0), s2 = wave_sum64(g_boc1), s3 = wave_sum64(g_boc2);
    if (lane == 0) { mine[192] = s0; mine[193] = s1; mine[194] = s2; mine[195] = s3; }
    const float e0 = wave_sum64(l_d), e1 = wave_sum64(l_c), e2 = wave_sum64(l_o);
    const float e3 = wave_sum64(l_f);
    if (lane == 0) { mine[196] = e0; mine[197] = e1; mine[198] = e2; mine[199] = e3; }
    if (c < 3) {
      float* dbw = red + NWAVE * NRED + w * 64;            // d B partial of this wave: [21 * 3]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dbw[(4 * g + r) * 3 + c] = accT0[r];
        if (16 + 4 * g + r < OBJ_NDIR) dbw[(16 + 4 * g + r) * 3 + c] = accT1[r];
      }
    }
  }
  __syncthreads();
  if (tid < 3 * OBJ_NDIR) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[NWAVE * NRED + ww * 64 + tid];
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

// ------------------------------------------------------------------------------------------------
// The step's last launch: grads[k][i] = sum_g slab[k][g][i] for every entry this configuration differentiates,
// loss_terms[k][:] = sum_g loss_part, the status word -- and, with an optimiser attached (objnerf_train_args.optim),
// torch.optim.AdamW on the element just summed (adamw_dyn_kernel's arithmetic; objnerf_misc.hip).
// No byte mask and no zero fill: [ng_lo, ng_hi) are the entries WITHOUT a gradient (the feature branch when gt_feat is
// NULL: .grad stays None, train.py:435-438), [ext_lo, ext_hi) the entries whose gradient another kernel has already
// written to `grads` (the 512-d head: feat_finish_kernel).  Block (0, 0) also sums the loss terms of ALL objects and
// WRITES the status word (bit 0: a term above 1e5, render_rays.py:109-111; bit 1: a term that is not finite).
// flat_nwg > 0 (objnerf_train_common.h): object k's partial slabs are those of the workgroups whose share touches it,
// slot 0 .. cnt - 1 of its Gs
constexpr int XCOLS = 33;      // [fh (32) | O]: the columns of the 512-d head's per-ray records and moments
struct FinalizeArgs {
  const float* slab; const float* loss_part;
  int K, G; long P, slab_stride, p_stride;
  long ng_lo, ng_hi, ext_lo, ext_hi;
  float* grads; float* loss_terms; int* status;
  int flat_nwg, NT, Gs;
  // AdamW (params == NULL: gradients only)
  float* params; float* m; float* v; const int* flags; int* steps; int bank;
  long lo1, lo2, hi2;
  double lr, b1, b2, wd; float eps;
  const int* counts_in; int* flags_out;      // OBJNERF_TRAIN_SELF_COUNTS: derive the flags from counts [K][2], publish them
  // the 512-d head's gradient, folded in (round 6; feat_finish_kernel's arithmetic, element for element): Tpart
  // [K][Gfin][C][33], Mpart [K][Gfin][33][33] of feat_post_kernel; NULL: the entries [ext_lo, ext_hi) of `grads` were
  // written by an earlier launch.  head_w [K][C 32 + C]: feat_pre_kernel's copy of [W_of | b_of] from BEFORE the step (the
  // head gradient of an entry reads a whole row of W_of and b_of, which other threads of this launch are stepping)
  const float* Tpart; const float* Mpart; const float* head_w; int Gfin, C; long of_w, of_b;
};
// Round 6: FIN_EPT elements per thread (their slab loads are independent: FIN_EPT x G requests in flight instead of a
// chain of G latencies per thread), so an object is ~30 workgroups instead of 120 and the whole grid is resident at once
// -- the double-precision pow() preamble of the optimiser (a few microseconds of one thread, ahead of the block's
// barrier) is paid ONCE, in parallel, instead of once per round of the chip (118 -> 53 us at 50 objects, 79 beside the background chain).
constexpr int FIN_EPT = 4;
__global__ __launch_bounds__(256) void finalize_kernel(const FinalizeArgs a) {
  const int k = blockIdx.y;
  const long i0 = (long)blockIdx.x * (256 * FIN_EPT) + threadIdx.x;
  __shared__ float s_step_size[3], s_bc2_sqrt[3];
  __shared__ int s_active[3];
  __shared__ int s_bad;
  __shared__ float s_M[XCOLS * XCOLS];
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  if (a.counts_in && first && threadIdx.x == 64) {     // derived flags: published for the host / later launches
    int e0 = 0, e1 = 0;
    for (int kk = 0; kk < a.K; ++kk) { e0 |= a.counts_in[2 * kk] == 0; e1 |= a.counts_in[2 * kk + 1] == 0; }
    a.flags_out[0] = e0; a.flags_out[1] = e1;
  }
  if (a.params && threadIdx.x < 3) {          // (as adamw_dyn_kernel: read bank `bank`, the first block writes the other)
    const int g = threadIdx.x;
    bool f0, f1;
    if (a.counts_in) {
      int e0 = 0, e1 = 0;
      for (int kk = 0; kk < a.K; ++kk) { e0 |= a.counts_in[2 * kk] == 0; e1 |= a.counts_in[2 * kk + 1] == 0; }
      f0 = e0 != 0; f1 = e1 != 0;
    } else {
      f0 = a.flags[0] != 0; f1 = a.flags[1] != 0;
    }
    s_active[g] = g == 0 ? !(f0 && f1) : !f0;
    const int old = a.steps[3 * a.bank + g];
    const double st = (double)(old + 1);
    s_step_size[g] = (float)(a.lr / (1.0 - pow(a.b1, st)));
    s_bc2_sqrt[g] = (float)sqrt(1.0 - pow(a.b2, st));
    if (first) a.steps[3 * (1 - a.bank) + g] = old + (s_active[g] ? 1 : 0);
  }
  if (first && threadIdx.x == 0) s_bad = 0;
  // the head's moments (workgroups that hold entries of [ext_lo, ext_hi) only): M = sum of the chunks' partials, in chunk order
  const long blk_lo = (long)blockIdx.x * (256 * FIN_EPT), blk_hi = blk_lo + 256 * FIN_EPT;
  const bool head_blk = a.Tpart && blk_lo < a.ext_hi && blk_hi > a.ext_lo;
  if (head_blk)
    for (int e = threadIdx.x; e < XCOLS * XCOLS; e += 256) {
      float v = 0.f;
      for (int g = 0; g < a.Gfin; ++g) v += a.Mpart[((long)k * a.Gfin + g) * XCOLS * XCOLS + e];
      s_M[e] = v;
    }
  if (a.params || first || head_blk) __syncthreads();
  int G = a.G;
  if (a.flat_nwg) {
    const long T = (long)a.K * a.NT;
    G = flat_wg_of(T, a.flat_nwg, (long)(k + 1) * a.NT - 1) - flat_wg_of(T, a.flat_nwg, (long)k * a.NT) + 1;
  }
  const int stride = a.flat_nwg ? a.Gs : a.G;
  float s[FIN_EPT];
  bool live[FIN_EPT], slabbed[FIN_EPT];
#pragma unroll
  for (int e = 0; e < FIN_EPT; ++e) {
    const long i = i0 + 256 * e;
    live[e] = i < a.P && !(i >= a.ng_lo && i < a.ng_hi);
    slabbed[e] = live[e] && !(i >= a.ext_lo && i < a.ext_hi);
    s[e] = 0.f;
  }
  const float* sl0 = a.slab + (long)k * stride * a.slab_stride + i0;
  for (int g = 0; g < G; ++g) {               // (slab order: one association for every run)
    float t[FIN_EPT];
#pragma unroll
    for (int e = 0; e < FIN_EPT; ++e) t[e] = slabbed[e] ? sl0[(long)g * a.slab_stride + 256 * e] : 0.f;
#pragma unroll
    for (int e = 0; e < FIN_EPT; ++e) s[e] += t[e];
  }
#pragma unroll
  for (int e = 0; e < FIN_EPT; ++e) {
    const long i = i0 + 256 * e;
    if (!live[e]) continue;
    const long idx = (long)k * a.p_stride + i;
    float sv_ = s[e];
    if (!slabbed[e]) {
      if (a.Tpart) {
        // d W_of[c][h] = T[c][h] + sum_j W_of[c][j] M2[j][h] + b_of[c] m1[h];  d b_of[c] = T[c][32] + W_of[c] . m1 + b_of[c] s2
        const bool is_b = i >= a.of_b;
        const int cc = is_b ? (int)(i - a.of_b) : (int)((i - a.of_w) >> 5), hh = is_b ? 32 : (int)((i - a.of_w) & 31);
        const float* hw = a.head_w + (long)k * ((long)a.C * 33);
        const float* W = hw + cc * 32;
        const float bc = hw[a.C * 32 + cc];
        float v = 0.f;
        for (int g = 0; g < a.Gfin; ++g) v += a.Tpart[((long)k * a.Gfin + g) * a.C * XCOLS + cc * XCOLS + hh];
        if (!is_b) {
          for (int j = 0; j < 32; ++j) v = fmaf(W[j], s_M[j * XCOLS + hh], v);
          sv_ = fmaf(bc, s_M[32 * XCOLS + hh], v);
        } else {
          for (int j = 0; j < 32; ++j) v = fmaf(W[j], s_M[32 * XCOLS + j], v);
          sv_ = fmaf(bc, s_M[32 * XCOLS + 32], v);
        }
      } else {
        sv_ = a.grads[idx];
      }
    }
    s[e] = sv_;
  }
#pragma unroll
  for (int e = 0; e < FIN_EPT; ++e) {
    const long i = i0 + 256 * e;
    if (!live[e]) continue;
    const long idx = (long)k * a.p_stride + i;
    const float sv_ = s[e];
    if (slabbed[e] || a.Tpart) a.grads[idx] = sv_;
    if (a.params) {
      const int g = (i >= a.lo1 && i < a.lo2) ? 1 : ((i >= a.lo2 && i < a.hi2) ? 2 : 0);
      if (s_active[g]) {
        const float decay = (float)(1.0 - a.lr * a.wd), w1 = (float)(1.0 - a.b1), w2 = (float)(1.0 - a.b2), beta2 = (float)a.b2;
        float p = a.params[idx] * decay;
        const float mo = a.m[idx];
        const float mn = mo + w1 * (sv_ - mo);
        const float vn = a.v[idx] * beta2 + (w2 * sv_) * sv_;
        const float denom = sqrtf(vn) / s_bc2_sqrt[g] + a.eps;
        p = p + (-s_step_size[g]) * (mn / denom);
        a.params[idx] = p;
        a.m[idx] = mn;
        a.v[idx] = vn;
      }
    }
  }
  if (first) {
    int bad = 0;
    for (int e = threadIdx.x; e < 4 * a.K; e += blockDim.x) {
      const int kk = e >> 2;
      int Gk = a.G;
      if (a.flat_nwg) {
        const long T = (long)a.K * a.NT;
        Gk = flat_wg_of(T, a.flat_nwg, (long)(kk + 1) * a.NT - 1) - flat_wg_of(T, a.flat_nwg, (long)kk * a.NT) + 1;
      }
      float sl = 0.f;
      for (int g = 0; g < Gk; ++g) sl += a.loss_part[((long)kk * stride + g) * 4 + (e & 3)];
      a.loss_terms[e] = sl;
      if (sl > 100000.0f) bad |= 1;          // render_rays.py:109-111
      if (!(fabsf(sl) <= 3.0e38f)) bad |= 2;   // NaN / Inf (the reference carries on with NaN parameters)
    }
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) *a.status = s_bad;
  }
}

// ------------------------------------------------------------------------------------------------
// Inference: 4 independent waves per workgroup, 16 points per wave per step (second-generation chain, objnerf_mlp32.h).
template <bool FEAT, bool FROM_EMB>
__global__ __launch_bounds__(256) void eval_kernel(const EvalDev a) {
  using namespace obj32n;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int k = blockIdx.x / a.G, gi = blockIdx.x % a.G;
  stage_weights32(lds, a.params + (long)k * a.p_stride, a.L, FEAT, tid, 256);
  const float* sv = lds + sv_base(FEAT);
  const float* wf = (const float*)__builtin_assume_aligned(lds + 4 * g * WROW + out_pos(c), 8);
  const float scale = FROM_EMB ? 1.0f : a.scale[k];
  const long ntiles = (a.N + 63) / 64;
  for (long tile = gi; tile < ntiles; tile += a.G) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    const long n = tile * 64 + 16 * w + c;
    const bool valid = n < a.N;
    const long o = (long)k * a.N + (valid ? n : 0);
    Emb32 e;
    if (FROM_EMB) {
      embed32_load(e, a.emb + o * OBJ_EMB, g);
    } else {
      const float* p = a.pts + o * 3;
      Pe32 pe;
      pe32_project(sv, g, p[0], p[1], p[2], scale, pe);
      embed32(e, pe, g);
    }
    Acts act;
    const float hout = mlp32_forward<FEAT>(wf, sv, g, e, act);    // group 0: 10 * raw alpha, groups 1..3: colour g - 1
    if (valid) {
      if (g == 0) a.alpha[o] = hout;
      else a.color[o * 3 + g - 1] = hout;
      if (FEAT) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          float4 v = make_float4(act.hf.t[tt][0], act.hf.t[tt][1], act.hf.t[tt][2], act.hf.t[tt][3]);
          *reinterpret_cast<float4*>(a.hfeat + o * H + 16 * tt + 4 * g) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Feature head hoisting, per-step helper kernels (HBM-bound on gt_feat, read twice per step).
// pre: u[r] = W_of^T g[r] is a batched GEMM (objgen::gemm_f32); this kernel adds beta[r] = b_of . g[r] and |g[r]|:
// one 16-lane group per ray, float4 loads.
__global__ __launch_bounds__(256) void feat_rowstats_kernel(const float* params, long p_stride, int off_b, int C, int R,
                                                            const float* gt_feat, float* rayin) {
  const int k = blockIdx.y;
  const int l16 = threadIdx.x & 15;
  const long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const float* B = params + (long)k * p_stride + off_b;
  float bs = 0.f, gs = 0.f;
  if (r < R) {
    const float* gp = gt_feat + ((long)k * R + r) * C;
    for (int cc = 4 * l16; cc < C; cc += 64) {
      const float4 gv = *reinterpret_cast<const float4*>(gp + cc);
      bs = fmaf(B[cc + 3], gv.w, fmaf(B[cc + 2], gv.z, fmaf(B[cc + 1], gv.y, fmaf(B[cc], gv.x, bs))));
      gs = fmaf(gv.w, gv.w, fmaf(gv.z, gv.z, fmaf(gv.y, gv.y, fmaf(gv.x, gv.x, gs))));
    }
  }
  bs = dpp_rowsum16(bs);
  gs = dpp_rowsum16(gs);
  if (r < R && l16 == 0) {
    float* o = rayin + ((long)k * R + r) * RAYIN;
    o[32] = bs;
    o[33] = sqrtf(gs);
  }
}
// The same three quantities in ONE pass over gt_feat (round 3): rayin[r] = [u (32) | beta | |g|] with u = W_of^T g[r],
// beta = b_of . g[r].  A workgroup of 8 waves takes 128 rays; [W_of | b_of] sits in LDS as the B operand of
// v_mfma_f32_16x16x4_f32 (two 16-column tiles of u; beta and |g|^2 on the VALU); a lane streams ITS ray's features as float4 --
// k-step j of block s uses feature 16 s + 4 q + j, a permutation of the contraction both operands share -- and squares
// them on the way for |g|.  gt_feat is read once instead of twice (GEMM + feat_rowstats_kernel: 240 + 101 us at
// K = 50, R = 4096 -> ~100 us).  C <= 512, C % 16 == 0.
constexpr int FEAT_PRE_MAXC = 512;
__global__ __launch_bounds__(512, 4) void feat_pre_kernel(const float* __restrict__ params, long p_stride, int off_w, int off_b,
                                                       int C, int R, const float* __restrict__ gt_feat,
                                                       float* __restrict__ rayin, float* __restrict__ gram,
                                                       float* __restrict__ head_snap, const uint8_t* __restrict__ labels,
                                                       int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) float fp_lds[];       // W image [C / 16][2][64][4] | b_of [C]
  const int k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const float* W = params + (long)k * p_stride + off_w;
  const float* Bv = params + (long)k * p_stride + off_b;
  const int nblk = C / 16;
  // B image: W_of row-major [c][h] read coalesced, scattered into (block c >> 4, tile h >> 4, lane (h & 15) + 16 ((c >> 2) & 3),
  // element c & 3).  Round 6: beta = b_of . g runs on the VALU beside |g|^2 (4 FMAs per float4 instead of the four MFMAs
  // of a third, 15/16 empty column tile: a third of the kernel's matrix-core time), b_of is a plain vector behind the
  // image, and the 66 KB of LDS let TWO workgroups share a compute unit.
  float* fp_b = fp_lds + nblk * 2 * 64 * 4;
  for (int i = tid; i < C * 32; i += 512) {
    const int c = i >> 5, hc = i & 31;
    fp_lds[(((c >> 4) * 2 + (hc >> 4)) * 64 + (hc & 15) + 16 * ((c >> 2) & 3)) * 4 + (c & 3)] = W[i];
  }
  for (int i = tid; i < C; i += 512) fp_b[i] = Bv[i];
  __syncthreads();
  if (blockIdx.x == 0) {
    // Round 6: the object's first workgroup also forms what used to be two launches ahead of this one (a batched GEMM +
    // featg_wb_kernel) -- G = W_of^T W_of, wb = W_of^T b_of, bb = b_of . b_of for the hoisted head, from
    // the image just staged -- and a copy of [W_of | b_of] as they are BEFORE the step: finalize_kernel forms the head's
    // gradient from it while the optimiser in the same launch is already writing the parameters.
    auto lw = [&](const int c, const int h) {
      return fp_lds[(((c >> 4) * 2 + (h >> 4)) * 64 + (h & 15) + 16 * ((c >> 2) & 3)) * 4 + (c & 3)];
    };
    auto lb = [&](const int c) { return fp_b[c]; };
    if (gram)
      for (int o = tid; o < 32 * 32 + 33; o += 512) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (o < 1024) {
          const int h1 = o >> 5, h2 = o & 31;
          for (int c = 0; c < C; c += 4) {
            s0 = fmaf(lw(c, h1), lw(c, h2), s0); s1 = fmaf(lw(c + 1, h1), lw(c + 1, h2), s1);
            s2 = fmaf(lw(c + 2, h1), lw(c + 2, h2), s2); s3 = fmaf(lw(c + 3, h1), lw(c + 3, h2), s3);
          }
        } else if (o < 1056) {
          const int h = o - 1024;
          for (int c = 0; c < C; c += 4) {
            s0 = fmaf(lw(c, h), lb(c), s0); s1 = fmaf(lw(c + 1, h), lb(c + 1), s1);
            s2 = fmaf(lw(c + 2, h), lb(c + 2), s2); s3 = fmaf(lw(c + 3, h), lb(c + 3), s3);
          }
        } else {
          for (int c = 0; c < C; c += 4) {
            s0 = fmaf(lb(c), lb(c), s0); s1 = fmaf(lb(c + 1), lb(c + 1), s1);
            s2 = fmaf(lb(c + 2), lb(c + 2), s2); s3 = fmaf(lb(c + 3), lb(c + 3), s3);
          }
        }
        gram[(long)k * GRAM + o] = (s0 + s1) + (s2 + s3);
      }
    if (head_snap) {
      float* hs = head_snap + (long)k * ((long)C * 33);
      for (int i = tid; i < C * 32; i += 512) hs[i] = W[i];
      for (int i = tid; i < C; i += 512) hs[C * 32 + i] = Bv[i];
    }
    if (counts) {      // OBJNERF_TRAIN_SELF_COUNTS: this object's n(label == 1), n(label != 2) (label_counts_kernel's job: a launch less)
      __shared__ int s_c[16];
      int c1 = 0, c2 = 0;
      for (int r = tid; r < R; r += 512) {
        const int l = labels[(long)k * R + r];
        c1 += (l == 1); c2 += (l != 2);
      }
      for (int d = 32; d >= 1; d >>= 1) { c1 += __shfl_xor(c1, d, 64); c2 += __shfl_xor(c2, d, 64); }
      if (lane == 0) { s_c[2 * w] = c1; s_c[2 * w + 1] = c2; }
      __syncthreads();
      if (tid < 2) {
        int t = 0;
        for (int q_ = 0; q_ < 8; ++q_) t += s_c[2 * q_ + tid];
        counts[2 * k + tid] = t;
      }
    }
  }
  const f32x4* bl = reinterpret_cast<const f32x4*>(fp_lds) + lane;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  constexpr int PF = 8;                                                 // float4 loads in flight per lane: 8 blocks ahead
  for (long t0 = (long)blockIdx.x * 128; t0 < R; t0 += (long)gridDim.x * 128) {
    const long r = t0 + 16 * w + n;                                     // this lane's ray (as A row)
    const bool on = r < R;
    const float* gp = gt_feat + ((long)k * R + (on ? r : 0)) * C + 4 * q;
    f32x4 acc0 = zero, acc1 = zero;
    float gs = 0.f, bs = 0.f;
    f32x4 buf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) buf[i] = (on && i < nblk) ? *reinterpret_cast<const f32x4*>(gp + 16 * i) : zero;
    for (int s0 = 0; s0 < nblk; s0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int sblk = s0 + i;
        const f32x4 av = buf[i];
        buf[i] = (on && sblk + PF < nblk) ? *reinterpret_cast<const f32x4*>(gp + 16 * (sblk + PF)) : zero;
        if (sblk < nblk) {
          const f32x4 b0 = bl[(sblk * 2 + 0) * 64], b1 = bl[(sblk * 2 + 1) * 64];
          const f32x4 bv = *reinterpret_cast<const f32x4*>(fp_b + 16 * sblk + 4 * q);      // b_of of this lane's four features
          gs = fmaf(av[3], av[3], fmaf(av[2], av[2], fmaf(av[1], av[1], fmaf(av[0], av[0], gs))));
          bs = fmaf(av[3], bv[3], fmaf(av[2], bv[2], fmaf(av[1], bv[1], fmaf(av[0], bv[0], bs))));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b1[j], acc1, 0, 0, 0);
          }
        }
      }
    }
    // |g|^2 and beta: the four lanes (q) of a ray
    gs += __shfl_xor(gs, 16, 64);
    gs += __shfl_xor(gs, 32, 64);
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    if (on && q == 0) {
      rayin[((long)k * R + r) * RAYIN + 32] = bs;
      rayin[((long)k * R + r) * RAYIN + 33] = sqrtf(gs);
    }
    // D: lane (n, q) register rr = row 4 q + rr (ray), column n
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const long ro = t0 + 16 * w + 4 * q + rr;
      if (ro < R) {
        float* o = rayin + ((long)k * R + ro) * RAYIN;
        o[n] = acc0[rr];
        o[16 + n] = acc1[rr];
      }
    }
  }
}
// post: the fused kernel left (fh[32], O, a, c) per ray.  d W_of = sum_r (a_r g_r + c_r F_r) fh_r^T with
// F_r = W_of fh_r + b_of O_r, so  d W_of = gt_feat^T [a fh] + W_of M2 + b_of m1^T  and
// d b_of = gt_feat^T [a O] + W_of m1 + b_of s2  with the moments  [M2 m1; . s2] = [c fh | c O]^T [fh | O].
// This kernel writes the two row-scaled copies X1 = [a fh | a O], X2 = [c fh | c O]; two GEMMs do the sums.
__global__ void feat_scale_kernel(long n, const float* rayfeat, float* X1, float* X2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * XCOLS) return;
  const long r = i / XCOLS;
  const int j = (int)(i - r * XCOLS);
  const float* rf = rayfeat + r * RAYFEAT;
  const float v = rf[j];                  // fh[0..31], O
  X1[i] = rf[33] * v;
  X2[i] = rf[34] * v;
}
// post, round 3: the same sums in ONE pass over gt_feat, straight from the fused kernel's per-ray record (fh[32], O, a, c)
// -- no X1 / X2 copies, no split-K GEMM launches, no zero fills.  Workgroup (g, k) takes the g-th chunk of object k's
// rays; wave w owns target-feature block [64 w, 64 w + 64).  v_mfma_f32_16x16x4_f32 with the RAY as contraction index:
// A lane (i, q) streams gt_feat[ray 4 s + q][64 w + 4 i + j] as float4 (element j = the A operand of MFMA j, whose
// output rows are the features 64 w + 4 m + j), B lane (n, q) = a_r [fh | O][16 t + n] for the three column tiles ->
// T = gt_feat^T [a fh | a O].  The moments M = [c fh | c O]^T [fh | O] ride along: the B values of a lane are also its A
// values (same lane index), each wave takes every eighth ray quad, the partial tiles meet in LDS.  Partials
// Tpart [K][G][C][33], Mpart [K][G][33][33] are summed by feat_finish_kernel.  C == 512.
#ifndef FEAT_POST_PF
#define FEAT_POST_PF 4
#endif
#ifndef FEAT_POST_PFF
#define FEAT_POST_PFF 1
#endif
#ifndef FEAT_POST_WPE
#define FEAT_POST_WPE 4          // waves per SIMD the register budget is set for (4 = two 512-thread workgroups per compute unit)
#endif
__global__ __launch_bounds__(512, FEAT_POST_WPE) void feat_post_kernel(int C, int R, const float* __restrict__ gt_feat,
                                                           const float* __restrict__ rayfeat, float* __restrict__ Tpart,
                                                           float* __restrict__ Mpart) {
  // Round 6: the column of O (T[:, 32] = gt_feat^T [a O], M[32][:] = M[:][32] = sum c O fh, M[32][32] = sum c O^2) is
  // formed on the VALU -- it used to be a third 16-column MFMA tile with one live column (4 of 12 MFMAs per ray quad, 5 of the
  // 9 moment MFMAs): a third less matrix-core time, 34 KB of LDS instead of 74 (two workgroups per compute unit).
  __shared__ float s_m[8][4][64][4];                   // per wave: the 2 x 2 fh x fh moment tiles as D fragments
  __shared__ float s_o[8][36];                         // per wave: sum c O fh[0..31], sum c O^2
  const int k = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const long r_lo = (long)R * g / G, r_hi = (long)R * (g + 1) / G;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][2], macc[2][2];
  float t32[4] = {0.f, 0.f, 0.f, 0.f};                 // sum_r a_r O_r g_r[64 w + 4 i + j] over this lane's rays (q)
  float mo0 = 0.f, mo1 = 0.f, moo = 0.f;               // sum c O fh[i], sum c O fh[16 + i], sum c O^2 (this wave's quads)
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = zero;
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_) macc[a_][0] = macc[a_][1] = zero;
  const float* gbase = gt_feat + (long)k * R * C + 64 * w + 4 * i;
  const float* fbase = rayfeat + (long)k * R * RAYFEAT;
  const long nq = (r_hi - r_lo + 3) / 4;               // ray quads of this chunk
  // in flight per lane: FEAT_POST_PF target-feature float4s (HBM) and FEAT_POST_PFF ray records (L2: the fused kernel has
  // just written them) -- 128 registers per wave at two workgroups per compute unit leave room for ~28 of prefetch
  constexpr int PF = FEAT_POST_PF, PFF = FEAT_POST_PFF;
  static_assert(PF % PFF == 0, "the record ring is indexed statically inside the unrolled loop");
  f32x4 abuf[PF];
  float fb[PFF][5];                                    // fh[i], fh[16 + i], O, a, c of this lane's ray
  auto fetch_g = [&](const long sq, f32x4& av) {
    const long r = r_lo + 4 * sq + q;
    av = (sq < nq && r < r_hi) ? *reinterpret_cast<const f32x4*>(gbase + r * C) : zero;
  };
  auto fetch_f = [&](const long sq, float (&f)[5]) {
    const long r = r_lo + 4 * sq + q;
    if (sq < nq && r < r_hi) {
      const float* rf = fbase + r * RAYFEAT;
      f[0] = rf[i]; f[1] = rf[16 + i]; f[2] = rf[32]; f[3] = rf[33]; f[4] = rf[34];
    } else {
      f[0] = f[1] = f[2] = f[3] = f[4] = 0.f;
    }
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) fetch_g(p, abuf[p]);
#pragma unroll
  for (int p = 0; p < PFF; ++p) fetch_f(p, fb[p]);
  for (long s0 = 0; s0 < nq; s0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const long sq = s0 + p;
      const f32x4 av = abuf[p];
      const float f0 = fb[p % PFF][0], f1 = fb[p % PFF][1], fo = fb[p % PFF][2], ar = fb[p % PFF][3], cr = fb[p % PFF][4];
      fetch_g(sq + PF, abuf[p]);
      fetch_f(sq + PFF, fb[p % PFF]);
      if (sq < nq) {
        const float x0 = ar * f0, x1 = ar * f1, xo = ar * fo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], x0, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], x1, acc[j][1], 0, 0, 0);
          t32[j] = fmaf(av[j], xo, t32[j]);
        }
        if ((int)(sq & 7) == w) {                      // (wave-uniform) this wave's share of the moments
          const float y0 = cr * f0, y1 = cr * f1, yo = cr * fo;
          macc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0, f0, macc[0][0], 0, 0, 0);
          macc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0, f1, macc[0][1], 0, 0, 0);
          macc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1, f0, macc[1][0], 0, 0, 0);
          macc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1, f1, macc[1][1], 0, 0, 0);
          mo0 = fmaf(yo, f0, mo0);
          mo1 = fmaf(yo, f1, mo1);
          moo = fmaf(yo, fo, moo);
        }
      }
    }
  }
  // T partial: D lane (n = i, q) register rr of MFMA j = T[c = 64 w + 4 (4 q + rr) + j][16 t + n]
  float* Tp = Tpart + ((long)k * G + g) * C * XCOLS;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float* row = Tp + (long)(64 * w + 4 * (4 * q + rr) + j) * XCOLS;
      row[i] = acc[j][0][rr];
      row[16 + i] = acc[j][1][rr];
    }
  // column 32: the four lanes (q) of a feature group meet
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = t32[j];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (q == 0) Tp[(long)(64 * w + 4 * i + j) * XCOLS + 32] = v;
  }
  // moments: the O row per wave (lanes q of an entry meet), then the eight waves' partial tiles / rows in LDS
  mo0 += __shfl_xor(mo0, 16, 64); mo0 += __shfl_xor(mo0, 32, 64);
  mo1 += __shfl_xor(mo1, 16, 64); mo1 += __shfl_xor(mo1, 32, 64);
  moo += __shfl_xor(moo, 16, 64); moo += __shfl_xor(moo, 32, 64);
  if (q == 0) { s_o[w][i] = mo0; s_o[w][16 + i] = mo1; if (i == 0) s_o[w][32] = moo; }
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s_m[w][2 * a_ + b_][lane][rr] = macc[a_][b_][rr];
  __syncthreads();
  float* Mp = Mpart + ((long)k * G + g) * XCOLS * XCOLS;
  for (int e = tid; e < 4 * 64 * 4; e += 512) {
    const int rr = e & 3, ln = (e >> 2) & 63, tile = e >> 8;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) v += s_m[ww][tile][ln][rr];
    const int m = 16 * (tile >> 1) + 4 * (ln >> 4) + rr, nn = 16 * (tile & 1) + (ln & 15);
    Mp[m * XCOLS + nn] = v;
  }
  if (tid < 33) {                                      // M is symmetric in its O row / column: sum_r c_r O_r fh_r[h]
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) v += s_o[ww][tid];
    Mp[32 * XCOLS + tid] = v;
    if (tid < 32) Mp[tid * XCOLS + 32] = v;
  }
}
// d W_of[c][h] = T[c][h] + sum_j W_of[c][j] M2[j][h] + b_of[c] m1[h];  d b_of[c] = T[c][32] + W_of[c] . m1 + b_of[c] s2
__global__ __launch_bounds__(256) void feat_finish_kernel(const float* params, long p_stride, int off_w, int off_b, int C,
                                                          const float* Tm /* [K][G][C][33] */,
                                                          const float* mom /* [K][G][33][33] */, float* grads, int G) {
  __shared__ float M[XCOLS][XCOLS];
  const int k = blockIdx.y;
  for (int i = threadIdx.x; i < XCOLS * XCOLS; i += 256) {
    float v = 0.f;
    for (int g = 0; g < G; ++g) v += mom[((long)k * G + g) * XCOLS * XCOLS + i];
    M[i / XCOLS][i % XCOLS] = v;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * XCOLS) return;
  const int cc = i / XCOLS, hh = i - cc * XCOLS;
  const float* W = params + (long)k * p_stride + off_w + cc * 32;
  const float bc = params[(long)k * p_stride + off_b + cc];
  float v = 0.f;
  for (int g = 0; g < G; ++g) v += Tm[((long)k * G + g) * C * XCOLS + i];
  if (hh < 32) {
    for (int j = 0; j < 32; ++j) v = fmaf(W[j], M[j][hh], v);              // M2[j][h]  (c fh_j . fh_h)
    grads[(long)k * p_stride + off_w + cc * 32 + hh] = fmaf(bc, M[32][hh], v);   // m1[h] = (c O) . fh_h
  } else {
    for (int j = 0; j < 32; ++j) v = fmaf(W[j], M[32][j], v);
    grads[(long)k * p_stride + off_b + cc] = fmaf(bc, M[32][32], v);      // s2 = (c O) . O
  }
}

int g_num_cu = 0;
int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return g_num_cu;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int eval_launch(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                const float* scale, const float* pts, const float* emb, float* out_alpha, float* out_color,
                float* out_hfeat, float* out_clip, void* stream) {
  EvalDev d;
  d.K = K; d.N = N; d.params = params; d.p_stride = p_stride; d.scale = scale; d.pts = pts; d.emb = emb;
  d.alpha = out_alpha; d.color = out_color; d.hfeat = out_hfeat;
  d.L = make_layout(net->feat_dim);
  const long ntiles = (N + 63) / 64;
  long G = (2L * num_cu()) / K;
  if (G < 1) G = 1;
  if (G > ntiles) G = ntiles;
  d.G = (int)G;
  hipStream_t st = (hipStream_t)stream;
  const bool feat = out_hfeat != nullptr;
  const size_t lds_bytes = (size_t)obj32n::img_floats(feat) * 4;
  // the image with the feature layer exceeds the 64 KB default
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)eval_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              obj32n::img_floats(true) * 4);
    (void)hipFuncSetAttribute((const void*)eval_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              obj32n::img_floats(true) * 4);
  });
  const dim3 grid(K * d.G), blk(256);
  if (emb) {
    if (feat) hipLaunchKernelGGL((eval_kernel<true, true>), grid, blk, lds_bytes, st, d);
    else hipLaunchKernelGGL((eval_kernel<false, true>), grid, blk, lds_bytes, st, d);
  } else {
    if (feat) hipLaunchKernelGGL((eval_kernel<true, false>), grid, blk, lds_bytes, st, d);
    else hipLaunchKernelGGL((eval_kernel<false, false>), grid, blk, lds_bytes, st, d);
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  if (out_clip) return objnerf_feature_head(net, K, N, params, p_stride, out_hfeat, nullptr, out_clip, stream);
  return OBJNERF_OK;
}

}  // namespace

extern "C" {

int objnerf_abi_version(void) { return OBJNERF_ABI_VERSION; }

int64_t objnerf_param_layout(const objnerf_net* net, int64_t offsets[OBJNERF_N_TENSORS + 1]) {
  if (!net || !offsets || net->hidden <= 0 || net->feat_dim <= 0) return OBJNERF_EINVAL;
  const int64_t H_ = net->hidden, C = net->feat_dim;
  const int64_t sizes[OBJNERF_N_TENSORS] = {H_ * OBJ_E1, H_, H_ * H_, H_, H_ * (H_ + OBJ_E1), H_, H_ * H_, H_, H_, 1,
                                            H_ * (H_ + OBJ_E2), H_, 3 * H_, 3, H_ * (H_ + OBJ_E2), H_, C * H_, C,
                                            OBJ_NDIR * 3};
  int64_t o = 0;
  for (int i = 0; i < OBJNERF_N_TENSORS; ++i) { offsets[i] = o; o += sizes[i]; }
  offsets[OBJNERF_N_TENSORS] = o;
  return (o + 63) / 64 * 64;
}

// Workgroups per object.  One workgroup per compute unit at a time (LDS), each sweeps NT / G tiles of its object and
// pays a fixed cost (weight staging, slab write + reduction) of about TILE_EQUIV tiles.  G = #CU / K fills the chip
// in one round when K divides it well (K = 50 -> 5 x 50 = 250 of 256); otherwise (K = 100, 130, 300 ...) more,
// shorter workgroups in several rounds come closer to K * NT / #CU tiles per compute unit.
static int grid_cap(int K) {
  const int g = num_cu() / (K > 0 ? K : 1);
  return g > 32 ? g : 32;
}
static int train_grid(int K, int NT) {
  constexpr long TILE_EQUIV = 4;
  const int cu = num_cu();
  int gmax = grid_cap(K);
  if (gmax > NT) gmax = NT;
  if (gmax < 1) gmax = 1;
  int best = 1;
  long best_cost = -1;
  for (int G = 1; G <= gmax; ++G) {
    const long rounds = ((long)K * G + cu - 1) / cu;
    const long cost = rounds * ((NT + G - 1) / G + TILE_EQUIV);
    if (best_cost < 0 || cost < best_cost) { best = G; best_cost = cost; }
  }
  return best;
}

// Flat mode of the second-generation fp32 kernel (objnerf_train_common.h): one workgroup per CU, equal shares of the flat
// (object, tile) space.  Taken when its longest workgroup (tiles + a fixed cost per segment) beats the strided grid's and
// an object's partial slabs fit the workspace (grid_cap(K) slots).
// tile_equiv: the fixed cost of a segment (LDS clear, weight staging, slab write + reduction) in tiles: ~4 for the fp32
// kernel (22 us per tile).  The bf16 kernel stays on the strided grid: its segment costs ~10 of its 7-us tiles, and
// wrapping its body in the segment loop alone cost 3.5 % (measured: 2.74 -> 2.83 ms strided, 2.79 flat).
static void choose_flat(TrainDev& d, int K, const long TILE_EQUIV) {
  const int cu = num_cu();
  const long T = (long)K * d.NT;
  if (T < cu) return;
  const long share = (T + cu - 1) / cu;
  const long segs = 1 + (share + d.NT - 1) / d.NT;                    // segments a share can be cut into
  const long cost_flat = share + segs * TILE_EQUIV;
  const long rounds = ((long)K * d.G + cu - 1) / cu;
  const long cost_strided = rounds * ((d.NT + d.G - 1) / d.G + TILE_EQUIV);
  const long gs = ((long)d.NT * cu + T - 1) / T + 1;                  // workgroups that can touch one object
  if (cost_flat < cost_strided && gs <= grid_cap(K)) { d.flat_nwg = cu; d.Gs = (int)gs; }
}

size_t objnerf_train_workspace_bytes(const objnerf_net* net, int32_t K, int32_t R, int32_t S, int32_t with_feat) {
  if (!net || K <= 0 || R <= 0 || S <= 0) return 0;
  // bit 1 of with_feat (value 2): size for the layer-wise path (OBJNERF_TRAIN_LAYERWISE)
  if (net->hidden != 32 || S > 64 || (with_feat & 2)) return objgen::train_workspace_bytes(net, K, R, S, with_feat & 1, (with_feat & 4) != 0);
  with_feat &= 1;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  const int64_t ps = objnerf_param_layout(net, offs);
  const int Gmax = grid_cap(K);   // upper bound on the workgroups per object
  size_t n = align256((size_t)K * Gmax * ps * 4) + align256((size_t)K * Gmax * 4 * 4) + align256((size_t)ps) + 256;
  if (with_feat)
    n += align256((size_t)K * R * RAYIN * 4) + align256((size_t)K * GRAM * 4) + align256((size_t)K * R * RAYFEAT * 4) +
         2 * align256((size_t)K * R * XCOLS * 4) +
         align256(((size_t)K * net->feat_dim * XCOLS + (size_t)K * XCOLS * XCOLS) * 4) +
         align256(objgen::wgrad_parts_floats(K, net->feat_dim, XCOLS, R) * 4) +
         align256(objgen::wgrad_parts_floats(K, XCOLS, XCOLS, R) * 4);
  return n;
}

int objnerf_train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream) {
  (void)hipGetLastError();   // drop stale non-sticky errors of other HIP users of this thread
  if (!net || !a || !a->params || !a->scale || !a->z || !a->gt_depth || !a->gt_rgb || !a->labels || !a->counts ||
      !a->flags || !a->grads || !a->loss_terms || !a->status || !a->workspace)
    return OBJNERF_EINVAL;
  if (!a->pts && (!a->origins || !a->dirs)) return OBJNERF_EINVAL;
  if (a->K <= 0 || a->R <= 0 || a->S <= 0) return OBJNERF_EINVAL;
  if (net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if ((a->mode & OBJNERF_TRAIN_FP16) && (a->mode & OBJNERF_TRAIN_BF16)) return OBJNERF_EINVAL;
  if (a->optim && (!a->optim->exp_avg || !a->optim->exp_avg_sq || !a->optim->group_steps ||
                   (a->optim->bank != 0 && a->optim->bank != 1)))
    return OBJNERF_EINVAL;
  if (net->hidden != 32 || a->S > 64 || (a->mode & (OBJNERF_TRAIN_LAYERWISE | OBJNERF_TRAIN_FP16))) {
    // wider networks (background: hidden 128) and long rays: layer-wise path, activations in the workspace
    if (a->workspace_bytes < objgen::train_workspace_bytes(net, a->K, a->R, a->S, a->gt_feat != nullptr,
                                                           (a->mode & (OBJNERF_TRAIN_FP16 | OBJNERF_TRAIN_BF16)) != 0))
      return OBJNERF_EINVAL;
    if (a->emb_debug) return OBJNERF_ENOTSUP;
    // (objgen::train_step takes OBJNERF_TRAIN_SELF_COUNTS and the optimiser itself where its one-launch kernels run
    // -- it reports what it has done through `done` -- and leaves them to the launches below otherwise)
    int done = 0;
    int rc;
#ifndef OBJ_NO_TRAIN256
    if (!(a->mode & OBJNERF_TRAIN_LAYERWISE) && obj256::applicable(net, a)) {
      if (a->mode & OBJNERF_TRAIN_SELF_COUNTS) {
        rc = objnerf_label_counts(a->K, a->R, a->labels, const_cast<int32_t*>(a->counts), const_cast<int32_t*>(a->flags), stream);
        if (rc) return rc;
      }
      rc = obj256::train_step(net, a, stream);
    } else
#endif
    rc = objgen::train_step(net, a, stream, &done);
    if (rc) return rc;
    if (a->optim && !(done & 2)) {
      int64_t offs[OBJNERF_N_TENSORS + 1];
      objnerf_param_layout(net, offs);
      const bool feat_ = a->gt_feat != nullptr;
      const objnerf_adamw_args* o = a->optim;
      return objmisc::adamw_flags_range(a->K, offs[OBJNERF_N_TENSORS], a->p_stride, const_cast<float*>(a->params), a->grads,
                                        o->exp_avg, o->exp_avg_sq, nullptr, a->flags, o->group_steps, o->bank,
                                        offs[OBJNERF_T_CL_W], offs[OBJNERF_T_FL_W], offs[OBJNERF_T_PE_B],
                                        feat_ ? offs[OBJNERF_N_TENSORS] : offs[OBJNERF_T_FL_W],
                                        feat_ ? offs[OBJNERF_N_TENSORS] : offs[OBJNERF_T_PE_B], o->lr, o->beta1, o->beta2,
                                        o->eps, o->weight_decay, stream);
    }
    return OBJNERF_OK;
  }
  const bool self_counts = (a->mode & OBJNERF_TRAIN_SELF_COUNTS) != 0;
  const bool feat = a->gt_feat != nullptr;
  // (with the feature loss feat_pre_kernel's first workgroup of every object counts the labels: no launch of its own)
  const bool counts_in_pre = self_counts && feat && net->feat_dim <= FEAT_PRE_MAXC && net->feat_dim % 16 == 0;
  if (self_counts && !counts_in_pre) {
    // per-object counts only (one workgroup per object, no zero fill, no atomics): the fused kernel's workgroups derive
    // the cross-object flags from them and finalize_kernel publishes the pair
    const int rc = objmisc::label_counts_only(a->K, a->R, a->labels, const_cast<int32_t*>(a->counts), stream);
    if (rc) return rc;
  }
  if (feat && (TS / a->S) > 16) return OBJNERF_ENOTSUP;
  const bool bf16 = (a->mode & OBJNERF_TRAIN_BF16) != 0;
  if (bf16 && (a->relu_masks || a->emb_debug)) return OBJNERF_ENOTSUP;
  if (a->workspace_bytes < objnerf_train_workspace_bytes(net, a->K, a->R, a->S, a->gt_feat != nullptr))
    return OBJNERF_EINVAL;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  const int64_t ps = objnerf_param_layout(net, offs);
  if (a->p_stride < offs[OBJNERF_N_TENSORS]) return OBJNERF_EINVAL;

  TrainDev d;
  d.K = a->K; d.R = a->R; d.S = a->S;
  d.TR = TS / a->S;
  d.NT = (a->R + d.TR - 1) / d.TR;
  d.G = train_grid(a->K, d.NT);
  d.flat_nwg = 0; d.Gs = 0;
  d.color_scaling = a->color_scaling; d.opacity_scaling = a->opacity_scaling; d.feat_scaling = a->feat_scaling;
  d.obj_center = a->obj_center;
  d.params = a->params; d.p_stride = a->p_stride; d.scale = a->scale;
  d.pts = a->pts; d.origins = a->origins; d.dirs = a->dirs; d.z = a->z;
  d.gt_depth = a->gt_depth; d.gt_rgb = a->gt_rgb; d.labels = a->labels; d.gt_feat = a->gt_feat;
  d.counts = a->counts; d.flags = a->flags; d.derive_flags = self_counts ? 1 : 0;
  d.L = make_layout(net->feat_dim);
  char* ws = (char*)a->workspace;
  const int Gmax = grid_cap(a->K);
  d.slab = (float*)ws;
  d.slab_stride = ps;
  ws += align256((size_t)a->K * Gmax * ps * 4);
  d.loss_part = (float*)ws;
  ws += align256((size_t)a->K * Gmax * 4 * 4);
  uint8_t* has_grad = (uint8_t*)ws;
  ws += align256((size_t)ps) + 256;
  float* rayin = nullptr; float* gram = nullptr; float* rayfeat = nullptr; float* head_snap = nullptr;
  float *X1 = nullptr, *X2 = nullptr, *Tm = nullptr, *mom = nullptr, *parts_t = nullptr, *parts_m = nullptr;
  const int C = net->feat_dim;
  if (feat) {
    rayin = (float*)ws;   ws += align256((size_t)a->K * a->R * RAYIN * 4);
    gram = (float*)ws;    ws += align256((size_t)a->K * GRAM * 4);
    rayfeat = (float*)ws; ws += align256((size_t)a->K * a->R * RAYFEAT * 4);
    X1 = (float*)ws;      ws += align256((size_t)a->K * a->R * XCOLS * 4);
    X2 = (float*)ws;      ws += align256((size_t)a->K * a->R * XCOLS * 4);
    Tm = (float*)ws;
    mom = Tm + (size_t)a->K * C * XCOLS;
    ws += align256(((size_t)a->K * C * XCOLS + (size_t)a->K * XCOLS * XCOLS) * 4);
    parts_t = (float*)ws; ws += align256(objgen::wgrad_parts_floats(a->K, C, XCOLS, a->R) * 4);
    parts_m = (float*)ws;
    // (the one-pass route does not use the split-K parts: the copy of [W_of | b_of] for finalize_kernel lives in their room,
    // K C 33 floats of the K ceil(R / ..) C 33 + 64 that wgrad_parts_floats reserves)
    head_snap = parts_t;
  }
  d.rayin = rayin; d.gram = gram; d.rayfeat = rayfeat;
  d.relu_masks = a->relu_masks;
  d.emb_debug = a->emb_debug;

  hipStream_t st = (hipStream_t)stream;
  // (no byte mask and no zero fills: finalize_kernel takes the ranges without a gradient as arguments -- without
  // gt_feat the whole feature branch has none (train.py:435-438 -> .grad stays None); with it, the 512-d head's
  // gradient is produced by feat_finish_kernel instead of the slabs -- and writes the status word itself)
  (void)has_grad;
  const size_t lds_bytes = (size_t)((feat ? W_FLOATS_FEAT : W_FLOATS_NOFEAT) + SM_FLOATS + STG_ROWS * STG_LD) * 4;
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)train_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((W_FLOATS_FEAT + SM_FLOATS + STG_ROWS * STG_LD) * 4));
    (void)hipFuncSetAttribute((const void*)train_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((W_FLOATS_FEAT + SM_FLOATS + STG_ROWS * STG_LD) * 4));
  });
  const float *fin_T = nullptr, *fin_M = nullptr;
  int fin_G = 0;
  if (feat) {
    const bool pre_one = C <= FEAT_PRE_MAXC && C % 16 == 0;
    if (!pre_one) objgen::feat_gram(stream, a->K, a->params, (long)a->p_stride, d.L.of_w, d.L.of_b, C, 32, gram, GRAM);
    // u = gt_feat W_of  ([R x C] [C x 32] per object) on the batched MFMA GEMM; beta, |g| beside it
    if (pre_one) {
      // u, beta, |g| in one pass over gt_feat (feat_pre_kernel)
      const size_t pre_lds = ((size_t)(C / 16) * 2 * 64 * 4 + C) * sizeof(float);
      objnerf_once_per_device([] {
        (void)hipFuncSetAttribute((const void*)feat_pre_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(((FEAT_PRE_MAXC / 16) * 2 * 64 * 4 + FEAT_PRE_MAXC) * sizeof(float)));
      });
      int gpo = 2 * num_cu() / a->K;              // workgroups per object: one round of the chip at two per compute unit
      if (gpo < 1) gpo = 1;
      if (gpo > (a->R + 127) / 128) gpo = (a->R + 127) / 128;
      hipLaunchKernelGGL(feat_pre_kernel, dim3(gpo, a->K), dim3(512), pre_lds, st, a->params, (long)a->p_stride,
                         d.L.of_w, d.L.of_b, C, a->R, a->gt_feat, rayin, gram, head_snap, a->labels,
                         counts_in_pre ? const_cast<int32_t*>(a->counts) : nullptr);
    } else {
      objgen::gemm_f32(stream, a->K, a->R, 32, C, a->gt_feat, C, 1, (long)a->R * C, a->params + d.L.of_w, 32, 1,
                       (long)a->p_stride, rayin, RAYIN, 1, (long)a->R * RAYIN, false);
      hipLaunchKernelGGL(feat_rowstats_kernel, dim3((a->R + 15) / 16, a->K), dim3(256), 0, st, a->params,
                         (long)a->p_stride, d.L.of_b, C, a->R, a->gt_feat, rayin);
    }
    if (bf16) launch_train_bf16(d, stream, true);
#ifdef OBJ_FEAT_GEN1      // diagnostic builds: the first-generation feature kernel (tools/build_variant.sh)
    else if (d.relu_masks) hipLaunchKernelGGL((train_fused_kernel<true, true>), dim3(a->K * d.G), dim3(NTHR), lds_bytes, st, d);
    else hipLaunchKernelGGL((train_fused_kernel<true, false>), dim3(a->K * d.G), dim3(NTHR), lds_bytes, st, d);
#else
    else { choose_flat(d, a->K, 4); launch_train32(d, stream, true); }
    (void)lds_bytes;
#endif
    if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
    // 512-d head gradient from the per-ray (fh, O, a, c): two split-K GEMMs over the rays + a small finish
    const long nr = (long)a->K * a->R;
    int gpo = 2 * num_cu() / a->K;                // ray chunks per object: one round of the chip at two workgroups per compute unit
    if (gpo < 1) gpo = 1;
    if (gpo > 16) gpo = 16;
    // the partials live in the room of X1 AND X2 (adjacent in the workspace, both unused on this route): 2 K R XCOLS floats.
    // (Round 4 used X1's room only and at most 8 chunks: a rank's share of configs[3] -- 15 objects -- then ran 105
    // workgroups over 126 MB of target features, 1.1 TB/s.)
    const size_t part_room = ((size_t)((char*)X2 - (char*)X1) / 4 + (size_t)a->K * a->R * XCOLS) / (size_t)a->K;
    while (gpo > 1 && (size_t)gpo * ((size_t)C * XCOLS + XCOLS * XCOLS) > part_room) --gpo;
    const bool one_pass = C == 512 && (size_t)gpo * ((size_t)C * XCOLS + XCOLS * XCOLS) <= part_room;
    int Gfin = 1;
    const float *Tsrc = Tm, *Msrc = mom;
    if (one_pass) {
      float* Tpart = X1;
      float* Mpart = X1 + (size_t)a->K * gpo * C * XCOLS;
      hipLaunchKernelGGL(feat_post_kernel, dim3(gpo, a->K), dim3(512), 0, st, C, a->R, a->gt_feat, rayfeat, Tpart, Mpart);
      Gfin = gpo; Tsrc = Tpart; Msrc = Mpart;
    } else {
      hipLaunchKernelGGL(feat_scale_kernel, dim3((unsigned)((nr * XCOLS + 255) / 256)), dim3(256), 0, st, nr, rayfeat, X1, X2);
      (void)hipMemsetAsync(Tm, 0, ((size_t)a->K * C * XCOLS + (size_t)a->K * XCOLS * XCOLS) * 4, st);
      objgen::wgrad_f32(stream, a->K, C, XCOLS, a->R, a->gt_feat, 1, C, (long)a->R * C, X1, XCOLS, 1, (long)a->R * XCOLS, Tm,
                        XCOLS, (long)C * XCOLS, parts_t, objgen::wgrad_parts_floats(a->K, C, XCOLS, a->R));
      objgen::wgrad_f32(stream, a->K, XCOLS, XCOLS, a->R, X2, 1, XCOLS, (long)a->R * XCOLS, rayfeat, RAYFEAT, 1,
                        (long)a->R * RAYFEAT, mom, XCOLS, (long)XCOLS * XCOLS, parts_m,
                        objgen::wgrad_parts_floats(a->K, XCOLS, XCOLS, a->R));
    }
    if (one_pass && pre_one) {      // finalize_kernel forms the head's gradient itself (round 6: one launch fewer)
      fin_T = Tsrc; fin_M = Msrc; fin_G = Gfin;
    } else {
      hipLaunchKernelGGL(feat_finish_kernel, dim3((C * XCOLS + 255) / 256, a->K), dim3(256), 0, st, a->params,
                         (long)a->p_stride, d.L.of_w, d.L.of_b, C, Tsrc, Msrc, a->grads, Gfin);
    }
  } else if (bf16) {
    launch_train_bf16(d, stream, false);
  } else {
    choose_flat(d, a->K, 4);
    launch_train32(d, stream, false);
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  const long P = offs[OBJNERF_N_TENSORS];
  dim3 fg((unsigned)((P + 256 * FIN_EPT - 1) / (256 * FIN_EPT)), (unsigned)a->K);
  FinalizeArgs fa;
  fa.slab = d.slab; fa.loss_part = d.loss_part; fa.K = a->K; fa.G = d.G; fa.P = P; fa.slab_stride = (long)ps;
  fa.p_stride = (long)a->p_stride;
  fa.ng_lo = feat ? P : d.L.fl_w; fa.ng_hi = feat ? P : d.L.pe_b;
  fa.ext_lo = feat ? d.L.of_w : P; fa.ext_hi = feat ? d.L.pe_b : P;
  fa.grads = a->grads; fa.loss_terms = a->loss_terms; fa.status = a->status;
  fa.flat_nwg = d.flat_nwg; fa.NT = d.NT; fa.Gs = d.Gs;
  fa.params = nullptr; fa.m = fa.v = nullptr; fa.flags = a->flags; fa.steps = nullptr; fa.bank = 0;
  fa.lo1 = offs[OBJNERF_T_CL_W]; fa.lo2 = offs[OBJNERF_T_FL_W]; fa.hi2 = offs[OBJNERF_T_PE_B];
  fa.lr = fa.b1 = fa.b2 = fa.wd = 0.0; fa.eps = 0.f;
  fa.counts_in = self_counts ? a->counts : nullptr; fa.flags_out = const_cast<int*>(a->flags);
  fa.Tpart = fin_T; fa.Mpart = fin_M; fa.head_w = head_snap; fa.Gfin = fin_G; fa.C = C; fa.of_w = d.L.of_w; fa.of_b = d.L.of_b;
  if (a->optim) {
    const objnerf_adamw_args* o = a->optim;
    fa.params = const_cast<float*>(a->params); fa.m = o->exp_avg; fa.v = o->exp_avg_sq; fa.steps = o->group_steps;
    fa.bank = o->bank; fa.lr = (double)o->lr; fa.b1 = (double)o->beta1; fa.b2 = (double)o->beta2;
    fa.wd = (double)o->weight_decay; fa.eps = o->eps;
  }
  hipLaunchKernelGGL(finalize_kernel, fg, dim3(256), 0, st, fa);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

#ifdef PHASE_TIMING
extern "C" int objnerf_debug_phase(unsigned long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return OBJNERF_ELAUNCH;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 8 * 24) == hipSuccess
             ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
#endif

int objnerf_eval_points(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                        const float* scale, const float* pts, float* out_alpha, float* out_color, float* out_hfeat,
                        float* out_clip, void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !scale || !pts || !out_alpha || !out_color || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  if (net->hidden != 32 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if (out_clip && !out_hfeat) return OBJNERF_EINVAL;   // the head runs on the H-wide hidden
  return eval_launch(net, K, N, params, p_stride, scale, pts, nullptr, out_alpha, out_color, out_hfeat, out_clip,
                     stream);
}

size_t objnerf_eval_workspace_bytes(const objnerf_net* net, int32_t K, int64_t N) {
  if (!net || K <= 0 || N <= 0 || net->hidden == 32) return 0;
  return objgen::eval_workspace_bytes(net, K, (long)N);
}

int objnerf_eval_points_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                           const float* scale, const float* pts, float* out_alpha, float* out_color, float* out_hfeat,
                           float* out_clip, void* workspace, size_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !scale || !pts || !out_alpha || !out_color || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  if (net->hidden == 32)
    return objnerf_eval_points(net, K, N, params, p_stride, scale, pts, out_alpha, out_color, out_hfeat, out_clip, stream);
  return objgen::eval_points(net, K, (long)N, params, (long)p_stride, scale, pts, out_alpha, out_color, out_hfeat,
                             out_clip, workspace, workspace_bytes, stream);
}

int objnerf_mlp_forward(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                        const float* emb, float* out_alpha, float* out_color, float* out_hfeat, float* out_clip,
                        void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !emb || !out_alpha || !out_color || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  if (net->hidden != 32 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if (out_clip && !out_hfeat) return OBJNERF_EINVAL;
  return eval_launch(net, K, N, params, p_stride, nullptr, nullptr, emb, out_alpha, out_color, out_hfeat, out_clip,
                     stream);
}

int objnerf_mlp_forward_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                           const float* emb, float* out_alpha, float* out_color, float* out_hfeat, float* out_clip,
                           void* workspace, size_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !emb || !out_alpha || !out_color || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  if (net->hidden == 32)
    return objnerf_mlp_forward(net, K, N, params, p_stride, emb, out_alpha, out_color, out_hfeat, out_clip, stream);
  return objgen::eval_points(net, K, (long)N, params, (long)p_stride, nullptr, nullptr, out_alpha, out_color, out_hfeat,
                             out_clip, workspace, workspace_bytes, stream, emb);
}

size_t objnerf_mlp_backward_workspace_bytes(const objnerf_net* net, int32_t K, int64_t N, int32_t with_clip) {
  if (!net || K <= 0 || N <= 0) return 0;
  return objgen::mlp_backward_workspace_bytes(net, K, (long)N, with_clip);
}

int objnerf_mlp_backward_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                            const float* emb, const float* d_alpha, const float* d_color, const float* d_clip,
                            float* grads, float* d_emb, void* workspace, size_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !emb || !d_alpha || !d_color || !grads || !d_emb || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  return objgen::mlp_backward(net, K, (long)N, params, (long)p_stride, emb, d_alpha, d_color, d_clip, grads, d_emb, workspace,
                              workspace_bytes, stream);
}

int objnerf_embed_bwd(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                      const float* scale, const float* pts, const float* d_emb, float* d_B, float* scratch, void* stream) {
  (void)hipGetLastError();
  if (!net || !params || !scale || !pts || !d_emb || !d_B || !scratch || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  return objgen::embed_backward(net, K, (long)N, params, (long)p_stride, scale, pts, d_emb, d_B, scratch, stream);
}

}  // extern "C"
