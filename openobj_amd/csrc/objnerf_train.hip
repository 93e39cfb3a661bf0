// Fused training iteration + point evaluation for K stacked hidden-32 object networks (gfx950).
//
// Replaces, for cfg.training_strategy == "hip", the reference's per-iteration op sequence
//   vmap(pe_model) -> vmap(fc_model) -> loss.step_batch_loss -> backward          (train.py:424-472)
// with ONE kernel: each 256-thread workgroup owns one object's weights in LDS and sweeps that
// object's rays in tiles of 128 samples (whole rays).  Per tile:
//   1. every wave embeds 32 samples and runs the MLP forward chain on MFMA, activations in registers;
//   2. sigma / rgb go to LDS, a wave per ray-group composites (segmented wave scans), evaluates the
//      masked losses of loss.py and writes d(loss)/d(sigma,rgb) back to LDS;
//   3. every wave back-propagates its 32 samples in registers (dgrad), transposes (d_out, input)
//      pairs through LDS and the four waves share the weight-gradient MFMAs (one 32x32 tile each per
//      layer group), accumulating in registers across the whole sweep.
// Each workgroup then writes one partial-gradient slab; objnerf_finalize sums the slabs (no atomics).
#include "objnerf_mlp.h"
#include "../../include/objnerf_hip.h"

using namespace obj32;

namespace {

constexpr int TS = 128;            // samples per workgroup tile
constexpr int STG_LD = TS + 1;     // staging row stride (floats), odd -> conflict-free columns
constexpr int STG_ROWS = 192;
constexpr int SM_FLOATS = 8 * TS;  // s_alpha, s_col[3], s_da, s_dc[3]
constexpr int NRED = 6 * 32 + 4 + 63 + 4;

struct TrainDev {
  int K, R, S, G, TR, NT;
  float color_scaling, opacity_scaling, feat_scaling, obj_center;
  const float* params; long p_stride; const float* scale;
  const float* pts; const float* origins; const float* dirs; const float* z;
  const float* gt_depth; const float* gt_rgb; const uint8_t* labels; const float* gt_feat;
  const int* counts; const int* flags;
  float* slab;        // [K][G][slab_stride]
  long slab_stride;
  float* loss_part;   // [K][G][4]
  Layout L;
};

struct EvalDev {
  int K, G; long N;
  const float* params; long p_stride; const float* scale; const float* pts;
  float* alpha; float* color; float* hfeat;
  Layout L;
};

__device__ __forceinline__ void store_tile_T(float* stg_lane, const int rowbase, const f32x16& v, const int n = 16) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (r < n) stg_lane[(rowbase + row0(r)) * STG_LD] = v[r];
}

// D[out][in] += sum_s dT[out][s] * aT[in][s] over the 128 staged samples
__device__ __forceinline__ void wgrad_tile(f32x16& acc, const float* dT, const float* aT) {
#pragma unroll 16
  for (int t = 0; t < TS / 2; ++t)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dT[2 * t], aT[2 * t], acc, 0, 0, 0);
}

__device__ __forceinline__ void write_tile(float* slab, const f32x16& acc, const int c, const int kh, const int ct,
                                           const int w_off, const int ncols, const int b_off) {
  const int col = 32 * ct + c;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int out = row0(r) + 4 * kh;
    if (col < ncols) slab[w_off + out * ncols + col] = acc[r];
    else if (col == ncols && b_off >= 0) slab[b_off + out] = acc[r];
  }
}

// ------------------------------------------------------------------------------------------------
template <bool FEAT>
__global__ __launch_bounds__(256, 1) void train_fused_kernel(const TrainDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, kh = lane >> 5;
  const int k = blockIdx.x / a.G, g = blockIdx.x % a.G;
  constexpr int WF = FEAT ? W_FLOATS_FEAT : W_FLOATS_NOFEAT;
  float* s_alpha = lds + WF;
  float* s_col = s_alpha + TS;
  float* s_da = s_col + 3 * TS;
  float* s_dc = s_da + TS;
  float* stg = lds + WF + SM_FLOATS;

  stage_weights(lds, a.params + (long)k * a.p_stride, a.L, FEAT, tid, 256);
  for (int i = tid; i < STG_ROWS * STG_LD; i += 256) stg[i] = 0.0f;
  __syncthreads();

  const float scale = a.scale[k];
  const int S = a.S, R = a.R, TR = a.TR;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  const float inv1 = a.flags[0] ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = a.flags[1] ? 0.0f : 1.0f / (n2 + 1e-10f);

  // persistent gradient accumulators
  f32x16 accA = zero16(), accB = zero16(), accC = zero16();
  // row-wise sums over samples: one register each, slot r (lane 16+r of a half) = feature row0(r)+4kh
  float g_bm1 = 0.f, g_bm2 = 0.f, g_wa = 0.f, g_woc0 = 0.f, g_woc1 = 0.f, g_woc2 = 0.f;
  float g_ba = 0.f, g_boc0 = 0.f, g_boc1 = 0.f, g_boc2 = 0.f;
  float g_B0 = 0.f, g_B1 = 0.f, g_B2 = 0.f;   // slot i of g_Bn = entry 16n+i of this half's 33 (dir, xyz) sums
  float l_d = 0.f, l_c = 0.f, l_o = 0.f;

  float* stg_lane = stg + (4 * kh) * STG_LD + 32 * w + c;
  const float* lane_rd = stg + c * STG_LD + kh;   // + row0*STG_LD + 2t

  const float* wt_in = lds + OFF_IN + (4 * kh) * ST_IN + c;
  const float* wt_m1 = lds + OFF_M1 + (4 * kh) * ST_M + c;
  const float* wt_cat = lds + OFF_CAT + (4 * kh) * ST_CAT + c;
  const float* wt_m2 = lds + OFF_M2 + (4 * kh) * ST_M + c;
  const float* wt_cl = lds + OFF_CL + (4 * kh) * ST_CL + c;

  for (int tile = g; tile < a.NT; tile += a.G) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    const int ray0 = tile * TR;
    // ---------------------------------------------------------------- 1. forward
    const int slot = 32 * w + c;
    const int q = slot / S, si = slot - q * S;
    const int ray = ray0 + q;
    const bool valid = (q < TR) && (ray < R);
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
      const long rr = (long)k * R + ray;
      if (a.pts) {
        const float* p = a.pts + (rr * S + si) * 3;
        px = p[0]; py = p[1]; pz = p[2];
      } else {
        const float zz = a.z[rr * S + si];
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        px = __fadd_rn(o[0], __fmul_rn(d[0], zz)) - a.obj_center;   // vmap.py:548-551
        py = __fadd_rn(o[1], __fmul_rn(d[1], zz)) - a.obj_center;
        pz = __fadd_rn(o[2], __fmul_rn(d[2], zz)) - a.obj_center;
      }
    }
    float t[3], proj[OBJ_NDIR];
    project(lds, px, py, pz, scale, t, proj);
    Emb e;
    embed(e, kh, t, proj);
    Acts act;
    Heads hd;
    mlp_forward<FEAT>(lds, c, kh, e, act, hd);
    if (kh == 0) {
      s_alpha[slot] = hd.alpha;
      s_col[slot] = hd.col[0];
      s_col[TS + slot] = hd.col[1];
      s_col[2 * TS + slot] = hd.col[2];
    }
    __syncthreads();
    // ---------------------------------------------------------------- 2. composite + loss (loss.py:27-101)
    {
      const int rpp = 64 / S;                       // rays per wave pass
      const int npass = (TR + rpp - 1) / rpp;
      for (int ps = w; ps < npass; ps += 4) {
        const int ql = lane / S, pos = lane - ql * S;
        const int qq = ps * rpp + ql;
        const int rayq = ray0 + qq;
        const bool on = (ql < rpp) && (qq < TR) && (rayq < R);
        const int sl = qq * S + pos;
        float al = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, zz = 0.f;
        float gtd = 0.f, gr = 0.f, gg = 0.f, gb = 0.f;
        int lab = 2;
        if (on) {
          const long rr = (long)k * R + rayq;
          al = s_alpha[sl]; c0 = s_col[sl]; c1 = s_col[TS + sl]; c2 = s_col[2 * TS + sl];
          zz = a.z[rr * S + pos];
          gtd = a.gt_depth[rr];
          gr = a.gt_rgb[rr * 3]; gg = a.gt_rgb[rr * 3 + 1]; gb = a.gt_rgb[rr * 3 + 2];
          lab = a.labels[rr];
        }
        const float occ = on ? sigmoid_acc(al) : 0.0f;               // render_rays.py:13
        const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;          // render_rays.py:38
        const float Pinc = seg_scan_mul(fr, pos, S);
        float T = __shfl_up(Pinc, 1, 64);
        if (pos == 0) T = 1.0f;
        const float wgt = occ * T;                                   // render_rays.py:43
        const int last = lane - pos + S - 1;
        const float D = __shfl(seg_scan_add(wgt * zz, pos, S), last, 64);       // loss.py:31
        const float O = __shfl(seg_scan_add(wgt, pos, S), last, 64);            // loss.py:35
        const float C0 = __shfl(seg_scan_add(wgt * c0, pos, S), last, 64);      // loss.py:34
        const float C1 = __shfl(seg_scan_add(wgt * c1, pos, S), last, 64);
        const float C2 = __shfl(seg_scan_add(wgt * c2, pos, S), last, 64);
        const float dz = zz - D;
        const float V = __shfl(seg_scan_add(wgt * (dz * dz), pos, S), last, 64);  // loss.py:32-33
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
        const float suf = seg_rscan_add(qv, pos, S) - qv;            // sum_{j>i} dL/dw_j * w_j
        const float docc = dw * T - suf / fr;
        if (on) {
          s_da[sl] = 10.0f * (docc * occ * (1.0f - occ));            // d / d raw alpha (model.py:88)
          s_dc[sl] = gC0 * wgt * c0 * (1.0f - c0);                   // d / d raw colour (pre-sigmoid)
          s_dc[TS + sl] = gC1 * wgt * c1 * (1.0f - c1);
          s_dc[2 * TS + sl] = gC2 * wgt * c2 * (1.0f - c2);
        }
      }
    }
    __syncthreads();
    // ---------------------------------------------------------------- 3. backward
    const float da = valid ? s_da[slot] : 0.0f;
    const float dc0 = valid ? s_dc[slot] : 0.0f;
    const float dc1 = valid ? s_dc[TS + slot] : 0.0f;
    const float dc2 = valid ? s_dc[2 * TS + slot] : 0.0f;
    if (kh == 0) { g_ba += da; g_boc0 += dc0; g_boc1 += dc1; g_boc2 += dc2; }
    // The sincos of the embedding is RE-computed below: hide proj from CSE, otherwise the compiler keeps
    // every forward cos value live across the whole backward pass.
#pragma unroll
    for (int j = 0; j < OBJ_NDIR; ++j) asm volatile("" : "+v"(proj[j]));
    float dproj[OBJ_NDIR];
#pragma unroll
    for (int j = 0; j < OBJ_NDIR; ++j) dproj[j] = 0.f;

    // ---- phase A: heads, colour layer, mid2
    f32x16 d_hc, d_h4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0(r) + 4 * kh;
      const float hv = act.hc[r];
      slot_accum(g_woc0, dc0 * hv, r, c);
      slot_accum(g_woc1, dc1 * hv, r, c);
      slot_accum(g_woc2, dc2 * hv, r, c);
      const float dv = fmaf(lds[OFF_WOC + 2 * H + row], dc2, fmaf(lds[OFF_WOC + H + row], dc1, lds[OFF_WOC + row] * dc0));
      d_hc[r] = hv > 0.0f ? dv : 0.0f;
      slot_accum(g_wa, da * act.h4[r], r, c);
      d_h4[r] = lds[OFF_WA + row] * da;
    }
    // group A staging: [h4 | x2] rows 0..75, h3 rows 96..127, d_hc rows 128.., d_h4pre rows 160..
    store_tile_T(stg_lane, 0, act.h4);
    store_tile_T(stg_lane, 32, e.x2[0]);
    store_tile_T(stg_lane, 64, e.x2[1], 7);
    store_tile_T(stg_lane, 96, act.h3);
    store_tile_T(stg_lane, 128, d_hc);
    mma_bwd<ST_CL>(d_h4, wt_cl, 0, d_hc);
    d_h4 = relu_mask16(d_h4, act.h4);
#pragma unroll
    for (int r = 0; r < 16; ++r) slot_accum(g_bm2, d_h4[r], r, c);
    store_tile_T(stg_lane, 160, d_h4);
    // PE backward, x2 part, one 32-row block at a time
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x16 d_x = zero16();
      mma_bwd<ST_CL>(d_x, wt_cl, 32 + 32 * b, d_hc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int e0 = 32 * b + row0(r);
        if (e0 < OBJ_E2) {
          const PeSel s0 = pe_sel_x2(e0), s1 = pe_sel_x2(e0 + 4);
          const float v = d_x[r] * pe_lane_value<true>(s0, s1, kh, t, proj);
          if (s0.kind == 1) dproj[s0.idx] += kh ? 0.0f : v;
          if (s1.kind == 1) dproj[s1.idx] += kh ? v : 0.0f;
        }
      }
    }
    f32x16 d_h3 = zero16();
    mma_bwd<ST_M>(d_h3, wt_m2, 0, d_h4);
    d_h3 = relu_mask16(d_h3, act.h3);
    __syncthreads();
    {
      const int dTr = (w < 3) ? 128 : 160;
      const int aTr = (w < 3) ? 32 * w : 96;
      wgrad_tile(accA, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    __syncthreads();
    // ---- phase B: cat layer.  [h2 | x1] rows 0..119, d_h3pre rows 128..
    store_tile_T(stg_lane, 0, act.h2);
    store_tile_T(stg_lane, 32, e.x1[0]);
    store_tile_T(stg_lane, 64, e.x1[1]);
    store_tile_T(stg_lane, 96, e.x1[2], 12);
    store_tile_T(stg_lane, 128, d_h3);
    f32x16 d_h2 = zero16();
    mma_bwd<ST_CAT>(d_h2, wt_cat, 0, d_h3);
    d_h2 = relu_mask16(d_h2, act.h2);
#pragma unroll
    for (int r = 0; r < 16; ++r) slot_accum(g_bm1, d_h2[r], r, c);
    f32x16 d_h1 = zero16();
    mma_bwd<ST_M>(d_h1, wt_m1, 0, d_h2);
    d_h1 = relu_mask16(d_h1, act.h1);
    // PE backward, x1 part: d x1 block = cat^T d_h3 + in^T d_h1, consumed block by block
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x16 d_x = zero16();
      mma_bwd<ST_CAT>(d_x, wt_cat, 32 + 32 * b, d_h3);
      mma_bwd<ST_IN>(d_x, wt_in, 32 * b, d_h1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int e0 = 32 * b + row0(r);
        if (e0 < OBJ_E1) {
          const PeSel s0 = pe_sel_x1(e0), s1 = pe_sel_x1(e0 + 4);
          if (s0.kind == 1 || s1.kind == 1) {
            const float v = d_x[r] * pe_lane_value<true>(s0, s1, kh, t, proj);
            if (s0.kind == 1) dproj[s0.idx] += kh ? 0.0f : v;
            if (s1.kind == 1) dproj[s1.idx] += kh ? v : 0.0f;
          }
        }
      }
    }
    // d B[j][:] += dproj[j] * t   (embedding.py:48); lanes kh=0 keep j = 0..10, kh=1 keep j = 11..20
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const float lo = xhalf_sum(dproj[j]);
      const float hi = (j + 11 < OBJ_NDIR) ? xhalf_sum(dproj[j + 11]) : 0.0f;
      const float dv = kh ? hi : lo;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const int i = 3 * j + x;
        slot_accum(i < 16 ? g_B0 : (i < 32 ? g_B1 : g_B2), dv * t[x], i & 15, c);
      }
    }
    __syncthreads();
    wgrad_tile(accB, lane_rd + 128 * STG_LD, lane_rd + (32 * w) * STG_LD);
    __syncthreads();
    // ---- phase C: in layer (x1 stays at rows 32..119) + mid1.  h1 rows 0.., d_h1pre 128.., d_h2pre 160..
    store_tile_T(stg_lane, 0, act.h1);
    store_tile_T(stg_lane, 128, d_h1);
    store_tile_T(stg_lane, 160, d_h2);
    __syncthreads();
    {
      const int dTr = (w < 3) ? 128 : 160;
      const int aTr = (w < 3) ? 32 + 32 * w : 0;
      wgrad_tile(accC, lane_rd + dTr * STG_LD, lane_rd + aTr * STG_LD);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ write this workgroup's slab
  float* slab = a.slab + ((long)k * a.G + g) * a.slab_stride;
  const Layout& L = a.L;
  if (w < 3) write_tile(slab, accA, c, kh, w, L.cl_w, H + OBJ_E2, L.cl_b);
  else write_tile(slab, accA, c, kh, 0, L.m2_w, H, -1);
  write_tile(slab, accB, c, kh, w, L.cat_w, H + OBJ_E1, L.cat_b);
  if (w < 3) write_tile(slab, accC, c, kh, w, L.in_w, OBJ_E1, L.in_b);
  else write_tile(slab, accC, c, kh, 0, L.m1_w, H, -1);

  // slot registers -> LDS (per wave), then sum the 4 waves
  float* red = stg;   // [4][NRED]
  {
    float* mine = red + w * NRED;
    if (c >= 16) {
      const int row = row0(c - 16) + 4 * kh;
      mine[row] = g_bm1; mine[32 + row] = g_bm2; mine[64 + row] = g_wa;
      mine[96 + row] = g_woc0; mine[128 + row] = g_woc1; mine[160 + row] = g_woc2;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int i = 16 * n + (c - 16);          // entry of this half's 33 sums: i = 3*j' + x
        const int j = i / 3 + 11 * kh;
        const float v = n == 0 ? g_B0 : (n == 1 ? g_B1 : g_B2);
        if (i < 33 && j < OBJ_NDIR) mine[196 + 3 * j + (i % 3)] = v;
      }
    }
    const float s0 = wave_sum64(g_ba), s1 = wave_sum64(g_boc0), s2 = wave_sum64(g_boc1), s3 = wave_sum64(g_boc2);
    if (lane == 0) { mine[192] = s0; mine[193] = s1; mine[194] = s2; mine[195] = s3; }
    const float e0 = wave_sum64(l_d), e1 = wave_sum64(l_c), e2 = wave_sum64(l_o);
    if (lane == 0) { mine[259] = e0; mine[260] = e1; mine[261] = e2; mine[262] = 0.0f; }
  }
  __syncthreads();
  for (int i = tid; i < NRED; i += 256) {
    const float v = red[i] + red[NRED + i] + red[2 * NRED + i] + red[3 * NRED + i];
    if (i < 32) slab[L.m1_b + i] = v;
    else if (i < 64) slab[L.m2_b + i - 32] = v;
    else if (i < 96) slab[L.a_w + i - 64] = v;
    else if (i < 192) slab[L.oc_w + i - 96] = v;
    else if (i == 192) slab[L.a_b] = v;
    else if (i < 196) slab[L.oc_b + i - 193] = v;
    else if (i < 259) slab[L.pe_b + i - 196] = v;
    else a.loss_part[((long)k * a.G + g) * 4 + (i - 259)] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// grads[k][i] = sum_g slab[k][g][i] for entries with has_grad; loss_terms[k][:] = sum_g loss_part.
__global__ void finalize_kernel(const float* slab, const float* loss_part, int K, int G, long P, long slab_stride,
                                long p_stride, const uint8_t* has_grad, float* grads, float* loss_terms, int* status) {
  const int k = blockIdx.y;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P && has_grad[i]) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += slab[((long)k * G + g) * slab_stride + i];
    grads[(long)k * p_stride + i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += loss_part[((long)k * G + g) * 4 + threadIdx.x];
    loss_terms[k * 4 + threadIdx.x] = s;
    if (s > 100000.0f) atomicOr(status, 1);       // render_rays.py:109-111
  }
}

// ------------------------------------------------------------------------------------------------
template <bool FEAT>
__global__ __launch_bounds__(256) void eval_kernel(const EvalDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, kh = lane >> 5;
  const int k = blockIdx.x / a.G, g = blockIdx.x % a.G;
  stage_weights(lds, a.params + (long)k * a.p_stride, a.L, FEAT, tid, 256);
  const float scale = a.scale[k];
  const long ntiles = (a.N + TS - 1) / TS;
  for (long tile = g; tile < ntiles; tile += a.G) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    const long n = tile * TS + 32 * w + c;
    const bool valid = n < a.N;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
      const float* p = a.pts + ((long)k * a.N + n) * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    float t[3], proj[OBJ_NDIR];
    project(lds, px, py, pz, scale, t, proj);
    Emb e;
    embed(e, kh, t, proj);
    Acts act;
    Heads hd;
    mlp_forward<FEAT>(lds, c, kh, e, act, hd);
    if (valid) {
      const long o = (long)k * a.N + n;
      if (kh == 0) {
        a.alpha[o] = hd.alpha;
        a.color[o * 3] = hd.col[0];
        a.color[o * 3 + 1] = hd.col[1];
        a.color[o * 3 + 2] = hd.col[2];
      }
      if (FEAT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.hfeat[o * H + row0(r) + 4 * kh] = act.hf[r];
      }
    }
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

static int train_grid(int K, int NT) {
  int G = num_cu() / (K > 0 ? K : 1);
  if (G < 1) G = 1;
  if (G > NT) G = NT;
  if (G < 1) G = 1;
  return G;
}

size_t objnerf_train_workspace_bytes(const objnerf_net* net, int32_t K, int32_t R, int32_t S, int32_t with_feat) {
  if (!net || K <= 0 || R <= 0 || S <= 0) return 0;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  const int64_t ps = objnerf_param_layout(net, offs);
  const int Gmax = num_cu();   // upper bound on G
  (void)with_feat;
  return align256((size_t)K * Gmax * ps * 4) + align256((size_t)K * Gmax * 4 * 4) + align256((size_t)ps) + 256;
}

int objnerf_train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream) {
  (void)hipGetLastError();   // drop stale non-sticky errors of other HIP users
  if (!net || !a || !a->params || !a->scale || !a->z || !a->gt_depth || !a->gt_rgb || !a->labels || !a->counts ||
      !a->flags || !a->grads || !a->loss_terms || !a->status || !a->workspace)
    return OBJNERF_EINVAL;
  if (!a->pts && (!a->origins || !a->dirs)) return OBJNERF_EINVAL;
  if (a->K <= 0 || a->R <= 0 || a->S <= 0) return OBJNERF_EINVAL;
  if (net->hidden != 32 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if (a->S > 64) return OBJNERF_ENOTSUP;
  if (a->gt_feat) return OBJNERF_ENOTSUP;   // feature-distillation branch: see objnerf_feat.hip
  if (a->workspace_bytes < objnerf_train_workspace_bytes(net, a->K, a->R, a->S, a->gt_feat != nullptr))
    return OBJNERF_EINVAL;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  const int64_t ps = objnerf_param_layout(net, offs);
  if (a->p_stride < ps) return OBJNERF_EINVAL;

  TrainDev d;
  d.K = a->K; d.R = a->R; d.S = a->S;
  d.TR = TS / a->S;
  d.NT = (a->R + d.TR - 1) / d.TR;
  d.G = train_grid(a->K, d.NT);
  d.color_scaling = a->color_scaling; d.opacity_scaling = a->opacity_scaling; d.feat_scaling = a->feat_scaling;
  d.obj_center = a->obj_center;
  d.params = a->params; d.p_stride = a->p_stride; d.scale = a->scale;
  d.pts = a->pts; d.origins = a->origins; d.dirs = a->dirs; d.z = a->z;
  d.gt_depth = a->gt_depth; d.gt_rgb = a->gt_rgb; d.labels = a->labels; d.gt_feat = a->gt_feat;
  d.counts = a->counts; d.flags = a->flags;
  d.L = make_layout(net->feat_dim);
  char* ws = (char*)a->workspace;
  const int Gmax = num_cu();
  d.slab = (float*)ws;
  ws += align256((size_t)a->K * Gmax * ps * 4);
  d.loss_part = (float*)ws;
  ws += align256((size_t)a->K * Gmax * 4 * 4);
  uint8_t* has_grad = (uint8_t*)ws;

  hipStream_t st = (hipStream_t)stream;
  // has_grad mask: everything except the feature branch (train.py:435-438 -> .grad stays None)
  hipMemsetAsync(has_grad, 1, (size_t)ps, st);
  hipMemsetAsync(has_grad + d.L.fl_w, 0, (size_t)(d.L.pe_b - d.L.fl_w), st);
  hipMemsetAsync(a->status, 0, sizeof(int), st);

  const size_t lds_bytes = (size_t)(W_FLOATS_NOFEAT + SM_FLOATS + STG_ROWS * STG_LD) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)train_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds_bytes);
    attr_set = true;
  }
  d.slab_stride = ps;
  hipLaunchKernelGGL(train_fused_kernel<false>, dim3(a->K * d.G), dim3(256), lds_bytes, st, d);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  const long P = offs[OBJNERF_N_TENSORS];
  dim3 fg((unsigned)((P + 255) / 256), (unsigned)a->K);
  hipLaunchKernelGGL(finalize_kernel, fg, dim3(256), 0, st, d.slab, d.loss_part, a->K, d.G, P, (long)ps, (long)a->p_stride,
                     has_grad, a->grads, a->loss_terms, a->status);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

int objnerf_eval_points(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                        const float* scale, const float* pts, float* out_alpha, float* out_color, float* out_hfeat,
                        float* out_clip, void* stream) {
  (void)hipGetLastError();   // drop stale non-sticky errors of other HIP users
  if (!net || !params || !scale || !pts || !out_alpha || !out_color || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  if (net->hidden != 32 || net->n_freqs != 6) return OBJNERF_ENOTSUP;
  if (out_clip && !out_hfeat) return OBJNERF_EINVAL;   // the head runs on the H-wide hidden
  EvalDev d;
  d.K = K; d.N = N; d.params = params; d.p_stride = p_stride; d.scale = scale; d.pts = pts;
  d.alpha = out_alpha; d.color = out_color; d.hfeat = out_hfeat;
  d.L = make_layout(net->feat_dim);
  const long ntiles = (N + TS - 1) / TS;
  long G = (2L * num_cu()) / K;
  if (G < 1) G = 1;
  if (G > ntiles) G = ntiles;
  d.G = (int)G;
  hipStream_t st = (hipStream_t)stream;
  const bool feat = out_hfeat != nullptr;
  const size_t lds_bytes = (size_t)(feat ? W_FLOATS_FEAT : W_FLOATS_NOFEAT) * 4;
  if (feat) hipLaunchKernelGGL(eval_kernel<true>, dim3(K * d.G), dim3(256), lds_bytes, st, d);
  else hipLaunchKernelGGL(eval_kernel<false>, dim3(K * d.G), dim3(256), lds_bytes, st, d);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  if (out_clip) return objnerf_feature_head(net, K, N, params, p_stride, out_hfeat, nullptr, out_clip, stream);
  return OBJNERF_OK;
}

}  // extern "C"
