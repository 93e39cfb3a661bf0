// Memory-bound companions of the fused MLP kernels (gfx950): compositing, the materialised-tensor
// loss, label statistics, AdamW, the un-fused embedding, the post-composite feature head, camera ray
// directions and the depth-guided ray sampler.  Built with -ffp-contract=off: every product/sum below
// rounds where the reference's separate ATen ops round; fused multiply-adds are written as fmaf.
#include "objnerf_device.h"
#include "objnerf_philox.h"
#include "../../include/objnerf_hip.h"
#include "objnerf_generic.h"

namespace {

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.0f : (x < 0.f ? -1.0f : 0.0f); }

// ------------------------------------------------------------------------------------------------
// composite: one wave per ray, samples on lanes in chunks of 64 with carries.
// render_rays.py:6-63 / loss.py:27-35,82.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_kernel(long n_rays, int S, int in_is_occ, const float* alpha,
                                                        const float* color, const float* z, const float* vals,
                                                        int V, float* out_term,
                                                        float* out_depth, float* out_var, float* out_rgb,
                                                        float* out_opacity, float* out_vals) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* wv = sm + (long)w * S;
  const long ray = (long)blockIdx.x * 4 + w;
  if (ray >= n_rays) return;
  float carry = 1.0f, D = 0.f, O = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S;
    const float al = on ? alpha[ray * S + s] : 0.f;
    const float occ = on ? (in_is_occ ? al : sigmoid_acc(al)) : 0.f;
    const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;
    const float pinc = SegRows::make(64, lane).scan_mul(fr, lane) * carry;
    float T = __shfl_up(pinc, 1, 64);
    if (lane == 0) T = carry;
    carry = __shfl(pinc, 63, 64);
    const float wt = occ * T;
    if (on) {
      wv[s] = wt;
      if (out_term) out_term[ray * S + s] = wt;
    }
    const float zz = (on && z) ? z[ray * S + s] : 0.f;
    D += wave_sum64(wt * zz);
    O += wave_sum64(wt);
    if (color) {
      const float* cp = color + (ray * S + s) * 3;
      C0 += wave_sum64(on ? wt * cp[0] : 0.f);
      C1 += wave_sum64(on ? wt * cp[1] : 0.f);
      C2 += wave_sum64(on ? wt * cp[2] : 0.f);
    }
  }
  float Vv = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S;
    const float dz = (on && z) ? z[ray * S + s] - D : 0.f;
    Vv += wave_sum64(on ? wv[s] * (dz * dz) : 0.f);
  }
  if (lane == 0) {
    if (out_depth) out_depth[ray] = D;
    if (out_var) out_var[ray] = Vv;
    if (out_opacity) out_opacity[ray] = O;
    if (out_rgb) { out_rgb[ray * 3] = C0; out_rgb[ray * 3 + 1] = C1; out_rgb[ray * 3 + 2] = C2; }
  }
  if (vals && out_vals) {
    for (int v = lane; v < V; v += 64) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += wv[s] * vals[(ray * S + s) * (long)V + v];
      out_vals[ray * (long)V + v] = acc;
    }
  }
}

// (the 512-d head applied after compositing, objnerf_feature_head, is a batched GEMM: objgen::feature_head)

// ------------------------------------------------------------------------------------------------
// embedding.py:46-55, one thread per output entry (coalesced stores; the row pitch is 129 floats).
// ------------------------------------------------------------------------------------------------
__global__ void embed_kernel(int K, long N, int n_freqs, const float* params, long p_stride, long off_B,
                             const float* scale, const float* pts, float* out) {
  const int E = 3 + OBJ_NDIR * n_freqs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (idx >= N * E) return;
  const long i = idx / E;
  const int e = (int)(idx - i * E);
  const float* B = params + (long)k * p_stride + off_B;
  const float sc = scale[k];
  const float* p = pts + ((long)k * N + i) * 3;
  float v;
  if (e < 3) {
    v = p[e] / sc;
  } else {
    const float t0 = p[0] / sc, t1 = p[1] / sc, t2 = p[2] / sc;
    const int f = (e - 3) / OBJ_NDIR, j = (e - 3) - f * OBJ_NDIR;
    const float pj = fmaf(t2, B[3 * j + 2], fmaf(t1, B[3 * j + 1], t0 * B[3 * j]));
    v = sin_acc((pj * (float)(1 << f)) * OBJ_PI_F);
  }
  out[(long)k * N * E + idx] = v;
}

// render_rays.py:6-14 (distances=None branch): occ = sigmoid(alpha)
__global__ void occupancy_kernel(long n, const float* alpha, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sigmoid_acc(alpha[i]);
}

// ------------------------------------------------------------------------------------------------
// label statistics (loss.py:16-21 masks; render_rays.py:88-89 counts)
// ------------------------------------------------------------------------------------------------
__global__ void label_counts_kernel(int K, int R, const uint8_t* labels, int* counts, int* flags) {
  const int k = blockIdx.x;
  int c1 = 0, c2 = 0;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const int l = labels[(long)k * R + r];
    c1 += (l == 1);
    c2 += (l != 2);
  }
  __shared__ int s1[4], s2[4];
  for (int d = 32; d >= 1; d >>= 1) {
    c1 += __shfl_xor(c1, d, 64);
    c2 += __shfl_xor(c2, d, 64);
  }
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = c1; s2[threadIdx.x >> 6] = c2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int a = s1[0] + s1[1] + s1[2] + s1[3], b = s2[0] + s2[1] + s2[2] + s2[3];
    counts[2 * k] = a;
    counts[2 * k + 1] = b;
    if (!flags) return;                 // (counts only: the consumer derives the flags, objnerf_train_common.h)
    if (K == 1) {                       // a single object owns the flags: no zero fill, no atomics
      flags[0] = a == 0;
      flags[1] = b == 0;
    } else {
      if (a == 0) atomicOr(&flags[0], 1);
      if (b == 0) atomicOr(&flags[1], 1);
    }
  }
}

// counts_in path of the loss in ONE launch: copy the (global) counts, derive the early-return flags from them,
// merge the caller's flags, zero the per-object loss terms
__global__ void loss_prologue_kernel(int K, const int* counts_in, int* counts, int* flags, const int* flags_in,
                                     float* loss_terms) {
  __shared__ int f[2];
  if (threadIdx.x < 2) f[threadIdx.x] = (flags_in && flags_in[threadIdx.x]) ? 1 : 0;
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const int a = counts_in[2 * k], b = counts_in[2 * k + 1];
    counts[2 * k] = a;
    counts[2 * k + 1] = b;
    if (a == 0) atomicOr(&f[0], 1);
    if (b == 0) atomicOr(&f[1], 1);
  }
  for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) loss_terms[i] = 0.f;
  __syncthreads();
  if (threadIdx.x < 2) flags[threadIdx.x] = f[threadIdx.x];
}

__global__ void merge_flags_kernel(int* flags, const int* flags_in) {
  if (threadIdx.x < 2 && flags_in && flags_in[threadIdx.x]) flags[threadIdx.x] = 1;
}

// ------------------------------------------------------------------------------------------------
// loss.step_batch_loss on materialised alpha/color (+ optional C-wide predicted features):
// one wave per ray.  Per-object terms are accumulated with float atomics.
// ------------------------------------------------------------------------------------------------
struct LossDev {
  int K, R, S, C;
  float cs, os, fs;
  const float* alpha; const float* color; const float* z; const float* gt_depth; const float* gt_rgb;
  const uint8_t* labels; const float* pred_feat; const float* gt_feat;
  const int* counts; const int* flags;
  float* loss_terms; float* d_alpha; float* d_color; float* d_pred_feat;
  // hoisted feature term (layer-wise training path): the 512-d head is linear, so F = W_of fh + b_of O with
  // fh = sum_s w_s hf_s; only hf [K*R*S][Hh] is materialised, never a C-wide tensor
  int Hh;
  const float* hf;        // [K][R][S][Hh]  relu output of the feature layer
  const float* rayin;     // [K][R][Hh + 2] u = W_of^T g, beta = b_of . g, |g|
  const float* gram;      // [K][Hh * Hh + Hh + 1]  G = W_of^T W_of, wb = W_of^T b_of, bb = b_of . b_of
  float* d_hf;            // [K][R][S][Hh]  d loss / d (pre-activation of the feature layer)
  float* rayfeat;         // [K][R][Hh + 3] fh, O, a, c  (-> head gradient GEMMs)
  float* loss_part;       // NULL: block sums go to loss_terms by float atomics; else [blocks][4] partials, added in block
                          // order by loss_total_kernel (bit-reproducible)
};

__global__ __launch_bounds__(1024) void loss_kernel(const LossDev a) {
  extern __shared__ float sm[];
  __shared__ float s_red[16][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;        // rays per block: divides R (host), so a block never straddles two objects
  const int S = a.S;
  float* wv = sm + (long)w * (3 * S + 2 * a.Hh);      // weights
  float* tv = wv + S;                    // transmittance
  float* dwv = tv + S;                   // feature contribution to dL/dw
  float* sfh = dwv + S;                  // (hoisted mode) composited hidden feature, then d loss / d fh
  float* sgf = sfh + a.Hh;               // (hoisted mode) G fh
  const long rr = (long)blockIdx.x * nw + w;
  const int k = (int)(rr / a.R);
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  const float inv1 = a.flags[0] ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = a.flags[1] ? 0.0f : 1.0f / (n2 + 1e-10f);
  const int lab = a.labels[rr];
  const float m1 = (lab == 1) ? 1.f : 0.f, m2 = (lab != 2) ? 1.f : 0.f, tgt = (lab != 0) ? 1.f : 0.f;
  float carry = 1.0f, D = 0.f, O = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S;
    const float occ = on ? sigmoid_acc(a.alpha[rr * S + s]) : 0.f;
    const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;
    const float pinc = SegRows::make(64, lane).scan_mul(fr, lane) * carry;
    float T = __shfl_up(pinc, 1, 64);
    if (lane == 0) T = carry;
    carry = __shfl(pinc, 63, 64);
    const float wt = occ * T;
    if (on) { wv[s] = wt; tv[s] = T; dwv[s] = 0.f; }
    const float* cp = a.color + (rr * S + (on ? s : 0)) * 3;
    D += wave_sum64(on ? wt * a.z[rr * S + s] : 0.f);
    O += wave_sum64(wt);
    C0 += wave_sum64(on ? wt * cp[0] : 0.f);
    C1 += wave_sum64(on ? wt * cp[1] : 0.f);
    C2 += wave_sum64(on ? wt * cp[2] : 0.f);
  }
  float Vv = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S;
    const float dz = on ? a.z[rr * S + s] - D : 0.f;
    Vv += wave_sum64(on ? wv[s] * (dz * dz) : 0.f);
  }
  const float info = 1.0f / (sqrtf(Vv) + 1e-4f);
  const float rd = D - a.gt_depth[rr];
  const float r0 = C0 - a.gt_rgb[rr * 3], r1 = C1 - a.gt_rgb[rr * 3 + 1], r2 = C2 - a.gt_rgb[rr * 3 + 2];
  const float ro = O - tgt;
  const float gD = m1 * sgnf(rd) * info * inv1;
  const float gC0 = a.cs * m1 * sgnf(r0) * inv1, gC1 = a.cs * m1 * sgnf(r1) * inv1, gC2 = a.cs * m1 * sgnf(r2) * inv1;
  const float gO = a.os * m2 * sgnf(ro) * inv2;
  float lf = 0.f;
  // feature term (loss.py:82-99): F = sum_s w_s f_s ; 1 - cos(F, g)
  if (a.pred_feat && a.gt_feat) {
    const int C = a.C;
    float dotFg = 0.f, nF2 = 0.f, ng2 = 0.f;
    for (int cc = lane; cc < C; cc += 64) {
      float F = 0.f;
      for (int s = 0; s < S; ++s) F += wv[s] * a.pred_feat[(rr * S + s) * (long)C + cc];
      const float gv = a.gt_feat[rr * (long)C + cc];
      dotFg += F * gv; nF2 += F * F; ng2 += gv * gv;
    }
    dotFg = wave_sum64(dotFg); nF2 = wave_sum64(nF2); ng2 = wave_sum64(ng2);
    const float nF = fmaxf(sqrtf(nF2), 1e-8f), ng = fmaxf(sqrtf(ng2), 1e-8f);
    const float cosv = dotFg / (nF * ng);
    lf = m1 * (1.0f - cosv) * inv1;
    const float gs = -a.fs * m1 * inv1;      // d total / d cos
    for (int cc = lane; cc < C; cc += 64) {
      float F = 0.f;
      for (int s = 0; s < S; ++s) F += wv[s] * a.pred_feat[(rr * S + s) * (long)C + cc];
      const float gv = a.gt_feat[rr * (long)C + cc];
      const float dF = gs * (gv / (nF * ng) - cosv * F / (nF * nF));
      for (int s = 0; s < S; ++s) {
        const long idx = (rr * S + s) * (long)C + cc;
        if (a.d_pred_feat) a.d_pred_feat[idx] = dF * wv[s];
        atomicAdd(&dwv[s], dF * a.pred_feat[idx]);
      }
    }
  }
  if (a.hf) {
    const int Hh = a.Hh;
    const float* hfr = a.hf + rr * S * (long)Hh;
    const float* Gk = a.gram + (long)k * ((long)Hh * Hh + Hh + 1);
    const float* wb = Gk + (long)Hh * Hh;
    const float bb = wb[Hh];
    const float* rin = a.rayin + rr * (long)(Hh + 2);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (int h = lane; h < Hh; h += 64) {
      float f = 0.f;
      for (int s = 0; s < S; ++s) f = fmaf(wv[s], hfr[(long)s * Hh + h], f);
      sfh[h] = f;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float fu = 0.f, fGf = 0.f, fwb = 0.f;
    for (int h = lane; h < Hh; h += 64) {
      float gf = 0.f;
      for (int h2 = 0; h2 < Hh; ++h2) gf = fmaf(Gk[(long)h * Hh + h2], sfh[h2], gf);
      sgf[h] = gf;
      fu = fmaf(sfh[h], rin[h], fu);
      fGf = fmaf(sfh[h], gf, fGf);
      fwb = fmaf(sfh[h], wb[h], fwb);
    }
    fu = wave_sum64(fu); fGf = wave_sum64(fGf); fwb = wave_sum64(fwb);
    const float beta = rin[Hh], ngv = rin[Hh + 1];
    const float dotFg = fu + O * beta;
    const float nF2 = fmaxf(fGf + 2.0f * O * fwb + O * O * bb, 0.0f);
    const float nF = fmaxf(sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
    const float cosv = dotFg / (nF * ngc);
    lf = m1 * (1.0f - cosv) * inv1;
    const float gam = -a.fs * m1 * inv1;                 // d total / d cos
    const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
    const float gof = ar * beta + cr * (fwb + O * bb);   // d total / d opacity (feature part)
    float* rf = a.rayfeat + rr * (long)(Hh + 3);
    for (int h = lane; h < Hh; h += 64) {
      rf[h] = sfh[h];
      sfh[h] = ar * rin[h] + cr * (sgf[h] + O * wb[h]);  // d total / d fh
    }
    if (lane == 0) { rf[Hh] = O; rf[Hh + 1] = ar; rf[Hh + 2] = cr; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (int s = 0; s < S; ++s) {
      float pp = 0.f;
      for (int h = lane; h < Hh; h += 64) {
        const float hv = hfr[(long)s * Hh + h];
        pp = fmaf(sfh[h], hv, pp);
        a.d_hf[(rr * S + s) * (long)Hh + h] = hv > 0.0f ? wv[s] * sfh[h] : 0.0f;
      }
      pp = wave_sum64(pp);
      if (lane == 0) dwv[s] = gof + pp;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  // per-object loss terms: the block's rays are combined in LDS first -- thousands of float atomics on ONE address
  // (a single object, e.g. the background network) serialise in L2 and were 65 us of a 600 us step
  if (lane == 0) {
    s_red[w][0] = m1 * fabsf(rd) * info * inv1;
    s_red[w][1] = m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
    s_red[w][2] = m2 * fabsf(ro) * inv2;
    s_red[w][3] = lf;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int i = 0; i < nw; ++i) v += s_red[i][threadIdx.x];
    if (a.loss_part) a.loss_part[(long)blockIdx.x * 4 + threadIdx.x] = v;
    else atomicAdd(&a.loss_terms[k * 4 + threadIdx.x], v);
  }
  if (!a.d_alpha && !a.d_color) return;
  float sufc = 0.f;
  const int nch = (S + 63) / 64;
  for (int ch = nch - 1; ch >= 0; --ch) {
    const int s = ch * 64 + lane;
    const bool on = s < S;
    const float* cp = a.color + (rr * S + (on ? s : 0)) * 3;
    const float c0 = cp[0], c1 = cp[1], c2 = cp[2];
    const float wt = on ? wv[s] : 0.f, T = on ? tv[s] : 0.f;
    const float zz = on ? a.z[rr * S + s] : 0.f;
    const float dw = on ? gD * zz + gO + gC0 * c0 + gC1 * c1 + gC2 * c2 + dwv[s] : 0.f;
    const float qv = dw * wt;
    const float inc = SegRows::make(64, lane).rscan_add(qv, lane);
    const float suf = inc - qv + sufc;
    sufc += __shfl(inc, 0, 64);
    if (on) {
      const float occ = sigmoid_acc(a.alpha[rr * S + s]);
      const float fr = (1.0f - occ) + 1e-10f;
      const float docc = dw * T - suf / fr;
      if (a.d_alpha) a.d_alpha[rr * S + s] = docc * occ * (1.0f - occ);
      if (a.d_color) {
        a.d_color[(rr * S + s) * 3] = gC0 * wt;
        a.d_color[(rr * S + s) * 3 + 1] = gC1 * wt;
        a.d_color[(rr * S + s) * 3 + 2] = gC2 * wt;
      }
    }
  }
}

__global__ void loss_total_kernel(int K, float* terms, float cs, float os, float fs, float* total, int* status,
                                  const float* loss_part, int blocks_per_obj) {
  if (loss_part) {         // the objects' terms from the blocks' partial sums, in block order
    for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) {
      const float* p = loss_part + (long)(i >> 2) * blocks_per_obj * 4 + (i & 3);
      float v = 0.f;
      for (int b = 0; b < blocks_per_obj; ++b) v += p[4 * b];
      terms[i] = v;
    }
    __syncthreads();
  }
  // single thread: K is small; sequential sum keeps the result run-to-run deterministic
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    int bad = 0;
    for (int k = 0; k < K; ++k) {
      const float d = terms[4 * k], c = terms[4 * k + 1], o = terms[4 * k + 2], f = terms[4 * k + 3];
      if (d > 100000.f || c > 100000.f || o > 100000.f || f > 100000.f) bad |= 1;        // render_rays.py:109-111
      if (!(fabsf(d) <= 3.0e38f && fabsf(c) <= 3.0e38f && fabsf(o) <= 3.0e38f && fabsf(f) <= 3.0e38f))
        bad |= 2;   // NaN / Inf: reported, NOT fatal (the reference's `> 100000` is false for NaN and it carries on)
      t += d + c * cs + o * os + f * fs;
    }
    if (total) *total = t;
    if (status) *status = bad;
  }
}

// ------------------------------------------------------------------------------------------------
// torch.optim.AdamW (single-tensor formulation) over the arena.
// ------------------------------------------------------------------------------------------------
__global__ void adamw_kernel(long P, long p_stride, float* params, const float* grads, float* m, float* v,
                             const uint8_t* has_grad, float decay, float w1, float beta2, float w2, float step_size,
                             float bc2_sqrt, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P || (has_grad && !has_grad[i])) return;
  const long idx = (long)blockIdx.y * p_stride + i;
  const float g = grads[idx];
  float p = params[idx] * decay;
  const float mo = m[idx];
  const float mn = mo + w1 * (g - mo);
  const float vn = v[idx] * beta2 + (w2 * g) * g;
  const float denom = sqrtf(vn) / bc2_sqrt + eps;
  p = p + (-step_size) * (mn / denom);
  params[idx] = p;
  m[idx] = mn;
  v[idx] = vn;
}

// The same with the "did this tensor receive a gradient?" decision made ON THE DEVICE from the iteration's early-return
// flags (render_rays.py:89-94).  When the label-1 masks trigger the early return, the depth / colour / feature terms are
// constants: the colour and feature branches get .grad = None and torch.optim.AdamW skips them entirely (no decay, no
// moment update, and their per-parameter step count does not advance); with both flags set nothing has a gradient.
// Three groups with their own step counters (device int32[3]): 0 = trunk + density head + B, 1 = colour branch
// [lo1, lo2), 2 = feature branch [lo2, hi2).
__global__ void adamw_dyn_kernel(long P, long p_stride, float* params, const float* grads, float* m, float* v,
                                 const uint8_t* has_grad, const int* flags, int* steps, int bank, long lo1, long lo2,
                                 long hi2, double lr, double b1, double b2, float eps, double wd, long ng_lo, long ng_hi) {
  // step counters: two banks of three; this call READS bank `bank` and its first block WRITES the advanced counters to
  // the other one -- no second launch, and no block can see a counter of its own call already advanced
  __shared__ float s_step_size[3], s_bc2_sqrt[3];
  __shared__ int s_active[3];
  if (threadIdx.x < 3) {
    const int g = threadIdx.x;
    const bool f0 = flags[0] != 0, f1 = flags[1] != 0;
    s_active[g] = g == 0 ? !(f0 && f1) : !f0;
    const int old = steps[3 * bank + g];
    const double st = (double)(old + 1);
    s_step_size[g] = (float)(lr / (1.0 - pow(b1, st)));
    s_bc2_sqrt[g] = (float)sqrt(1.0 - pow(b2, st));
    if (blockIdx.x == 0 && blockIdx.y == 0) steps[3 * (1 - bank) + g] = old + (s_active[g] ? 1 : 0);
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P || (has_grad && !has_grad[i]) || (i >= ng_lo && i < ng_hi)) return;
  const int g = (i >= lo1 && i < lo2) ? 1 : ((i >= lo2 && i < hi2) ? 2 : 0);
  if (!s_active[g]) return;
  const float decay = (float)(1.0 - lr * wd), w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2), beta2 = (float)b2;
  const long idx = (long)blockIdx.y * p_stride + i;
  const float gr = grads[idx];
  float p = params[idx] * decay;
  const float mo = m[idx];
  const float mn = mo + w1 * (gr - mo);
  const float vn = v[idx] * beta2 + (w2 * gr) * gr;
  const float denom = sqrtf(vn) / s_bc2_sqrt[g] + eps;
  p = p + (-s_step_size[g]) * (mn / denom);
  params[idx] = p;
  m[idx] = mn;
  v[idx] = vn;
}

// ------------------------------------------------------------------------------------------------
// vmap.py:701-720
// ------------------------------------------------------------------------------------------------
__global__ void rays_dirs_kernel(int W, int H, float fx, float fy, float cx, float cy, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)W * H) return;
  const int iw = (int)(i / H), ih = (int)(i % H);
  out[i * 3] = ((float)iw - cx) / fx;
  out[i * 3 + 1] = ((float)ih - cy) / fy;
  out[i * 3 + 2] = 1.0f;
}

// ------------------------------------------------------------------------------------------------
// vmap.py:386-554: pixel gather (pass 1, also the batch depth maximum) and z placement (pass 2).
// ------------------------------------------------------------------------------------------------
// object blockIdx.y of a stacked call: its keyframe store from the table, its slices of the stacked arrays
__device__ __forceinline__ objnerf_sample_args sample_args_of(const objnerf_sample_args& a, const objnerf_kf_store* table,
                                                              const int k) {
  objnerf_sample_args b = a;
  if (table) {
    b.rgbs = table[k].rgbs; b.depth = table[k].depth; b.t_wc = table[k].t_wc; b.bbox = table[k].bbox;
    const long n = (long)a.n_frames * a.n_px, S = a.n_cam2surf + a.n_bins;
    if (b.kf_ids) b.kf_ids += (long)k * a.n_frames;
    if (b.u_w) { b.u_w += k * n; b.u_h += k * n; b.u += k * n * S; b.g += k * n * a.n_bins; }
    if (b.kf_meta) b.kf_meta += 4 * k;
    if (b.out_kf) b.out_kf += (long)k * a.n_frames;
    if (b.out_px) b.out_px += k * n * 2;
    if (b.out_origins) { b.out_origins += k * n * 3; b.out_dirs += k * n * 3; }
    if (b.out_partfeat) { b.out_partfeat += k * n * b.pf_c; b.use_frame += (long)k * a.F; }
    if (b.out_pts) b.out_pts += k * n * S * 3;
    b.obj_index = a.obj_index + k;
    b.out_rgb += k * n * 3; b.out_depth += k * n; b.out_valid += k * n; b.out_labels += k * n;
    b.out_z += k * n * S;
    b.max_depth_ws += k * (1 + 6 * n);
  }
  if (b.kf_meta) b.obj_index = b.kf_meta[3];      // the object's own random-stream id
  return b;
}

__global__ __launch_bounds__(256) void sample_gather_kernel(const objnerf_sample_args a_, const objnerf_kf_store* table) {
  const objnerf_sample_args a = sample_args_of(a_, table, blockIdx.y);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n_frames * a.n_px;
  __shared__ long pf_row[256];                                   // source row of every ray of the block (part features)
  if (i < n) {
    float* origins_ws = a.out_origins ? a.out_origins : a.max_depth_ws + 1;
    float* dirs_ws = a.out_origins ? a.out_dirs : a.max_depth_ws + 1 + (size_t)n * 3;
    const int f = i / a.n_px;
    const uint32_t tag = (uint32_t)a.draw << 3;
    long kf;
    if (a.kf_ids) {
      kf = a.kf_ids[f];
    } else {                                                       // vmap.py:390-401
      const int nk = a.kf_meta[0];
      const int tail = f - (a.n_frames - 2);                       // 0 / 1 for the last two frames of the draw
      if (tail >= 0 && a.kf_meta[1 + tail] >= 0) {
        kf = a.kf_meta[1 + tail];
      } else {
        const float uk = objrng::uniform1(a.seed, objrng::S_KEYFRAME | tag, (uint32_t)a.obj_index, 0u, (uint32_t)f);
        kf = min((int)(uk * (float)nk), nk - 1);
      }
      if (a.out_kf && i == f * a.n_px) a.out_kf[f] = kf;
    }
    const float* bb = a.bbox + kf * 4;
    float uw, uh;
    if (a.u_w) {
      uw = a.u_w[i]; uh = a.u_h[i];
    } else {
      float r4[4];
      objrng::uniform4(a.seed, objrng::S_PIXEL_W | tag, (uint32_t)a.obj_index, (uint32_t)i, 0u, r4);
      uw = r4[0]; uh = r4[1];
    }
    const float fw = uw * (bb[1] - bb[0]) + bb[0];
    const float fh = uh * (bb[3] - bb[2]) + bb[2];
    const long iw = (long)fw, ih = (long)fh;                       // .long() truncation (vmap.py:418-419)
    if (a.out_px) { a.out_px[2 * i] = (int)iw; a.out_px[2 * i + 1] = (int)ih; }
    const long pix = (kf * a.W + iw) * a.H + ih;
    const uint8_t* px = a.rgbs + pix * 4;
    a.out_rgb[i * 3] = px[0]; a.out_rgb[i * 3 + 1] = px[1]; a.out_rgb[i * 3 + 2] = px[2];
    a.out_labels[i] = px[3];
    const float d = a.depth[pix];
    a.out_depth[i] = d;
    atomicMax((int*)a.max_depth_ws, __float_as_int(fmaxf(d, 0.0f)));
    const float* dc = a.rays_dir_cache + (iw * a.H + ih) * 3;
    const float* T = a.t_wc + kf * 16;
    for (int r = 0; r < 3; ++r) {
      dirs_ws[i * 3 + r] = fmaf(T[r * 4 + 2], dc[2], fmaf(T[r * 4 + 1], dc[1], T[r * 4] * dc[0]));
      origins_ws[i * 3 + r] = T[r * 4 + 3];
    }
    if (a.out_partfeat) {                                          // vmap.py:437-452
      long fid = (long)((double)a.use_frame[kf] / (double)a.pf_stride);          // :439-440 (float64, truncated)
      long pw = (long)floorf(fw / a.part_down), ph = (long)floorf(fh / a.part_down);   // :441-442 (fp32 quotient)
      fid = min(max(fid, 0l), (long)a.pf_frames - 1);
      pw = min(max(pw, 0l), (long)a.pf_w - 1);
      ph = min(max(ph, 0l), (long)a.pf_h - 1);
      pf_row[threadIdx.x] = (fid * a.pf_w + pw) * a.pf_h + ph;
    }
  }
  if (!a.out_partfeat) return;                                     // (uniform over the launch)
  __syncthreads();
  // the block's rows, one wave per ray and 16 bytes per lane: global_partfeat[fid, pw, ph, :] -> out_partfeat[i, :]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.pf_c;
  const int r_end = min(256, n - (int)(blockIdx.x * blockDim.x));
  const bool vec = (C & 3) == 0 && (((size_t)a.global_partfeat | (size_t)a.out_partfeat) & 15) == 0;
  for (int r = wave; r < r_end; r += 4) {
    const float* src = a.global_partfeat + pf_row[r] * C;
    float* dst = a.out_partfeat + ((long)blockIdx.x * blockDim.x + r) * C;
    if (vec) {
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      for (int c = lane; c < C / 4; c += 64) ((f32x4v*)dst)[c] = ((const f32x4v*)src)[c];
    } else {
      for (int c = lane; c < C; c += 64) dst[c] = src[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// train.py:196-256: one frame into a keyframe slot of every visible object (blockIdx.y = object)
// ------------------------------------------------------------------------------------------------
__global__ void ingest_frame_kernel(int W, int H, const uint8_t* rgb, const float* depth, const int32_t* inst,
                                    const float* t_wc, const objnerf_ingest_item* items) {
  const objnerf_ingest_item it = items[blockIdx.y];
  const long npx = (long)W * H;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 16) it.t_wc[(long)it.slot * 16 + threadIdx.x] = t_wc[threadIdx.x];
    if (threadIdx.x < 4) it.bbox[(long)it.slot * 4 + threadIdx.x] = it.box[threadIdx.x];
  }
  if (p >= npx) return;
  const int id = inst[p];
  const uint32_t state = id == it.obj_id ? 1u : (id == -1 ? 2u : 0u);        // train.py:201-203
  const uint32_t px = (uint32_t)rgb[p * 3] | ((uint32_t)rgb[p * 3 + 1] << 8) | ((uint32_t)rgb[p * 3 + 2] << 16) |
                      (state << 24);
  reinterpret_cast<uint32_t*>(it.rgbs)[(long)it.slot * npx + p] = px;        // [F][W][H][4] u8: rgb + state
  it.depth[(long)it.slot * npx + p] = depth[p];
}


// ------------------------------------------------------------------------------------------------
// Trainer.sample_points_bbox (trainer.py:130-198)
// ------------------------------------------------------------------------------------------------
__global__ void box_rays_kernel(long P, const float* T_WC, const float* T_OC, const float* half, const float* dirs_C,
                                float* dirs_W, float* near_o, float* far_o, uint8_t* hit_o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float d0 = dirs_C[i * 3], d1 = dirs_C[i * 3 + 1], d2 = dirs_C[i * 3 + 2];
  float nr = -INFINITY, fr = INFINITY;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // utils.py:324-336: R @ d, products accumulated left to right
    dirs_W[i * 3 + r] = (T_WC[r * 4] * d0 + T_WC[r * 4 + 1] * d1) + T_WC[r * 4 + 2] * d2;
    const float dr = (T_OC[r * 4] * d0 + T_OC[r * 4 + 1] * d1) + T_OC[r * 4 + 2] * d2;
    const float o = T_OC[r * 4 + 3];
    const float ta = (-half[r] - o) / dr, tb = (half[r] - o) / dr;       // utils.py:311-312
    nr = fmaxf(nr, fminf(ta, tb));
    fr = fminf(fr, fmaxf(ta, tb));
  }
  hit_o[i] = (nr <= fr) && (fr > 0.f);                                    // :317
  near_o[i] = fmaxf(nr, 0.f);                                             // trainer.py:166
  far_o[i] = fr + 0.2f;                                                   // :167
}

__global__ void box_points_kernel(long n, int n_bins, const float* origin, const float* dirs_W, const float* near_i,
                                  const float* far_i, const float* u, uint64_t seed, uint32_t draw, float* out_z,
                                  float* out_pts) {
  const int S = n_bins - 1;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * S) return;
  const long r = idx / S;
  const int s = (int)(idx - r * S);
  const float lo = near_i[r], hi = far_i[r];
  const uint32_t st = objrng::S_BOX_U | (draw << 3);
  const float u0 = u ? u[r * n_bins + s] : objrng::uniform1(seed, st, (uint32_t)(r >> 32), (uint32_t)r, (uint32_t)s);
  const float u1 = u ? u[r * n_bins + s + 1] : objrng::uniform1(seed, st, (uint32_t)(r >> 32), (uint32_t)r, (uint32_t)s + 1);
  const float z0 = strat(lo, hi, s, n_bins, u0);
  const float z1 = strat(lo, hi, s + 1, n_bins, u1);
  const float z = 0.5f * (z1 + z0);                                       // trainer.py:175
  out_z[idx] = z;
#pragma unroll
  for (int x = 0; x < 3; ++x) out_pts[idx * 3 + x] = origin[x] + dirs_W[r * 3 + x] * z;   // :176
}

__global__ void sample_place_kernel(const objnerf_sample_args a_, const objnerf_kf_store* table) {
  const objnerf_sample_args a = sample_args_of(a_, table, blockIdx.y);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n_frames * a.n_px;
  if (i >= n) return;
  const float* origins_ws = a.out_origins ? a.out_origins : a.max_depth_ws + 1;
  const float* dirs_ws = a.out_origins ? a.out_dirs : a.max_depth_ws + 1 + (size_t)n * 3;
  const int N = a.n_cam2surf, M = a.n_bins, S = N + M;
  const float d = a.out_depth[i];
  const float maxd = *a.max_depth_ws;
  const uint8_t lab = a.out_labels[i];
  float* z = a.out_z + (long)i * S;
  const bool invalid = d <= a.min_bound;
  a.out_valid[i] = invalid ? 0 : 1;
  if (a.u) {
    const float* u = a.u + (long)i * S;
    if (invalid) {
      for (int s = 0; s < S; ++s) z[s] = strat(a.min_bound, maxd, s, S, u[s]);
    } else {
      for (int s = 0; s < N; ++s) z[s] = strat(a.min_bound, d - a.surface_eps, s, N, u[s]);
      if (lab == 1) {
        // sorted N(0,(eps/3)^2) draws, clipped to +-eps, around the surface (utils.py:382-397)
        const float* g = a.g + (long)i * M;
        for (int s = 0; s < M; ++s) {          // rank sort: stable for ties
          const float v = g[s];
          int rank = 0;
          for (int j = 0; j < M; ++j) rank += (g[j] < v) || (g[j] == v && j < s);
          z[N + rank] = d + fminf(fmaxf(v, -a.surface_eps), a.surface_eps);
        }
      } else {
        for (int s = 0; s < M; ++s) z[N + s] = strat(d - a.surface_eps, d + a.stop_eps, s, M, u[N + s]);
      }
    }
  } else {
    // seeded: the same placement, every draw a function of (seed; purpose, draw; object, ray, bin)
    const uint32_t tag = (uint32_t)a.draw << 3, ko = (uint32_t)a.obj_index;
    const bool normal = !invalid && lab == 1;
    const int n_u = invalid ? S : (normal ? N : S);
    for (int s0 = 0; s0 < n_u; s0 += 4) {
      float r4[4];
      objrng::uniform4(a.seed, objrng::S_BINS_U | tag, ko, (uint32_t)i, (uint32_t)(s0 >> 2), r4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s0 + j;
        if (s >= n_u) break;
        z[s] = invalid ? strat(a.min_bound, maxd, s, S, r4[j])
               : s < N ? strat(a.min_bound, d - a.surface_eps, s, N, r4[j])
                       : strat(d - a.surface_eps, d + a.stop_eps, s - N, M, r4[j]);
      }
    }
    if (normal) {
      const float sd = a.surface_eps / 3.0f;
      for (int s = 0; s < M; ++s)
        z[N + s] = fminf(fmaxf(sd * objrng::normal1(a.seed, objrng::S_BINS_G | tag, ko, (uint32_t)i, (uint32_t)s),
                               -a.surface_eps), a.surface_eps);
      for (int s = 1; s < M; ++s) {          // insertion sort in place (clipping is monotone: sort after it)
        const float v = z[N + s];
        int j = s - 1;
        while (j >= 0 && z[N + j] > v) { z[N + j + 1] = z[N + j]; --j; }
        z[N + j + 1] = v;
      }
      for (int s = 0; s < M; ++s) z[N + s] += d;
    }
  }
  if (!a.out_pts) return;
  const float* o = origins_ws + (long)i * 3;
  const float* dr = dirs_ws + (long)i * 3;
  for (int s = 0; s < S; ++s) {
    float* p = a.out_pts + ((long)i * S + s) * 3;
    p[0] = (o[0] + dr[0] * z[s]) - a.obj_center;
    p[1] = (o[1] + dr[1] * z[s]) - a.obj_center;
    p[2] = (o[2] + dr[2] * z[s]) - a.obj_center;
  }
}

// sceneObject.sample_3d_points alone (vmap.py:456-554): the pixels are already gathered.  This pass puts them where
// sample_place_kernel expects the gather pass's outputs: rgb / state split (:552-554), the batch depth maximum (:489),
// the per-frame origin broadcast over the frame's rays (:548).
__global__ void sample_prepare_kernel(const objnerf_sample_args a, const uint8_t* __restrict__ rgbs4,
                                      const float* __restrict__ depth, const float* __restrict__ origins,
                                      const float* __restrict__ dirs_w, float* origins_ws, float* dirs_ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n_frames * a.n_px;
  if (i >= n) return;
  const uint8_t* px = rgbs4 + (long)i * 4;
  a.out_rgb[i * 3] = px[0]; a.out_rgb[i * 3 + 1] = px[1]; a.out_rgb[i * 3 + 2] = px[2];
  a.out_labels[i] = px[3];
  const float d = depth[i];
  a.out_depth[i] = d;
  atomicMax((int*)a.max_depth_ws, __float_as_int(fmaxf(d, 0.0f)));
  const int f = i / a.n_px;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    origins_ws[i * 3 + r] = origins[f * 3 + r];
    dirs_ws[i * 3 + r] = dirs_w[i * 3 + r];
  }
}

}  // namespace

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH; } while (0)
// a stale non-sticky error of another HIP user of this thread (e.g. hipErrorNotReady from event polling)
// must not be mistaken for a launch failure
#define CLEAR_STALE() (void)hipGetLastError()

namespace objmisc {
int step_batch_loss_impl(const objnerf_loss_args* a, const LossHoisted* hz, void* stream, float* loss_part) {
  CLEAR_STALE();
  if (!a || !a->alpha || !a->color || !a->z || !a->gt_depth || !a->gt_rgb || !a->labels || !a->loss_terms ||
      !a->counts || a->K <= 0 || a->R <= 0 || a->S <= 0)
    return OBJNERF_EINVAL;
  if (!hz && (a->pred_feat == nullptr) != (a->gt_feat == nullptr)) return OBJNERF_EINVAL;
  if (hz && (!a->gt_feat || a->pred_feat || !hz->hf || !hz->rayin || !hz->gram || !hz->d_hf || !hz->rayfeat || hz->Hh <= 0))
    return OBJNERF_EINVAL;
  if (a->S > 2048) return OBJNERF_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  int* flags = a->counts + 2 * a->K;     // counts workspace is [K][2] + [2]
  if (a->counts_in) {
    hipLaunchKernelGGL(loss_prologue_kernel, dim3(1), dim3(256), 0, st, a->K, a->counts_in, a->counts, flags, a->flags_in,
                       a->loss_terms);
  } else {
    if (a->K > 1) (void)hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
    hipLaunchKernelGGL(label_counts_kernel, dim3(a->K), dim3(256), 0, st, a->K, a->R, a->labels, a->counts, flags);
    if (a->flags_in) hipLaunchKernelGGL(merge_flags_kernel, dim3(1), dim3(64), 0, st, flags, a->flags_in);
    (void)hipMemsetAsync(a->loss_terms, 0, (size_t)a->K * 4 * sizeof(float), st);
  }
  LossDev d;
  d.K = a->K; d.R = a->R; d.S = a->S; d.C = a->C;
  d.cs = a->color_scaling; d.os = a->opacity_scaling; d.fs = a->feat_scaling;
  d.alpha = a->alpha; d.color = a->color; d.z = a->z; d.gt_depth = a->gt_depth; d.gt_rgb = a->gt_rgb;
  d.labels = a->labels; d.pred_feat = a->pred_feat; d.gt_feat = a->gt_feat; d.counts = a->counts; d.flags = flags;
  d.loss_terms = a->loss_terms; d.d_alpha = a->d_alpha; d.d_color = a->d_color; d.d_pred_feat = a->d_pred_feat;
  d.Hh = hz ? hz->Hh : 0;
  d.hf = hz ? hz->hf : nullptr; d.rayin = hz ? hz->rayin : nullptr; d.gram = hz ? hz->gram : nullptr;
  d.d_hf = hz ? hz->d_hf : nullptr; d.rayfeat = hz ? hz->rayfeat : nullptr;
  const long nr = (long)a->K * a->R;
  int rb = 16;                                          // rays (waves) per block: the largest power of two dividing R
  while (rb > 1 && (a->R % rb != 0 || (size_t)rb * (3 * a->S + 2 * d.Hh) * 4 > 60000)) rb >>= 1;
  d.loss_part = loss_part;
  hipLaunchKernelGGL(loss_kernel, dim3((unsigned)(nr / rb)), dim3(64 * rb), (size_t)rb * (3 * a->S + 2 * d.Hh) * 4, st, d);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(256), 0, st, a->K, a->loss_terms, a->color_scaling,
                     a->opacity_scaling, (a->pred_feat || hz) ? a->feat_scaling : 0.0f, a->total, a->status,
                     (const float*)loss_part, a->R / rb);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}
int label_counts_only(int32_t K, int32_t R, const uint8_t* labels, int32_t* counts, void* stream) {
  CLEAR_STALE();
  if (K <= 0 || R <= 0 || !labels || !counts) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(label_counts_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, K, R, labels, counts, (int*)nullptr);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}
int adamw_flags_range(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads, float* exp_avg,
                      float* exp_avg_sq, const uint8_t* has_grad, const int32_t* flags, int32_t* group_steps, int32_t bank,
                      int64_t colour_lo, int64_t feature_lo, int64_t feature_hi, int64_t ng_lo, int64_t ng_hi, float lr,
                      float beta1, float beta2, float eps, float weight_decay, void* stream) {
  CLEAR_STALE();
  if (K <= 0 || P <= 0 || p_stride < P || !params || !grads || !exp_avg || !exp_avg_sq || !flags || !group_steps ||
      (bank != 0 && bank != 1) ||
      colour_lo > feature_lo || feature_lo > feature_hi || feature_hi > P)
    return OBJNERF_EINVAL;
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)K);
  hipLaunchKernelGGL(adamw_dyn_kernel, grid, dim3(256), 0, (hipStream_t)stream, (long)P, (long)p_stride, params, grads,
                     exp_avg, exp_avg_sq, has_grad, flags, group_steps, (int)bank, (long)colour_lo, (long)feature_lo,
                     (long)feature_hi, (double)lr, (double)beta1, (double)beta2, eps, (double)weight_decay, (long)ng_lo,
                     (long)ng_hi);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}
}  // namespace objmisc

// Saturated-MFMA loop for bench.py's `roofline.peak_measured`: 256-thread workgroups (one wave per SIMD), TWO
// independent 32x32 accumulator chains per wave (a 16-pass MFMA issues every 64 cycles per chain: two chains keep
// the pipe full with nothing else to issue), operands in registers, lane-dependent non-trivial data (the clock the
// chip holds depends on the operands: MI355X_MICROARCH.md, DVFS give-back).  The instructions are the ones the training
// kernels use.  DT 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16; four MFMAs per iteration.
template <int DT>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
  typedef float f16v __attribute__((ext_vector_type(16)));
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  f16v a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  const float av = 0.37f + 0.011f * lane, bv = -0.52f + 0.007f * (lane ^ 21);
  bf8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(av + 0.05f * i); bb[i] = (__bf16)(bv - 0.03f * i); }
  for (int i = 0; i < iters; ++i) {
    if (DT == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, av, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, bv, a1, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, bb, a1, 0, 0, 0);
    }
    // keep the accumulators bounded without leaving the MFMA-only regime: rescale once in a while
    if ((i & 127) == 127) { a0 *= 1e-3f; a1 *= 1e-3f; }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
  sink[blockIdx.x * 256 + threadIdx.x] = r;
}

extern "C" {

int objnerf_composite(int64_t n_rays, int32_t S, int32_t flags, const float* alpha, const float* color,
                      const float* z, const float* vals, int32_t V, float* out_term, float* out_depth, float* out_var,
                      float* out_rgb, float* out_opacity, float* out_vals, void* stream) {
  CLEAR_STALE();
  if (n_rays <= 0 || S <= 0 || !alpha) return OBJNERF_EINVAL;
  if (!z && (out_depth || out_var)) return OBJNERF_EINVAL;
  if (S > 4096) return OBJNERF_ENOTSUP;
  hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), (size_t)4 * S * 4,
                     (hipStream_t)stream, (long)n_rays, S, (int)(flags & OBJNERF_COMPOSITE_INPUT_IS_OCCUPANCY), alpha,
                     color, z, vals, V, out_term, out_depth, out_var, out_rgb, out_opacity, out_vals);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_mfma_peak(int32_t dtype, int32_t iters, int32_t n_wg, float* sink, void* stream) {
  CLEAR_STALE();
  if (!sink || iters <= 0 || n_wg <= 0 || (dtype != 0 && dtype != 1)) return OBJNERF_EINVAL;
  if (dtype == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, iters, sink);
  else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, iters, sink);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_occupancy(int64_t n, const float* alpha, float* out, void* stream) {
  CLEAR_STALE();
  if (n <= 0 || !alpha || !out) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n,
                     alpha, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_feature_head(const objnerf_net* net, int32_t K, int64_t n, const float* params, int64_t p_stride,
                         const float* hfeat, const float* weight, float* out, void* stream) {
  CLEAR_STALE();
  if (!net || !params || !hfeat || !out || K <= 0 || n <= 0) return OBJNERF_EINVAL;
  if (net->hidden % 32 != 0) return OBJNERF_ENOTSUP;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  if (objnerf_param_layout(net, offs) < 0) return OBJNERF_EINVAL;
  // a batched GEMM [n x H] [H x C] with the bias scaled per row (objnerf_generic.hip)
  objgen::feature_head(stream, K, (long)n, net->hidden, net->feat_dim, params, (long)p_stride, (long)offs[OBJNERF_T_OF_W],
                       (long)offs[OBJNERF_T_OF_B], hfeat, weight, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_embed(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                  const float* scale, const float* pts, float* out_emb, void* stream) {
  CLEAR_STALE();
  if (!net || !params || !scale || !pts || !out_emb || K <= 0 || N <= 0) return OBJNERF_EINVAL;
  int64_t offs[OBJNERF_N_TENSORS + 1];
  if (objnerf_param_layout(net, offs) < 0) return OBJNERF_EINVAL;
  const long total = (long)N * (3 + OBJ_NDIR * net->n_freqs);
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)K);
  hipLaunchKernelGGL(embed_kernel, grid, dim3(256), 0, (hipStream_t)stream, K, (long)N, net->n_freqs, params,
                     (long)p_stride, (long)offs[OBJNERF_T_PE_B], scale, pts, out_emb);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_label_counts(int32_t K, int32_t R, const uint8_t* labels, int32_t* counts, int32_t* flags_out,
                         void* stream) {
  CLEAR_STALE();
  if (K <= 0 || R <= 0 || !labels || !counts || !flags_out) return OBJNERF_EINVAL;
  if (K > 1) (void)hipMemsetAsync(flags_out, 0, 2 * sizeof(int), (hipStream_t)stream);
  hipLaunchKernelGGL(label_counts_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, K, R, labels, counts, flags_out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_step_batch_loss(const objnerf_loss_args* a, void* stream) {
  return objmisc::step_batch_loss_impl(a, nullptr, stream, nullptr);
}

int objnerf_adamw_step(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads, float* exp_avg,
                       float* exp_avg_sq, const uint8_t* has_grad, int32_t step, float lr, float beta1, float beta2,
                       float eps, float weight_decay, void* stream) {
  CLEAR_STALE();
  if (K <= 0 || P <= 0 || p_stride < P || !params || !grads || !exp_avg || !exp_avg_sq || step < 1)
    return OBJNERF_EINVAL;
  // scalars exactly as torch.optim.adamw._single_tensor_adamw forms them (python doubles -> fp32 op)
  const double lr_d = (double)lr, b1 = (double)beta1, b2 = (double)beta2, wd = (double)weight_decay;
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  const float decay = (float)(1.0 - lr_d * wd);
  const float w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2);
  const float step_size = (float)(lr_d / bc1), bc2_sqrt = (float)sqrt(bc2);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)K);
  hipLaunchKernelGGL(adamw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (long)P, (long)p_stride, params, grads,
                     exp_avg, exp_avg_sq, has_grad, decay, w1, beta2, w2, step_size, bc2_sqrt, eps);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_adamw_step_flags(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads, float* exp_avg,
                             float* exp_avg_sq, const uint8_t* has_grad, const int32_t* flags, int32_t* group_steps,
                             int32_t bank, int64_t colour_lo, int64_t feature_lo, int64_t feature_hi, float lr, float beta1,
                             float beta2, float eps, float weight_decay, void* stream) {
  return objmisc::adamw_flags_range(K, P, p_stride, params, grads, exp_avg, exp_avg_sq, has_grad, flags, group_steps, bank,
                                    colour_lo, feature_lo, feature_hi, P, P, lr, beta1, beta2, eps, weight_decay, stream);
}

int objnerf_rays_dirs(int32_t W, int32_t H, float fx, float fy, float cx, float cy, float* out, void* stream) {
  CLEAR_STALE();
  if (W <= 0 || H <= 0 || !out) return OBJNERF_EINVAL;
  const long n = (long)W * H;
  hipLaunchKernelGGL(rays_dirs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, H, fx,
                     fy, cx, cy, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_box_rays(int64_t P, const float* T_WC, const float* T_OC, const float* half_extent, const float* dirs_C,
                     float* out_dirs_W, float* out_near, float* out_far, uint8_t* out_hit, void* stream) {
  CLEAR_STALE();
  if (P <= 0 || !T_WC || !T_OC || !half_extent || !dirs_C || !out_dirs_W || !out_near || !out_far || !out_hit)
    return OBJNERF_EINVAL;
  hipLaunchKernelGGL(box_rays_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)P, T_WC,
                     T_OC, half_extent, dirs_C, out_dirs_W, out_near, out_far, out_hit);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_box_points(int64_t n, int32_t n_bins, const float* origin, const float* dirs_W, const float* near,
                       const float* far, const float* u, uint64_t seed, uint32_t draw, float* out_z, float* out_pts,
                       void* stream) {
  CLEAR_STALE();
  if (n <= 0 || n_bins < 2 || !origin || !dirs_W || !near || !far || !out_z || !out_pts) return OBJNERF_EINVAL;
  const long total = (long)n * (n_bins - 1);
  hipLaunchKernelGGL(box_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n,
                     n_bins, origin, dirs_W, near, far, u, seed, draw, out_z, out_pts);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

// the injected draws come all together or not at all; seeded keyframes need kf_meta; points or origins + directions
static bool sample_args_ok(const objnerf_sample_args* a) {
  const int n_inj = (a->u_w != nullptr) + (a->u_h != nullptr) + (a->u != nullptr) + (a->g != nullptr);
  if (n_inj != 0 && n_inj != 4) return false;
  if (!a->kf_ids && (n_inj != 0 || !a->kf_meta)) return false;
  if (!a->out_pts && (!a->out_origins || !a->out_dirs)) return false;
  if ((a->out_origins != nullptr) != (a->out_dirs != nullptr)) return false;
  if (a->out_partfeat && (!a->global_partfeat || !a->use_frame || a->pf_frames <= 0 || a->pf_w <= 0 || a->pf_h <= 0 ||
                          a->pf_c <= 0 || a->pf_stride <= 0 || !(a->part_down > 0.0f))) return false;
  return a->rays_dir_cache && a->out_rgb && a->out_depth && a->out_valid && a->out_labels && a->out_z &&
         a->max_depth_ws && a->n_frames > 0 && a->n_px > 0 && a->n_cam2surf > 0 && a->n_bins > 0;
}

int objnerf_sample_rays(const objnerf_sample_args* a, void* stream) {
  CLEAR_STALE();
  if (!a || !a->rgbs || !a->depth || !a->t_wc || !a->bbox || !sample_args_ok(a)) return OBJNERF_EINVAL;
  const int n = a->n_frames * a->n_px;
  hipStream_t st = (hipStream_t)stream;
  // (max_depth_ws[0] = the batch depth maximum, then the world-frame origins / directions between the passes)
  (void)hipMemsetAsync(a->max_depth_ws, 0, sizeof(float), st);
  hipLaunchKernelGGL(sample_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *a, (const objnerf_kf_store*)nullptr);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(sample_place_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *a, (const objnerf_kf_store*)nullptr);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_sample_points(const objnerf_sample_args* a, const uint8_t* sampled_rgbs, const float* sampled_depth,
                          const float* origins, const float* dirs_w, void* stream) {
  CLEAR_STALE();
  if (!a || !sampled_rgbs || !sampled_depth || !origins || !dirs_w) return OBJNERF_EINVAL;
  if ((a->u != nullptr) != (a->g != nullptr)) return OBJNERF_EINVAL;
  if (!a->out_pts && (!a->out_origins || !a->out_dirs)) return OBJNERF_EINVAL;
  if ((a->out_origins != nullptr) != (a->out_dirs != nullptr)) return OBJNERF_EINVAL;
  if (!(a->out_rgb && a->out_depth && a->out_valid && a->out_labels && a->out_z && a->max_depth_ws && a->n_frames > 0 &&
        a->n_px > 0 && a->n_cam2surf > 0 && a->n_bins > 0))
    return OBJNERF_EINVAL;
  const int n = a->n_frames * a->n_px;
  hipStream_t st = (hipStream_t)stream;
  objnerf_sample_args b = *a;
  b.kf_meta = nullptr;                    // (the random stream is a->obj_index; there is no keyframe draw here)
  float* origins_ws = b.out_origins ? b.out_origins : b.max_depth_ws + 1;
  float* dirs_ws = b.out_origins ? b.out_dirs : b.max_depth_ws + 1 + (size_t)n * 3;
  (void)hipMemsetAsync(b.max_depth_ws, 0, sizeof(float), st);
  hipLaunchKernelGGL(sample_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b, sampled_rgbs, sampled_depth, origins,
                     dirs_w, origins_ws, dirs_ws);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(sample_place_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b, (const objnerf_kf_store*)nullptr);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_ingest_frame(int32_t W, int32_t H, const uint8_t* rgb, const float* depth, const int32_t* inst,
                         const float* t_wc, int32_t K, const objnerf_ingest_item* items, void* stream) {
  CLEAR_STALE();
  if (W <= 0 || H <= 0 || K <= 0 || K > 65535 || !rgb || !depth || !inst || !t_wc || !items) return OBJNERF_EINVAL;
  const long npx = (long)W * H;
  hipLaunchKernelGGL(ingest_frame_kernel, dim3((unsigned)((npx + 255) / 256), K), dim3(256), 0, (hipStream_t)stream, W, H,
                     rgb, depth, inst, t_wc, items);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_sample_rays_stacked(const objnerf_sample_args* a, int32_t K, const objnerf_kf_store* table, void* stream) {
  CLEAR_STALE();
  if (!a || !table || K <= 0 || K > 65535 || !sample_args_ok(a)) return OBJNERF_EINVAL;
  const int n = a->n_frames * a->n_px;
  hipStream_t st = (hipStream_t)stream;
  // the K depth maxima sit (1 + 6 n) floats apart: clearing the whole scratch is one call
  (void)hipMemsetAsync(a->max_depth_ws, 0, (size_t)K * (1 + 6 * (size_t)n) * sizeof(float), st);
  const dim3 grid((n + 255) / 256, K);
  hipLaunchKernelGGL(sample_gather_kernel, grid, dim3(256), 0, st, *a, table);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(sample_place_kernel, grid, dim3(256), 0, st, *a, table);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

}  // extern "C"
