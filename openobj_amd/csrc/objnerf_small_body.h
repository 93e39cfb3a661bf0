// The small-batch iteration of the hidden-128 network as ONE forward + loss + backward launch (included by
// objnerf_generic.hip inside namespace objgen, after mlp_fwd_small_kernel / mlp_bwd_small_kernel whose tiling it keeps).
//
// Round 4's chain for the background network (train.py:447-463 at 1200 rays x 14 .. 64 samples) was embed -> forward ->
// loss prologue -> loss -> loss total -> heads backward -> head weight gradients -> backward -> PE backward -> grouped
// weight gradients -> reductions -> AdamW, i.e. twelve dependent launches and, per sample, the round trips of the
// embedding (read twice), alpha / colour, d alpha / d colour, dhead, h1 .. h4 as ReLU masks and d_emb through HBM.
// Here a workgroup owns WHOLE RAYS (rpw rays = up to 16 RT samples), so the compositing (render_rays.py:32-63), the
// four loss terms (loss.py:27-101), the head gradients and the positional encoding's backward happen between its
// forward and its backward pass on data that is still in LDS / registers:
//
//   points -> embedding (embedding.py:46-55; registers + HBM for the weight-gradient GEMMs)
//   -> h1 .. hc (model.py:61-79; LDS ping-pong, every layer's weights streamed L2 -> LDS; HBM copy for the weight
//      gradients; ReLU branch bits kept in registers)
//   -> heads (model.py:81-96) -> a wave per ray: occupancy, transmittance scan, depth / colour / opacity, variance,
//      loss terms, d alpha / d colour (the arithmetic of loss_kernel, objnerf_misc.hip)
//   -> head weight-gradient partials (per workgroup, from h4 / hc still in LDS) -> d_hc
//   -> d_h4 .. d_h1 (LDS ping-pong, HBM copies for the weight gradients) with the embedding gradient kept in the
//      accumulators -> d B partial (embedding.py:48-52) per workgroup.
//
// What is left for other launches: the seven weight-gradient GEMMs (one grouped launch) and ONE reduction launch that
// also sums the loss terms, writes the status word and applies AdamW (reduce_parts_kernel).  No zero fills, no atomics:
// every partial is written once and summed in a fixed order (bit-reproducible).
//
// The label statistics (counts [K][2], flags [2]) are inputs, or -- OBJNERF_TRAIN_SELF_COUNTS with ONE object, the
// background network -- counted by every workgroup from the R label bytes (1.2 KB) and published by workgroup 0.
#ifndef SM_ABL
#define SM_ABL 0      // diagnostic builds (tools/small_ablation.sh): bit 0 no HBM activation stores, 1 no sin / cos, 2 no
#endif                // MFMAs, 3 no heads / compositing, 4 no weight staging, 5 no PE backward, 6 no weight LOADS (the LDS
                      // images are still written, from registers that were never filled), 7 no LDS image WRITES (the loads
                      // are still issued and waited for) -- results are then wrong
struct SmallFused {
  int K, R, S, rpw;                       // rpw: rays per workgroup (rpw * S <= 16 RT rows)
  const float* params; long ps; const float* scale;
  const float* pts; const float* origins; const float* dirs; const float* z; float centre;
  const float* gt_depth; const float* gt_rgb; const uint8_t* labels;
  int* counts; int* flags; int self_counts;
  float cs, os;
  float* emb;                             // [K][n][129]
  float *h1, *h2, *h3, *h4, *hc;          // [K][n][128]
  float *d_hc, *d_h4, *d_h3, *d_h2, *d_h1;
  float *partA, *partW, *rsA, *rsW;       // head weight-gradient partials, reduce_parts_kernel layout [K][nwg][M][N]
  float* pe_part;                         // [K][nwg][63]
  float* loss_part;                       // [K][nwg][4]
  int o_in_w, o_in_b, o_m1_w, o_m1_b, o_cat_w, o_cat_b, o_m2_w, o_m2_b, o_a_w, o_a_b, o_cl_w, o_cl_b, o_oc_w, o_oc_b, o_B;
  // feature-distillation branch (FEAT; the 512-d head hoisted past the compositing): inputs from the
  // preparation launches, outputs for the weight-gradient launches
  float fs; int o_fl_w, o_fl_b;
  const float* rayin;                     // [K][R][H + 2]  u = W_of^T g, beta = b_of . g, |g|
  const float* gram;                      // [K][H H + H + 1]  G = W_of^T W_of (symmetric), wb = W_of^T b_of, bb
  float *hf, *d_hf;                       // [K][n][128]  feature hidden / its pre-activation gradient
  float* rayfeat;                         // [K][R][H + 3]  fh, O, a, c
  float *X1, *X2;                         // [K][R][H + 1]  [a fh | a O], [c fh | c O]  (featg_scale_kernel's outputs)
};
constexpr int SF_SMALL = 1536;            // floats of small LDS arrays (below)
constexpr int SF_FEAT_RT4 = 5632, SF_FEAT_RT5 = 1408;      // extra small arrays of the feature branch (floats): RT 4 / RT 5
template <int RT, bool FEAT = false> constexpr size_t sf_lds_bytes() {
  return (size_t)((FS_H + 2 * 16 * RT) * FS_P + SF_SMALL + (FEAT ? (RT == 5 ? SF_FEAT_RT5 : SF_FEAT_RT4) : 0)) * sizeof(float);
}
// rays per workgroup: as many whole rays as 80 rows hold (16 RT <= 80); with the feature branch the composited hidden and
// its gradient (2 x 128 floats per ray) live in LDS too -- at most 5 rays beside an 80-row tile, 20 beside a 64-row one
__host__ __device__ inline int sf_rays_per_wg(int S, bool feat) {
  int rpw = 80 / S;
  if (feat && rpw > 5) rpw = 64 / S;
  return rpw;
}

__device__ __forceinline__ float sf_sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

template <int RT, bool BF, bool FEAT>
__global__ __launch_bounds__(512) void train_small_kernel(const SmallFused a) {
  constexpr int BM = 16 * RT, H = FS_H, PT = FS_P, PW = BS_PW;
  constexpr int MAXRAY = RT == 5 ? 5 : 20;         // rays per workgroup with the feature branch (sf_rays_per_wg)
  static_assert(!FEAT || (BM + 2 * MAXRAY * FS_H + 4 * MAXRAY <= (RT == 5 ? SF_FEAT_RT5 : SF_FEAT_RT4)), "feature LDS arrays");
  static_assert(BM * 12 + 32 + 128 * 3 + 48 * 3 + 16 <= SF_SMALL, "small LDS arrays");
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  float* Wb = fs_lds;                   // forward: [out][in] pitch 132; backward: [k = out][n = in] pitch 130
  float* Xa = Wb + H * PT;              // [BM][132]
  float* Xb = Xa + BM * PT;
  float* s_t = Xb + BM * PT;            // [BM][3]  p / scale
  float* s_alpha = s_t + BM * 3;        // [BM]
  float* s_col = s_alpha + BM;          // [BM][3]
  float* s_dh = s_col + BM * 3;         // [BM][4]  dhead = (10 d alpha, d colour * c (1 - c))
  float* s_red = s_dh + BM * 4;         // [8][4]   loss terms of the waves
  float* s_pe1 = s_red + 32;            // [128][3] d B contributions per x1 column
  float* s_pe2 = s_pe1 + 128 * 3;       // [48][3]  ... per x2 column
  int* s_cnt = reinterpret_cast<int*>(s_pe2 + 48 * 3);      // [16]
  // feature branch
  float* s_wt = reinterpret_cast<float*>(s_cnt + 16);       // [BM]           ray weight of every sample; later d loss / d weight (feature part)
  float* s_fh = s_wt + BM;                                   // [MAXRAY][128]  composited feature hidden
  float* s_dfh = s_fh + MAXRAY * H;                          // [MAXRAY][128]  d loss / d fh
  float* s_rayf = s_dfh + MAXRAY * H;                        // [MAXRAY][4]    O, m1, gof of the ray
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, gg = lane >> 4;
  const long z = blockIdx.y;
  const int S = a.S;
  const long n = (long)a.R * S;
  const int ray0 = blockIdx.x * a.rpw;
  const int nr = min(a.rpw, a.R - ray0);
  const int rows = nr * S;
  const long m0 = (long)ray0 * S;
  const long zb = z * gridDim.x + blockIdx.x;
  const float* P = a.params + z * a.ps;
  const int kk = tid & 127, rg = tid >> 7;
  const int fb = 16 * w + c;              // this lane's feature column in every MFMA result

  // ---- label statistics
  if (a.self_counts) {                    // (one object: the host enables this for K == 1 only)
    int c1 = 0, c2 = 0;
    for (int r = tid; r < a.R; r += 512) {
      const int l = a.labels[z * a.R + r];
      c1 += (l == 1); c2 += (l != 2);
    }
    for (int d = 32; d >= 1; d >>= 1) { c1 += __shfl_xor(c1, d, 64); c2 += __shfl_xor(c2, d, 64); }
    if (lane == 0) { s_cnt[2 * w] = c1; s_cnt[2 * w + 1] = c2; }
  }
  // ---- scaled points of the workgroup's rows
  if (tid < BM * 3) {
    const int m = tid / 3, x = tid - 3 * m;
    float t = 0.f;
    if (m < rows) {
      const long i = z * n + m0 + m;
      float p;
      if (a.pts) p = a.pts[i * 3 + x];
      else {
        const long ray = z * a.R + ray0 + m / S;
        p = (a.origins[ray * 3 + x] + a.dirs[ray * 3 + x] * a.z[i]) - a.centre;      // vmap.py:548-551: two roundings
      }
      t = p / a.scale[z];
    }
    s_t[tid] = t;
  }
  for (int i = tid; i < BM * 4; i += 512) s_dh[i] = 0.f;
  // (per-lane constants -- biases, head weights, encoding directions -- are loaded one stage before their use instead of
  // at the top: held for the whole kernel they cost ~30 registers and the first build spilled 200+)
  const float* Bp = P + a.o_B;

  float wr[32];
  if (SM_ABL & 64) {
#pragma unroll
    for (int i = 0; i < 32; ++i) wr[i] = 0.f;
  }
  auto fetch_w = [&](const float* W, int ld, const int KC) {
    if (SM_ABL & 16) return;
    if (SM_ABL & 64) return;
    // (the row offsets are re-formed per call from an OPAQUE stride: left visible, the compiler hoists the 32 offsets of
    // every distinct stride to the top of the kernel and keeps ~100 registers of addresses alive -- 80+ spills)
    asm volatile("" : "+s"(ld));
    const float* p = W + rg * ld + kk;
    const long step = 4l * ld;
#pragma unroll
    for (int i = 0; i < 32; ++i) { wr[i] = kk < KC ? *p : 0.f; p += step; }
  };
  auto put_wf = [&]() {                   // forward image
    if (SM_ABL & 16) return;
    if (SM_ABL & 128) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" :: "v"(wr[i]));
      return;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) Wb[(rg + 4 * i) * PT + kk] = wr[i];
  };
  constexpr int PW16 = 288;
  auto put_wb = [&]() {                   // backward image (bf16 mode: rounded rows, 288-byte pitch)
    if (SM_ABL & 16) return;
    if (SM_ABL & 128) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" :: "v"(wr[i]));
      return;
    }
    if (BF) {
      __bf16* W16 = reinterpret_cast<__bf16*>(Wb);
#pragma unroll
      for (int i = 0; i < 32; ++i) W16[(rg + 4 * i) * (PW16 / 2) + kk] = (__bf16)wr[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) Wb[(rg + 4 * i) * PW + kk] = wr[i];
  };
  fetch_w(P + a.o_in_w, OBJ_E1, OBJ_E1);
  __syncthreads();                        // s_t, s_cnt
  int n1i, n2i, fl0, fl1;
  if (a.self_counts) {
    n1i = 0; n2i = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { n1i += s_cnt[2 * q]; n2i += s_cnt[2 * q + 1]; }
    fl0 = n1i == 0; fl1 = n2i == 0;
    if (blockIdx.x == 0 && tid == 0) {
      a.counts[2 * z] = n1i; a.counts[2 * z + 1] = n2i;
      a.flags[0] = fl0; a.flags[1] = fl1;
    }
  } else {
    n1i = a.counts[2 * z]; n2i = a.counts[2 * z + 1];
    fl0 = a.flags[0]; fl1 = a.flags[1];
  }

  // ---- embedding of the rows this thread stages (row rg + 4 i, column kk): embedding.py:46-55 as embed_kernel.
  // x1 = columns [0, 87) now (layer 1 and the cat layer read it); x2 = [87, 129) right before the colour layer.
  constexpr int ER = (BM + 3) / 4;
  float er[ER];
  float* eo = a.emb + (z * n + m0) * OBJ_EMB;
  {
    const int e1 = kk - 3, fo1 = e1 >= 0 ? e1 / OBJ_NDIR : 0, j1 = e1 >= 0 ? e1 - fo1 * OBJ_NDIR : 0;
    const float B10 = Bp[3 * j1], B11 = Bp[3 * j1 + 1], B12 = Bp[3 * j1 + 2];
    const float sf1 = (float)(1 << (fo1 & 7));
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int m = rg + 4 * i;
      float v1 = 0.f;
      if (m < BM) {
        const float t0 = s_t[3 * m], t1 = s_t[3 * m + 1], t2 = s_t[3 * m + 2];
        if (kk < 3) v1 = kk == 0 ? t0 : (kk == 1 ? t1 : t2);
        else if (kk < OBJ_E1) v1 = (SM_ABL & 2) ? fmaf(t2, B12, t0 * B10) : sin_acc((fmaf(t2, B12, fmaf(t1, B11, t0 * B10)) * sf1) * OBJ_PI_F);
        if (m < rows && kk < OBJ_E1) eo[(long)m * OBJ_EMB + kk] = v1;
      }
      er[i] = v1;
    }
  }
  auto embed_x2 = [&]() {                 // er <- x2 (octaves 4, 5)
    const int fo2 = 4 + kk / OBJ_NDIR, j2 = kk % OBJ_NDIR;
    const float B20 = Bp[3 * j2], B21 = Bp[3 * j2 + 1], B22 = Bp[3 * j2 + 2];
    const float sf2 = (float)(1 << (fo2 & 7));
#pragma unroll
    for (int i = 0; i < ER; ++i) {
      const int m = rg + 4 * i;
      float v2 = 0.f;
      if (m < BM && kk < OBJ_E2) {
        const float t0 = s_t[3 * m], t1 = s_t[3 * m + 1], t2 = s_t[3 * m + 2];
        v2 = (SM_ABL & 2) ? fmaf(t2, B22, t0 * B20) : sin_acc((fmaf(t2, B22, fmaf(t1, B21, t0 * B20)) * sf2) * OBJ_PI_F);
        if (m < rows) eo[(long)m * OBJ_EMB + OBJ_E1 + kk] = v2;
      }
      er[i] = v2;
    }
  };
  auto put_emb = [&](float* X, const int KC) {
    const int KC16 = BF ? ((KC + 31) & ~31) : ((KC + 15) & ~15);
    if (kk < KC16) {
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const int m = rg + 4 * i;
        if (m < BM) X[m * PT + kk] = er[i];
      }
    }
  };
  f32x4 acc[RT], accE[RT];
  auto zero = [&](f32x4 (&v)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // forward contraction: acc += X[:, :KC] Wb^T for this wave's 16 output features (mlp_fwd_small_kernel's loop)
  auto mma_f = [&](const float* X, const int KC) {
    if (SM_ABL & 4) return;
    if (BF) {
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
  // HBM copies of the activations / pre-activation gradients (operands of the grouped weight-gradient launch).  Round 6,
  // bf16 mode: stored ALREADY ROUNDED, two bytes per element (the buffers keep their fp32 spacing; element e of object z's
  // tensor sits at 16-bit index z n H + e) -- gemm_group16_kernel rounded every fp32 element it staged with the same
  // conversion, so its results are bit-identical and both sides of the round trip move half the bytes.
  auto hst = [&](float* base, const long e, const float v) {
    if (BF) reinterpret_cast<__bf16*>(base)[e] = (__bf16)v; else base[e] = v;
  };
  // relu(acc + bias) -> LDS buffer, the HBM activation (weight-gradient operand) and the branch bits (bit 4 i + r)
  auto store = [&](float* X, float* hbm, const float bv) -> unsigned {
    const long out = (z * n + m0) * H + fb;
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[i][r] = fmaxf(acc[i][r] + bv, 0.f);
        bits |= (acc[i][r] > 0.f ? 1u : 0u) << (4 * i + r);
      }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(16 * i + 4 * gg + r) * PT + fb] = acc[i][r];
    if (SM_ABL & 1) return bits;
    if (16 * RT <= rows) {                  // (a full tile: no per-row test)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) hst(hbm, out + (long)(16 * i + 4 * gg + r) * H, acc[i][r]);
    } else {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * gg + r;
          if (m < rows) hst(hbm, out + (long)m * H, acc[i][r]);
        }
    }
    return bits;
  };
  constexpr int E1P = BF ? ((OBJ_E1 + 31) & ~31) : ((OBJ_E1 + 15) & ~15), E2P = BF ? ((OBJ_E2 + 31) & ~31) : ((OBJ_E2 + 15) & ~15);
  // ---- h1 = relu(x1 W_in^T + b)                                  (model.py:63-66)
  put_emb(Xa, OBJ_E1);
  put_wf();
  __syncthreads();
  fetch_w(P + a.o_m1_w, H, H);
  float bv = P[a.o_in_b + fb];
  zero(acc);
  mma_f(Xa, E1P);
  __syncthreads();
  const unsigned mk1 = store(Xb, a.h1, bv);
  put_wf();
  __syncthreads();
  // ---- h2
  fetch_w(P + a.o_cat_w, H + OBJ_E1, H);
  bv = P[a.o_m1_b + fb];
  zero(acc);
  mma_f(Xb, H);
  __syncthreads();
  const unsigned mk2 = store(Xa, a.h2, bv);
  put_wf();
  put_emb(Xb, OBJ_E1);
  __syncthreads();
  // ---- h3 = relu([h2 | x1] W_cat^T + b)
  fetch_w(P + a.o_cat_w + H, H + OBJ_E1, OBJ_E1);
  zero(acc);
  mma_f(Xa, H);
  __syncthreads();
  put_wf();
  __syncthreads();
  fetch_w(P + a.o_m2_w, H, H);
  bv = P[a.o_cat_b + fb];
  mma_f(Xb, E1P);
  __syncthreads();
  const unsigned mk3 = store(Xa, a.h3, bv);
  put_wf();
  __syncthreads();
  // ---- h4
  fetch_w(P + a.o_cl_w, H + OBJ_E2, H);
  bv = P[a.o_m2_b + fb];
  zero(acc);
  mma_f(Xa, H);
  embed_x2();
  __syncthreads();
  const unsigned mk4 = store(Xb, a.h4, bv);
  put_wf();
  put_emb(Xa, OBJ_E2);
  __syncthreads();
  // ---- hc = relu([h4 | x2] W_cl^T + b)
  fetch_w(P + a.o_cl_w + H, H + OBJ_E2, OBJ_E2);
  zero(acc);
  mma_f(Xb, H);
  __syncthreads();
  put_wf();
  __syncthreads();
  fetch_w(P + a.o_cl_w, H + OBJ_E2, H);        // (already the backward's first weight block: d_h4 = d_hc W_cl[:, :H])
  bv = P[a.o_cl_b + fb];
  const float wa0 = P[a.o_a_w + lane], wa1 = P[a.o_a_w + lane + 64];
  const float w00 = P[a.o_oc_w + lane], w01 = P[a.o_oc_w + lane + 64];
  const float w10 = P[a.o_oc_w + H + lane], w11 = P[a.o_oc_w + H + lane + 64];
  const float w20 = P[a.o_oc_w + 2 * H + lane], w21 = P[a.o_oc_w + 2 * H + lane + 64];
  const float ba = P[a.o_a_b], bc0 = P[a.o_oc_b], bc1 = P[a.o_oc_b + 1], bc2 = P[a.o_oc_b + 2];
  mma_f(Xa, E2P);
  __syncthreads();
  float* Xc = Wb;                              // hc: the weight buffer is free until the backward's first put_wb
  const unsigned mkc = store(Xc, a.hc, bv);
  __syncthreads();
  // ---- heads: alpha = 10 (h4 . wa + ba), colour = sigmoid(hc Woc^T + boc)      (model.py:81-96); a wave per row
  for (int m = w; m < ((SM_ABL & 8) ? 0 : rows); m += 8) {
    const float x0 = Xb[m * PT + lane], x1 = Xb[m * PT + lane + 64];
    const float y0 = Xc[m * PT + lane], y1 = Xc[m * PT + lane + 64];
    const float sa = wave_sum64(fmaf(wa1, x1, wa0 * x0));
    const float s0 = wave_sum64(fmaf(w01, y1, w00 * y0));
    const float s1 = wave_sum64(fmaf(w11, y1, w10 * y0));
    const float s2 = wave_sum64(fmaf(w21, y1, w20 * y0));
    if (lane == 0) {
      s_alpha[m] = (sa + ba) * 10.0f;
      s_col[3 * m] = sigmoid_acc(s0 + bc0);
      s_col[3 * m + 1] = sigmoid_acc(s1 + bc1);
      s_col[3 * m + 2] = sigmoid_acc(s2 + bc2);
    }
  }
  const float wa_f = P[a.o_a_w + fb];                                   // backward: column fb of the two heads
  const float woc0_f = P[a.o_oc_w + fb], woc1_f = P[a.o_oc_w + H + fb], woc2_f = P[a.o_oc_w + 2 * H + fb];
  __syncthreads();
  // ---- a wave per ray: occupancy_activation / occupancy_to_termination / render (render_rays.py:6-63), the loss terms
  // of loss.py:27-79 and their gradients -- loss_kernel's arithmetic (objnerf_misc.hip) on one 64-lane chunk (S <= 64).
  // With the feature branch the pass runs twice: FIRST without the feature term's d loss / d weight (it yields the ray
  // weights, the loss terms, the rays' opacity and the colour head's gradients, none of which depend on it), and again
  // once that term is known (s_wt then holds it), for d alpha alone.
  const float inv1 = fl0 ? 0.0f : 1.0f / ((float)n1i + 1e-10f);
  const float inv2 = fl1 ? 0.0f : 1.0f / ((float)n2i + 1e-10f);
  float lt0 = 0.f, lt1 = 0.f, lt2 = 0.f, lt3 = 0.f;
  auto composite = [&](const bool first, const bool with_dwv) {
    const bool on = lane < S;
    for (int lr = w; lr < ((SM_ABL & 8) ? 0 : nr); lr += 8) {
      const long rr = z * a.R + ray0 + lr;
      const int m = lr * S + (on ? lane : 0);
      const int lab = a.labels[rr];
      const float m1 = (lab == 1) ? 1.f : 0.f, m2 = (lab != 2) ? 1.f : 0.f, tgt = (lab != 0) ? 1.f : 0.f;
      const float occ = on ? sigmoid_acc(s_alpha[m]) : 0.f;
      const float fr = on ? (1.0f - occ) + 1e-10f : 1.0f;
      const float pinc = SegRows::make(64, lane).scan_mul(fr, lane);
      float T = __shfl_up(pinc, 1, 64);
      if (lane == 0) T = 1.0f;
      const float wt = occ * T;
      const float c0 = s_col[3 * m], c1 = s_col[3 * m + 1], c2 = s_col[3 * m + 2];
      const float zz = on ? a.z[z * n + m0 + m] : 0.f;
      const float D = wave_sum64(on ? wt * zz : 0.f);
      const float O = wave_sum64(wt);
      const float C0 = wave_sum64(on ? wt * c0 : 0.f);
      const float C1 = wave_sum64(on ? wt * c1 : 0.f);
      const float C2 = wave_sum64(on ? wt * c2 : 0.f);
      const float dz = on ? zz - D : 0.f;
      const float Vv = wave_sum64(on ? wt * (dz * dz) : 0.f);
      const float info = 1.0f / (sqrtf(Vv) + 1e-4f);
      const float rd = D - a.gt_depth[rr];
      const float r0 = C0 - a.gt_rgb[rr * 3], r1 = C1 - a.gt_rgb[rr * 3 + 1], r2 = C2 - a.gt_rgb[rr * 3 + 2];
      const float ro = O - tgt;
      const float gD = m1 * sf_sgn(rd) * info * inv1;
      const float gC0 = a.cs * m1 * sf_sgn(r0) * inv1, gC1 = a.cs * m1 * sf_sgn(r1) * inv1, gC2 = a.cs * m1 * sf_sgn(r2) * inv1;
      const float gO = a.os * m2 * sf_sgn(ro) * inv2;
      if (first) {
        lt0 += m1 * fabsf(rd) * info * inv1;
        lt1 += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
        lt2 += m2 * fabsf(ro) * inv2;
      }
      const float dwf = (with_dwv && on) ? s_wt[m] : 0.0f;        // (second pass: the feature term's d loss / d weight)
      const float dw = on ? gD * zz + gO + gC0 * c0 + gC1 * c1 + gC2 * c2 + dwf : 0.f;
      const float qv = dw * wt;
      const float inc = SegRows::make(64, lane).rscan_add(qv, lane);
      const float suf = inc - qv;
      if (on) {
        const float docc = dw * T - suf / fr;
        const float da = docc * occ * (1.0f - occ);
        float* dh = s_dh + 4 * m;                 // heads_bwd_kernel: dhead = (10 d_alpha, d_colour * c (1 - c))
        dh[0] = 10.0f * da;
        if (first) {
          dh[1] = (gC0 * wt) * c0 * (1.0f - c0);
          dh[2] = (gC1 * wt) * c1 * (1.0f - c1);
          dh[3] = (gC2 * wt) * c2 * (1.0f - c2);
        }
      }
      if (FEAT && first) {
        if (on) s_wt[m] = wt;
        if (lane == 0) { s_rayf[4 * lr] = O; s_rayf[4 * lr + 1] = m1; }
      }
    }
  };
  // head weight-gradient partials of this workgroup (head_wgrad_kernel's sums over the rows in LDS): d wa[f] = sum_m
  // dhead[m][0] h4[m][f] (q = 0), d Woc[x][f] = sum_m dhead[m][1 + x] hc[m][f] (q = 1..3); biases = column sums of dhead
  auto head_partials = [&](const int q_lo, const int q_hi) {
    const int q = tid >> 7;                    // 0: alpha head (h4 = Xb), 1..3: colour head rows (hc = Xc)
    if (q >= q_lo && q <= q_hi) {
      const float* X = q == 0 ? Xb : Xc;
      float sacc = 0.f;
      for (int m = 0; m < rows; ++m) sacc = fmaf(s_dh[4 * m + q], X[m * PT + kk], sacc);
      if (q == 0) a.partA[zb * H + kk] = sacc;
      else a.partW[(zb * 3 + (q - 1)) * H + kk] = sacc;
    }
    if (tid < 4 && tid >= q_lo && tid <= q_hi) {
      float b = 0.f;
      for (int m = 0; m < rows; ++m) b += s_dh[4 * m + tid];
      if (tid == 0) a.rsA[zb] = b; else a.rsW[zb * 3 + tid - 1] = b;
    }
  };
  composite(true, false);
  unsigned mkf = 0;
  if constexpr (!FEAT) {
    __syncthreads();
    head_partials(0, 3);
  } else {
    // ================= feature branch: hf = relu([h4 | x2] W_fl^T + b) (model.py:98-101), fh = sum_s w_s hf_s per ray, the
    // cosine term in Gram form (loss.py:81-99 through DESIGN.md 4.3: loss_kernel's hoisted arithmetic), d loss / d fh ->
    // d hf and the feature part of d loss / d weight
    __syncthreads();                             // dhead's colour columns, s_wt, s_rayf
    head_partials(1, 3);                         // (the colour head's weights need hc, which the next staging overwrites)
    fetch_w(P + a.o_fl_w, H + OBJ_E2, H);
    __syncthreads();
    put_wf();
    __syncthreads();
    fetch_w(P + a.o_fl_w + H, H + OBJ_E2, OBJ_E2);
    bv = P[a.o_fl_b + fb];
    zero(acc);
    mma_f(Xb, H);
    __syncthreads();
    put_wf();
    __syncthreads();
    // (the object's Gram matrix G = W_of^T W_of, 64 KB, takes the weight buffer next: the feature layer's weights are
    // dead after this contraction and the backward's first block is staged after the loss)
    const float* Gk = a.gram + z * ((long)H * H + H + 1);
    fetch_w(Gk, H, H);
    mma_f(Xa, E2P);
    __syncthreads();
    mkf = store(Xa, a.hf, bv);                   // hf replaces x2 in LDS (its ReLU mask is the sign of these values)
    (void)mkf;
    put_wf();                                    // Wb <- G
    __syncthreads();
    fetch_w(P + a.o_fl_w, H + OBJ_E2, H);       // (the backward's first weight block)
    // fh[ray][f] = sum_s w_s hf[s][f]      (thread (f, q) takes rays q, q + 4, ..; s ascending as loss_kernel)
    for (int lr = rg; lr < nr; lr += 4) {
      float f = 0.f;
      for (int si = 0; si < S; ++si) f = fmaf(s_wt[lr * S + si], Xa[(lr * S + si) * PT + kk], f);
      s_fh[lr * H + kk] = f;
    }
    __syncthreads();
    {
      const float* wbv = Gk + (long)H * H;
      const float bb = wbv[H];
      for (int lr = w; lr < nr; lr += 8) {       // a wave per ray; lane owns entries lane, lane + 64
        const long rr = z * a.R + ray0 + lr;
        const float* rin = a.rayin + rr * (H + 2);
        const float* fhr = s_fh + lr * H;
        // G fh from the LDS copy of G (rows lane, lane + 64; 16-byte reads, conflict-free at the 132-float pitch; the
        // first build read G's columns from L2 inside this loop: 128 load latencies per ray, +1.2 ms on configs[2])
        float gf0 = 0.f, gf1 = 0.f;
        const float* g0p = Wb + lane * PT;
        const float* g1p = Wb + (lane + 64) * PT;
#pragma unroll 8
        for (int h2 = 0; h2 < H; h2 += 4) {
          const f32x4 fv = *reinterpret_cast<const f32x4*>(fhr + h2);
          const f32x4 ga = *reinterpret_cast<const f32x4*>(g0p + h2), gb = *reinterpret_cast<const f32x4*>(g1p + h2);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gf0 = fmaf(ga[e], fv[e], gf0); gf1 = fmaf(gb[e], fv[e], gf1); }
        }
        const float f0 = fhr[lane], f1 = fhr[lane + 64];
        const float u0 = rin[lane], u1 = rin[lane + 64], wb0 = wbv[lane], wb1 = wbv[lane + 64];
        const float fu = wave_sum64(fmaf(f1, u1, f0 * u0));
        const float fGf = wave_sum64(fmaf(f1, gf1, f0 * gf0));
        const float fwb = wave_sum64(fmaf(f1, wb1, f0 * wb0));
        const float O = s_rayf[4 * lr], m1 = s_rayf[4 * lr + 1];
        const float beta = rin[H], ngv = rin[H + 1];
        const float dotFg = fu + O * beta;
        const float nF2 = fmaxf(fGf + 2.0f * O * fwb + O * O * bb, 0.0f);
        const float nF = fmaxf(sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
        const float cosv = dotFg / (nF * ngc);
        lt3 += m1 * (1.0f - cosv) * inv1;
        const float gam = -a.fs * m1 * inv1;                 // d total / d cos
        const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
        const float gof = ar * beta + cr * (fwb + O * bb);   // d total / d opacity (feature part)
        float* rf = a.rayfeat + rr * (H + 3);
        float* x1o = a.X1 + rr * (H + 1);
        float* x2o = a.X2 + rr * (H + 1);
        rf[lane] = f0; rf[lane + 64] = f1;
        x1o[lane] = ar * f0; x1o[lane + 64] = ar * f1;
        x2o[lane] = cr * f0; x2o[lane + 64] = cr * f1;
        if (lane == 0) {
          rf[H] = O; rf[H + 1] = ar; rf[H + 2] = cr;
          x1o[H] = ar * O; x2o[H] = cr * O;
          s_rayf[4 * lr + 2] = gof;
        }
        s_dfh[lr * H + lane] = ar * u0 + cr * (gf0 + O * wb0);   // d total / d fh
        s_dfh[lr * H + lane + 64] = ar * u1 + cr * (gf1 + O * wb1);
      }
    }
    __syncthreads();
    // per sample (a wave per row): the feature part of d loss / d weight = gof + d fh . hf, and hf -> d hf in place
    // (pre-activation gradient of the feature layer: relu'(hf) w_s d fh)
    {
      const long out = (z * n + m0) * H;
      for (int m = w; m < rows; m += 8) {
        const int lr = m / S;
        const float h0 = Xa[m * PT + lane], h1 = Xa[m * PT + lane + 64];
        const float d0 = s_dfh[lr * H + lane], d1 = s_dfh[lr * H + lane + 64];
        const float pp = wave_sum64(fmaf(d1, h1, d0 * h0));
        const float wt = s_wt[m];
        const float g0 = h0 > 0.0f ? wt * d0 : 0.0f, g1 = h1 > 0.0f ? wt * d1 : 0.0f;
        Xa[m * PT + lane] = g0; Xa[m * PT + lane + 64] = g1;
        hst(a.d_hf, out + (long)m * H + lane, g0); hst(a.d_hf, out + (long)m * H + lane + 64, g1);
        if (lane == 0) s_wt[m] = s_rayf[4 * lr + 2] + pp;
      }
      for (int m = rows + w; m < BM; m += 8) { Xa[m * PT + lane] = 0.f; Xa[m * PT + lane + 64] = 0.f; }
    }
    __syncthreads();
    composite(false, true);                      // d alpha with the feature term
    __syncthreads();
    head_partials(0, 0);
  }
  if (lane == 0) { s_red[4 * w] = lt0; s_red[4 * w + 1] = lt1; s_red[4 * w + 2] = lt2; s_red[4 * w + 3] = lt3; }
  __syncthreads();
  if (tid < 4) {
    float v = 0.f;
    for (int q = 0; q < 8; ++q) v += s_red[4 * q + tid];
    a.loss_part[zb * 4 + tid] = v;
  }
  // ---- d_hc = relu'(hc) Woc^T d_craw -> Da (the colour layer's backward operand) and HBM
  float* Da = Xa;
  float* Db = Xb;
  auto form_dhc = [&]() {
    const long out = (z * n + m0) * H;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        const float* dh = s_dh + 4 * m;
        float v = fmaf(woc2_f, dh[3], fmaf(woc1_f, dh[2], woc0_f * dh[1]));
        v = ((mkc >> (4 * i + r)) & 1u) ? v : 0.f;
        Da[m * PT + fb] = v;
        if (m < rows) hst(a.d_hc, out + (long)m * H + fb, v);
      }
  };
  if constexpr (!FEAT) form_dhc();
  __syncthreads();                             // Xb (h4) / Xc (hc) have been read; Da is complete
  // ================= backward (mlp_bwd_small_kernel's chain; masks from the branch bits, no d_emb round trip)
  auto mma_b = [&](const float* D, f32x4 (&v)[RT]) {
    if (SM_ABL & 4) return;
    if (BF) {
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
  // masked d_h: (acc [+ the alpha head's rank-1 part wa[f] dhead[m][0]]) where the forward's ReLU passed
  auto store_dh = [&](float* X, float* hbm, const unsigned bits, const bool add_alpha) {
    const long out = (z * n + m0) * H + fb;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * gg + r;
        float v = acc[i][r];
        if (add_alpha) v += wa_f * s_dh[4 * m];
        acc[i][r] = ((bits >> (4 * i + r)) & 1u) ? v : 0.f;
        X[m * PT + fb] = acc[i][r];
      }
    if (SM_ABL & 1) return;
    if (16 * RT <= rows) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) hst(hbm, out + (long)(16 * i + 4 * gg + r) * H, acc[i][r]);
    } else {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * gg + r;
          if (m < rows) hst(hbm, out + (long)m * H, acc[i][r]);
        }
    }
  };
  // d B contributions of the embedding-gradient columns in accE (embedding.py:48-52; pe_bwd_kernel's terms): this lane
  // holds d_emb[m][col0 + fb] for its rows; octave fo, direction j of that column
  auto pe_bwd = [&](float* s_pe, const int ncol, const int e_of_col0) {
    if (SM_ABL & 32) return;
    const int e = e_of_col0 + fb;              // index into the 126 sin entries (column - 3)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (fb < ncol && e >= 0) {
      const int fo = e / OBJ_NDIR, j = e - fo * OBJ_NDIR;
      const float b0 = Bp[3 * j], b1 = Bp[3 * j + 1], b2 = Bp[3 * j + 2];
      const float sf = (float)(1 << fo);
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * gg + r;
          const float t0 = s_t[3 * m], t1 = s_t[3 * m + 1], t2 = s_t[3 * m + 2];
          const float pj = fmaf(t2, b2, fmaf(t1, b1, t0 * b0));
          float sv, cv;
          if (SM_ABL & 2) { sv = pj; cv = pj * sf; } else
          sincos_acc((pj * sf) * OBJ_PI_F, sv, cv);
          const float dp = accE[i][r] * ((cv * OBJ_PI_F) * sf);
          a0 = fmaf(dp, t0, a0); a1 = fmaf(dp, t1, a1); a2 = fmaf(dp, t2, a2);
        }
    }
    // the four lane groups hold different rows of the same column: add them in a fixed order
    a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
    a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
    if (gg == 0 && fb < ncol) { s_pe[3 * fb] = a0; s_pe[3 * fb + 1] = a1; s_pe[3 * fb + 2] = a2; }
  };
  const bool e1_wave = 16 * w < OBJ_E1, e2_wave = 16 * w < OBJ_E2;      // waves that own embedding columns
  zero(acc);
  zero(accE);
  if constexpr (FEAT) {
    // ---- feature layer first (Da = d_hf): d_h4 += d_hf W_fl[:, :H], d_x2 = d_hf W_fl[:, H:]; the colour layer
    // accumulates onto both
    put_wb();
    __syncthreads();
    fetch_w(P + a.o_fl_w + H, H + OBJ_E2, OBJ_E2);
    mma_b(Da, acc);
    __syncthreads();
    put_wb();
    __syncthreads();
    fetch_w(P + a.o_cl_w, H + OBJ_E2, H);
    if (e2_wave) mma_b(Da, accE);
    __syncthreads();                             // every wave has read d_hf
    form_dhc();                                  // Da <- d_hc
    __syncthreads();
  }
  // ---- colour layer: d_h4 = relu'(h4) (wa d_araw + [d_hf W_fl1] + d_hc W_cl[:, :H]), d_x2 (+)= d_hc W_cl[:, H:]
  put_wb();
  __syncthreads();
  fetch_w(P + a.o_cl_w + H, H + OBJ_E2, OBJ_E2);
  mma_b(Da, acc);
  __syncthreads();
  store_dh(Db, a.d_h4, mk4, true);               // Db = d_h4
  put_wb();
  __syncthreads();
  fetch_w(P + a.o_m2_w, H, H);
  if (e2_wave) { mma_b(Da, accE); pe_bwd(s_pe2, OBJ_E2, OBJ_E1 - 3); }
  __syncthreads();
  // ---- mid2: d_h3 = relu'(h3) (d_h4 W_m2)
  put_wb();
  __syncthreads();
  fetch_w(P + a.o_cat_w, H + OBJ_E1, H);
  zero(acc);
  mma_b(Db, acc);
  __syncthreads();
  store_dh(Da, a.d_h3, mk3, false);              // Da = d_h3
  put_wb();
  __syncthreads();
  // ---- cat layer: d_h2 = relu'(h2) (d_h3 W_cat[:, :H]), d_x1 = d_h3 W_cat[:, H:]
  fetch_w(P + a.o_cat_w + H, H + OBJ_E1, OBJ_E1);
  zero(acc);
  mma_b(Da, acc);
  __syncthreads();
  store_dh(Db, a.d_h2, mk2, false);              // Db = d_h2
  put_wb();
  __syncthreads();
  fetch_w(P + a.o_m1_w, H, H);
  zero(accE);
  if (e1_wave) mma_b(Da, accE);
  __syncthreads();
  // ---- mid1: d_h1 = relu'(h1) (d_h2 W_m1)
  put_wb();
  __syncthreads();
  fetch_w(P + a.o_in_w, OBJ_E1, OBJ_E1);
  zero(acc);
  mma_b(Db, acc);
  __syncthreads();
  store_dh(Da, a.d_h1, mk1, false);              // Da = d_h1
  put_wb();
  __syncthreads();
  // ---- in layer: d_x1 += d_h1 W_in; then d B
  if (e1_wave) { mma_b(Da, accE); pe_bwd(s_pe1, OBJ_E1, -3); }
  __syncthreads();
  if (tid < 63) {
    const int jj = tid / 3, x = tid - 3 * jj;
    float v = ((s_pe1[3 * (3 + jj) + x] + s_pe1[3 * (24 + jj) + x]) + s_pe1[3 * (45 + jj) + x]) + s_pe1[3 * (66 + jj) + x];
    v = (v + s_pe2[3 * jj + x]) + s_pe2[3 * (21 + jj) + x];
    a.pe_part[zb * 63 + tid] = v;
  }
}

template <int RT, bool FEAT>
static void launch_train_small(hipStream_t st, const SmallFused& f, int nwg, bool bf) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)train_small_kernel<RT, false, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sf_lds_bytes<RT, FEAT>());
    (void)hipFuncSetAttribute((const void*)train_small_kernel<RT, true, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sf_lds_bytes<RT, FEAT>());
  });
  dim3 grid((unsigned)nwg, (unsigned)f.K);
  constexpr size_t lds_bytes = sf_lds_bytes<RT, FEAT>();
  if (bf) hipLaunchKernelGGL((train_small_kernel<RT, true, FEAT>), grid, dim3(512), lds_bytes, st, f);
  else hipLaunchKernelGGL((train_small_kernel<RT, false, FEAT>), grid, dim3(512), lds_bytes, st, f);
}
