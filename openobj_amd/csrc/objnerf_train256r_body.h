// Kernel A of the hidden-256 path in its ROW-SPLIT form (round 5).  Included by objnerf_train256.hip inside namespace
// obj256: it shares the operand types, the index maps, the packed weight image (pack256_kernel), the workspace layout,
// the positional-encoding helpers, kernel B and finalize256_kernel with the first form (fwd256_kernel), and produces
// bit-for-bit the same kind of output (fragments of h1..h4, hc and the five pre-activation gradients, x1 / x2, head
// gradients, d B and loss partials).
//
// Why a second form.  fwd256_kernel gives each of the four waves 32 samples and ALL 256 output rows of a layer: the
// activations (B operand) stay in registers and every wave reads every weight fragment (A operand) from the LDS ring --
// 64 KB of LDS reads per 16-KB stage against 512 MFMA cycles, plus the LDS-DMA writes and the hand-off stores: the LDS
// pipe is as busy as the matrix core, and a stage ends in a workgroup barrier (83 per tile).  Here the roles are
// swapped: wave w owns the output rows 64 w .. 64 w + 63 (two 32-row blocks) of every hidden layer for ALL 128 samples
// of the tile (four column groups of 32).  Its weight fragments -- 2 x NK per layer, a quarter of the layer -- go from
// L2 straight into registers (no ring, no LDS-DMA, no stage schedule) one layer ahead; the activations of the tile live
// in LDS (two 64 KB buffers, layer l reads one and writes the other) and are the B operand, 16 bytes per lane and
// k-step, read once per wave and column group: 256 KB of LDS reads per layer instead of 512 + 128 + 64, and ONE barrier
// per layer.  A layer runs as four phases (one per column group, two row blocks x two accumulator chains = four
// independent MFMA chains); the epilogue of phase c - 1 (add the chains, convert, ReLU / mask, pack, LDS + workspace
// stores) rides in the MFMA shadow of phase c, and the last phase's epilogue in the shadow of the slot pass that follows
// where there is one.  The slot-shaped passes (alpha row, colour head, d x2 / d x1) keep the first form's split -- every
// wave its own 32 samples, all 1..3 row blocks -- because their results feed per-sample code (compositing, the
// embedding's chain rule) of the lane that owns the sample.
//
// LDS: activations 2 x 64 KB | x1 pieces [4][6] (24 KB; after the cat layer the same bytes hold x2 [4][3], the strips of
// the compositing and the head-gradient pieces) | bias tables 5 KB | embedding rows, d B sums.  The ReLU bits of a
// wave's own rows stay in 20 registers (the backward masks exactly the rows it produced).

constexpr int R_HBSZ = 4 * KS_H * PIECE;                       // one activation buffer: [column group][k-step] pieces
constexpr int R_XB = 2 * R_HBSZ;                               // x1 [4][KS_X1] pieces; after F3: x2 [4][KS_X2] pieces ...
constexpr int R_STRIP = R_XB + 4 * KS_X2 * PIECE;              // ... the strips (raw alpha | colour pre [3] | z; the ray targets sit beside the embedding rows)
constexpr int R_DH = R_STRIP + 3072;                           // ... and the head-gradient pieces [4]
constexpr int R_BIAS = R_XB + 4 * KS_X1 * PIECE;
constexpr int R_SMALL = R_BIAS + 5 * 256 * 4;
constexpr int R_DB = R_SMALL + 96 * 4;
constexpr int R_T = (R_DB + 4 * 68 * 4 + 15) & ~15;             // x / scale of the tile's 128 samples [3][128]: read back where the
constexpr int R_TOTAL = R_T + 3 * 128 * 4;                     // embedding's chain rule needs it (three registers through every layer otherwise)
static_assert(R_DH + 4 * PIECE <= R_BIAS && R_TOTAL <= 163840, "LDS budget of the row-split kernel");
static_assert((5 * 128 + 8 * 8) * 4 <= 3072, "strips");

// consumers of A fragments in consumption order; the successor of the last is the next tile's first
enum RC { C_F1, C_F2, C_F3, C_F4, C_F5, C_AL, C_F6, C_B6, C_B5H, C_B5X, C_B4, C_B3H, C_B3X, C_B2, C_B1, C_END };
__host__ __device__ constexpr int rc_q(int c) {
  return c == C_F1 ? F1 : c == C_F2 ? F2 : c == C_F3 ? F3 : c == C_F4 ? F4 : c == C_F5 ? F5 : c == C_AL ? F5 : c == C_F6 ? F6
       : c == C_B6 ? B6 : c == C_B5H ? B5H : c == C_B5X ? B5X : c == C_B4 ? B4 : c == C_B3H ? B3H : c == C_B3X ? B3X : c == C_B2 ? B2 : B1;
}
__host__ __device__ constexpr bool rc_slot(int c) { return c == C_AL || c == C_F6 || c == C_B5X || c == C_B3X || c == C_B1; }
__host__ __device__ constexpr int rc_nb(int c) { return rc_slot(c) ? (c == C_AL || c == C_F6 ? 1 : seq_nb(rc_q(c))) : 2; }
__host__ __device__ constexpr int rc_nk(int c) { return seq_nk(rc_q(c)); }
__host__ __device__ constexpr int rc_cnt(int c) { return rc_nb(c) * rc_nk(c); }
__host__ __device__ constexpr int rc_bmul(int c) { return rc_slot(c) ? 0 : 2; }          // first block = bmul * wave + badd
__host__ __device__ constexpr int rc_badd(int c) { return c == C_AL ? 8 : 0; }
// A fragment registers: 32 (two row blocks x 16 k-steps) + 12 for the k-steps past 16 of the two layers that take x
// slots (loaded inside the layer itself).  Register r of the 32 holds, per consumer:
//   layer            block r / 16 (of the wave's two), k-step r % 16
//   slot pass, 1 block (alpha: 19 k-steps, colour: 16)   k-step r
//   slot pass, 2 / 3 blocks   blocks 0 / 1 as a layer's; block 2 takes block 0's registers as they fall free
constexpr int R_NA = 32, R_NAX = 12;
__host__ __device__ constexpr int rc_nk16(int c) { return rc_nk(c) < 16 ? rc_nk(c) : 16; }
__host__ __device__ constexpr bool rc_reg_valid(int n, int r) {
  return n >= C_END ? false : rc_slot(n) && rc_nb(n) == 1 ? r < rc_nk(n) : (r & 15) < rc_nk16(n);
}
__host__ __device__ constexpr int rc_reg_b(int n, int r) { return rc_slot(n) && rc_nb(n) == 1 ? 0 : r >> 4; }
__host__ __device__ constexpr int rc_reg_ks(int n, int r) { return rc_slot(n) && rc_nb(n) == 1 ? r : r & 15; }
static_assert(2 * (rc_nk(C_F3) - 16) <= R_NAX && rc_nk(C_AL) <= R_NA, "A registers");

// The tile is ONE basic block of ~20 k instructions after unrolling; left alone the pre-RA scheduler pulls hundreds of
// LDS reads, bias rows and zero vectors to its top and the allocator spills them (2.4 KB of scratch per lane).  A fence
// behind every k-step keeps the prefetch distances the code spells out (B ring of four, A one consumer ahead).
#define R256_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef OBJ256_TIMING      // diagnostic build: cycles per part of a tile (s_memtime), printed by workgroup 0 / wave 0
#define RT(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); T256(i); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RSYNC() do { RT(0); __syncthreads(); RT(1); } while (0)
#define RTP(i) RT(i)
#else
#define RT(i) do {} while (0)
#define RTP(i) do {} while (0)
#define RSYNC() __syncthreads()
#endif


template <typename OT, int S>
__global__ __launch_bounds__(256) void fwdr256_kernel(const FwdArgs a) {
  typedef SQ<false> SQT;
  constexpr long IMG_BYTES = SQT::IMG_BYTES;
  constexpr int NTHR = 256, TSAMP = 128, NWAVE = 4;
  static_assert(TSAMP % S == 0, "whole rays per tile");
  typedef typename Op<OT>::V V;
  typedef __attribute__((address_space(1))) V GV;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int lane_k = lane;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = lane & 31;
  constexpr int TR = TSAMP / S;                   // rays per tile
  // (the same tile -> workgroup map as fwd256_kernel: an XCD streams at most two objects' weight images through its L2)
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const long T = (long)a.K * a.ntile;
  const long tau0 = T * wg / nwg, tau1 = T * (wg + 1) / nwg;
  if (tau0 >= tau1) return;

  float* s_bias = reinterpret_cast<float*>(lds + R_BIAS);
  float* s_raw = reinterpret_cast<float*>(lds + R_STRIP);
  float* s_col = s_raw + TSAMP;
  float* s_z = s_raw + 4 * TSAMP;
  float* s_ray = reinterpret_cast<float*>(lds + R_SMALL) + 68;      // per ray of the tile: gt depth, gt rgb, label (stride 5: 20 of the 28 spare floats)
  float* s_small = reinterpret_cast<float*>(lds + R_SMALL);
  float* s_db = reinterpret_cast<float*>(lds + R_DB);
  const uint32_t lane_off_k = (uint32_t)lane * 16u;
  uint32_t lane_off = lane_off_k;
  char* const hb0 = lds, * const hb1 = lds + R_HBSZ, * const xb = lds + R_XB, * const dhb = lds + R_DH;

  const float gs = a.grad_scale, inv_gs = 1.0f / a.grad_scale;
  T256_DECL;
  long tile_i; int k_i;
  { const long obj0 = tau0 / a.ntile; k_i = (int)obj0; tile_i = tau0 - obj0 * a.ntile; }
  int cur_obj = -1;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f;
  // d B of the object, one entry per LANE: lane s of half h keeps entry 33 h + s of the wave's [63] sums (direction
  // 11 h + dd, coordinate c at 3 dd + c; half 0's 33rd entry in dbl32) -- every contribution is summed over the half-wave
  // first (DPP row sums + one row swap), so two registers replace 33 per-lane accumulators
  float dbl = 0.0f, dbl32 = 0.0f;
  float scale = 1.0f, inv1 = 0.f, inv2 = 0.f, ba = 0.f, boc0 = 0.f, boc1 = 0.f, boc2 = 0.f;

  // wave-uniform floats as scalars (a uniform value that arrives through a vector load stays in a vector register otherwise)
  auto rfl = [](const float v) __attribute__((always_inline)) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  };
  auto flush_object = [&]() {       // partial d B and loss terms of (cur_obj, this workgroup)
    __syncthreads();
    float* pw = s_db + w * 68;
    {
      const int s_ = lane & 31, h_ = lane >> 5;
      if (s_ < 30 || h_ == 0) pw[33 * h_ + s_] = dbl;
      if (lane == 0) pw[32] = dbl32;
      dbl = 0.0f; dbl32 = 0.0f;
    }
    l_d = seg_sum<64>(l_d); l_c = seg_sum<64>(l_c); l_o = seg_sum<64>(l_o);
    if (lane == 0) { pw[64] = l_d; pw[65] = l_c; pw[66] = l_o; pw[67] = 0.0f; }
    __syncthreads();
    if (tid < 68) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) v += s_db[ww * 68 + tid];
      a.part[((long)cur_obj * NWG_A + wg) * PART_FLOATS + tid] = v;
    }
    __syncthreads();
  };

  auto load_point = [&](const int k_, const long tile_, float& px, float& py, float& pz, float& zv) __attribute__((always_inline)) {
    const int sidx = 32 * w + s, q_ = sidx / S, si_ = sidx - q_ * S;
    const long ray_ = tile_ * TR + q_;
    px = py = pz = zv = 0.f;
    if (ray_ < a.R) {
      const long rr = (long)k_ * a.R + ray_;
      zv = a.z[rr * S + si_];
      if (a.pts) {
        const float* p = a.pts + (rr * S + si_) * 3;
        px = p[0]; py = p[1]; pz = p[2];
      } else {
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        px = (o[0] + d[0] * zv) - a.obj_center;
        py = (o[1] + d[1] * zv) - a.obj_center;
        pz = (o[2] + d[2] * zv) - a.obj_center;
      }
    }
  };
  float npx, npy, npz, nzv;
  load_point(k_i, tile_i, npx, npy, npz, nzv);

  // ---- the A fragments of the consumer in progress and, as its registers fall free, of the next one
  V A[R_NA], AX[R_NAX];
  // this wave's first fragment of consumer c in image img -- ONE opaque scalar base per consumer and use: its fragments
  // are compile-time offsets from it (as one expression per fragment the compiler hoists ~300 64-bit offsets out of the
  // tile loop, spills them to vector lanes and reads them back lane by lane)
  auto cons_base = [&](auto c_tag, const char* img) __attribute__((always_inline)) -> const char* {
    constexpr int c = decltype(c_tag)::value;
    const char* p = img + ((long)(SQT::seq_off(rc_q(c)) + (rc_bmul(c) * w + rc_badd(c)) * rc_nk(c)) << 10);
    asm volatile("" : "+s"(p));
    return p;
  };
  auto frag_base = [&](auto c_tag, const int b, const int ks, const char* cbase) __attribute__((always_inline)) -> const char* {
    constexpr int c = decltype(c_tag)::value;
    return cbase + ((b * rc_nk(c) + ks) << 10);
  };
  auto frag_load = [&](V& dst, const char* base) __attribute__((always_inline)) { dst = *(const GV*)(base + lane_off); };
  // register r of the next consumer n (if n keeps a fragment there)
  auto loadR = [&](auto n_tag, const int r, const char* img) __attribute__((always_inline)) {
    constexpr int n = decltype(n_tag)::value;
    if (rc_reg_valid(n, r)) frag_load(A[r], frag_base(n_tag, rc_reg_b(n, r), rc_reg_ks(n, r), img));
  };
  {
    const char* cb0 = cons_base(std::integral_constant<int, C_F1>{}, (const char*)a.img + (long)k_i * IMG_BYTES);
#pragma unroll
    for (int r = 0; r < R_NA; ++r) loadR(std::integral_constant<int, C_F1>{}, r, cb0);
  }

  for (long tau = tau0; tau < tau1; ++tau) {
    // (opaque per tile: left loop-invariant, every fragment address -- a 64-bit vector pair each -- is hoisted out of the
    // tile loop and spilled; hidden, the bases stay scalar and the lane offset is the access's own offset operand)
    // (likewise everything derived from the lane index: ~60 hoisted addresses / predicates were spilled across the loop)
    int lane = lane_k;
    asm volatile("" : "+v"(lane));
    const int s = lane & 31, h = lane >> 5, tid = w * 64 + lane;
    lane_off = (uint32_t)lane * 16u;
    const int k = k_i;
    const long tile = tile_i;
    const bool live = tau + 1 < tau1;
    const long next_obj = (tile + 1 == a.ntile) ? (long)k + 1 : (long)k;
    if (k != cur_obj) {
      if (cur_obj >= 0) flush_object();
      cur_obj = k;
      const float* P = a.params + (long)k * a.p_stride;
      for (int i = tid; i < 5 * 256; i += NTHR) {          // bias tables [layer][blk][h][n]
        const int layer = i >> 8, rem = i & 255, blk = rem >> 5, hh = (rem >> 4) & 1, n = rem & 15;
        const int off = layer == 0 ? a.L.in_b : layer == 1 ? a.L.m1_b : layer == 2 ? a.L.cat_b : layer == 3 ? a.L.m2_b : a.L.cl_b;
        s_bias[i] = P[off + 32 * blk + acc_row(n, hh)];
      }
      if (tid < 63) s_small[tid] = P[a.L.pe_b + tid];
      if (tid == 64) s_small[64] = P[a.L.a_b];
      if (tid >= 65 && tid < 68) s_small[tid] = P[a.L.oc_b + tid - 65];
      for (int i = tid; i < NWAVE * 68; i += NTHR) s_db[i] = 0.0f;
      l_d = l_c = l_o = 0.f;
      scale = rfl(a.scale[k]);
      const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
      inv1 = rfl(a.flags[0] ? 0.0f : 1.0f / (n1 + 1e-10f));    // render_rays.py:89-94 early return / :103 mean
      inv2 = rfl(a.flags[1] ? 0.0f : 1.0f / (n2 + 1e-10f));
    }
    RSYNC();                                            // tables visible; previous tile done with every buffer
    ba = rfl(s_small[64]); boc0 = rfl(s_small[65]); boc1 = rfl(s_small[66]); boc2 = rfl(s_small[67]);

    const char* img_k = (const char*)a.img + (long)k * IMG_BYTES;
    const char* img_n = (const char*)a.img + (live ? next_obj : (long)k) * IMG_BYTES;
    char* ws_obj = a.ws + (long)k * a.wl.obj_bytes;
    const long sg_own = tile * 4 + w;                           // this wave's own sample group (column group w)
    // fragment (tensor, sample group of column group cg, k-step ks) in the workspace
    auto act_base = [&](const int tensor, const int cg, const int ks) __attribute__((always_inline)) -> GV* {
      uint32_t lo = lane_off;
      asm volatile("" : "+v"(lo));
      return (GV*)(ws_obj + (long)tensor * a.wl.act_stride + (((tile * 4 + cg) * KS_H + ks) << 10) + lo);
    };

    // ------------------------------------------------------------------ sample point of this lane (vmap.py:548-551)
    const int st_idx = 32 * w + s;
    const int q = st_idx / S, si = st_idx - q * S;
    const long ray = tile * TR + q;
    const bool valid = ray < a.R;
    const float px = npx, py = npy, pz = npz, z_own = nzv;      // (requested during the previous tile)
    float rg0 = 0.f, rg1 = 0.f, rg2 = 0.f, rg3 = 0.f; int rlab = 2;
    if (valid && si == 0) {
      const long rr = (long)k * a.R + ray;
      rg0 = a.gt_depth[rr]; rg1 = a.gt_rgb[rr * 3]; rg2 = a.gt_rgb[rr * 3 + 1]; rg3 = a.gt_rgb[rr * 3 + 2]; rlab = a.labels[rr];
    }
    const float t0 = px / scale, t1 = py / scale, t2 = pz / scale;       // embedding.py:47
    // x / scale of this lane's sample back from LDS (opaque address: see below)
    auto load_t = [&](float& u0, float& u1, float& u2) __attribute__((always_inline)) {
      uint32_t to = R_T + 4 * st_idx;
      asm volatile("" : "+v"(to));
      const float* st = reinterpret_cast<const float*>(lds + to);
      u0 = st[0]; u1 = st[128]; u2 = st[256];
    };
    // (the table pointers are opaque at every use: visible, the compiler reads the 33 embedding rows and a layer's bias rows
    // ONCE per tile / layer and keeps them in registers from the first use to the last -- and spills them)
    auto project = [&](float (&vh)[11], float (&vl)[11], const float t0, const float t1, const float t2) __attribute__((always_inline)) {
      uint32_t smo = R_SMALL + 132 * h;                     // rows of this half's directions (jd = min(11 h + dd, 20)), LDS bytes
      asm volatile("" : "+v"(smo));
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) {
        const float* sr = reinterpret_cast<const float*>(lds + smo + 4 * (dd < 10 ? 3 * dd : (h ? 27 : 30)));
        const float r0 = sr[0], r1 = sr[1], r2 = sr[2];
        const float p = fmaf(t2, r2, fmaf(t1, r1, t0 * r0));     // :48
        const float a0 = p * OBJ_PI_F;                                                                        // :52
        const float v = a0 * OBJ_INV2PI_HI_;
        vh[dd] = v;
        vl[dd] = fmaf(a0, OBJ_INV2PI_LO_, fmaf(a0, OBJ_INV2PI_HI_, -v));
      }
    };
    {   // x1 pieces of this wave's samples: slot u = 4 dd + f -> LDS (every wave's F1 / F3 reads them) and workspace
      float vh[11], vl[11];
      project(vh, vl, t0, t1, t2);
      if (h == 0) { float* st = reinterpret_cast<float*>(lds + R_T); st[st_idx] = t0; st[128 + st_idx] = t1; st[256 + st_idx] = t2; }
      GV* xp = (GV*)(ws_obj + a.wl.off_x1 + ((sg_own * KS_X1) << 10) + lane_off);
#pragma unroll
      for (int t = 0; t < KS_X1; ++t) {
        V xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = 8 * t + j, dd = u >> 2, f = u & 3;
          float v = 0.0f;
          if (dd < 10) v = rev_sin(vh[dd], vl[dd], (float)(1 << f));
          else if (dd == 10) {
            const float sv = rev_sin(vh[10], vl[10], (float)(1 << f));
            const float tv = f == 0 ? t0 : (f == 1 ? t1 : (f == 2 ? t2 : 0.0f));
            v = h == 0 ? sv : tv;
          }
          xf[j] = Op<OT>::cvt(v);
        }
        *reinterpret_cast<V*>(xb + ((w * KS_X1 + t) << 10) + lane_off) = xf;
        xp[t * 64] = xf;
      }
    }
    RT(17);
    RSYNC();

    // ------------------------------------------------------------------ the layer machinery
    f32x16 acc[2][2][2];                 // [phase = pair of column groups][column group of the pair][row block]
    uint32_t msk[5][4];                  // ReLU bits of this wave's rows: [layer][column group], block b in bits 8 b .. (pair form)
    uint32_t ewd2[2][8], ebits2[2] = {0, 0};
    // piece j (block j >> 3, value pair j & 7) of the epilogue of column group cgp = 2 par + ci, whose sums sit in acc[par][ci]
    auto epi_piece = [&](const bool fwd, const int par, const int ci, const int lay, const int tensor, const uint32_t hout_l, const int j)
        __attribute__((always_inline)) {
      const int b = j >> 3, i = j & 7, cgp = 2 * par + ci;
      uint32_t (&ewd)[8] = ewd2[ci];
      uint32_t& ebits = ebits2[ci];
      if (i == 0) ebits = 0;
      const uint32_t c = pk_cvt<OT>(acc[par][ci][b][2 * i], acc[par][ci][b][2 * i + 1]);
      if (fwd) {
        ewd[i] = pk_relu(c);
        ebits = ((c >> (15 - i)) & (0x00010001u << i)) | ebits;      // SIGN bits: bit i value 2 i, bit 16 + i value 2 i + 1
      } else {
        const uint32_t mb = msk[lay][cgp] >> (8 * b);
        ewd[i] = pk_mask(c, (mb >> i) & 0x00010001u);
      }
      if (i == 7) {
        const uint4 u0 = make_uint4(ewd[0], ewd[1], ewd[2], ewd[3]), u1 = make_uint4(ewd[4], ewd[5], ewd[6], ewd[7]);
        const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
        const int ks0 = 2 * (2 * w + b);
        char* dl = lds + hout_l + ((cgp * KS_H + ks0) << 10);
        *reinterpret_cast<V*>(dl) = f0;
        *reinterpret_cast<V*>(dl + PIECE) = f1;
        GV* dst = act_base(tensor, cgp, ks0);
        __builtin_nontemporal_store(f0, dst);
        __builtin_nontemporal_store(f1, dst + 64);
        if (fwd) {
          // (pinned here: left free, the compiler SINKS these bit operations to the mask's first use in the backward pass and
          // carries the sixteen packed pre-activations they come from instead -- ~150 spilled registers)
          uint32_t bits = ~ebits & 0x00ff00ffu;
          bits = b == 0 ? bits : (msk[lay][cgp] | (bits << 8));
          asm volatile("" : "+v"(bits));
          msk[lay][cgp] = bits;
        }
      }
    };
    // B fragment (column group cg, k-step ks) of a consumer whose first nkh k-steps come from the activation buffer hin
    // and the others from the nx pieces per column group at xin
    // (hin_l / xin_l: LDS byte offsets that already hold the lane's 16-byte offset -- and are OPAQUE per layer: as plain
    // expressions every fragment address of a buffer is one value the compiler computes once per tile and keeps, 136 + of
    // them, with the 64-bit weight addresses on top: ~500 spilled registers)
    auto bfrag = [&](const uint32_t hin_l, const uint32_t xin_l, const int nkh, const int nx, const int cg, const int ks) __attribute__((always_inline)) -> V {
      const uint32_t o = ks < nkh ? hin_l + ((cg * KS_H + ks) << 10) : xin_l + ((cg * nx + (ks - nkh)) << 10);
      return *reinterpret_cast<const V*>(lds + o);
    };
    auto opaque_lds = [&](const char* pbase) __attribute__((always_inline)) -> uint32_t {
      uint32_t o = (uint32_t)(pbase - lds) + lane_off;
      asm volatile("" : "+v"(o));
      return o;
    };
    [[maybe_unused]] auto opaque_img = [&](const char* pimg) __attribute__((always_inline)) -> const char* {
      asm volatile("" : "+s"(pimg));
      return pimg;
    };
    // a hidden layer (consumer c, successor n): four phases; defer: the last phase's epilogue is left to the slot pass
    // that follows (rslot's `pend`)
    auto rlayer = [&](auto c_tag, auto n_tag, auto fwd_tag, const int lay, const int tensor, const char* hin, const char* xin,
                      const int nkh, const int nx, char* hout, const char* img_next, auto defer_tag) __attribute__((always_inline)) {
      constexpr int c = decltype(c_tag)::value;
      constexpr int NK = rc_nk(c), NK16 = rc_nk16(c), NKX = NK - NK16;
      constexpr bool fwd = decltype(fwd_tag)::value, defer = decltype(defer_tag)::value;
      const uint32_t hin_l = opaque_lds(hin), xin_l = opaque_lds(xin), hout_l = opaque_lds(hout);
      const char* img_c = cons_base(c_tag, img_k);
      const char* img_x = cons_base(n_tag, img_next);
      if constexpr (NK > 16) {            // the x-slot k-steps' fragments: needed at the END of phase 0's k loop
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int x = 0; x < NKX; ++x) AX[b * NKX + x] = *(const GV*)(frag_base(c_tag, b, 16 + x, img_c) + lane_off);
      }
      // Two phases, each a PAIR of column groups: 2 row blocks x 2 column groups = four independent accumulator chains
      // (an MFMA that depends on the previous one starts ~110 cycles after it, an independent one after ~34), every chain
      // runs the whole contraction -- no second chain per block to add in the epilogue (2 of its 12 instructions per value
      // pair, + 2 accumulator reads).  The epilogue of pair 0 (32 pieces) rides behind pair 1's k-steps, two pieces each.
      V bq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < 2 * NK) {
          bq[0][i] = bfrag(hin_l, xin_l, nkh, nx, 2 * (i / NK), i % NK);
          bq[1][i] = bfrag(hin_l, xin_l, nkh, nx, 2 * (i / NK) + 1, i % NK);
        }
#pragma unroll
      for (int pair = 0; pair < 2; ++pair) {
        const int par = pair;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (fwd) {
              uint32_t bpo = R_BIAS + 4 * (lay * 256 + (2 * w + b) * 32 + h * 16);
              asm volatile("" : "+v"(bpo));
              const float* bp = reinterpret_cast<const float*>(lds + bpo);
#pragma unroll
              for (int n4 = 0; n4 < 4; ++n4) {
                const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bp + 4 * n4);
                acc[par][ci][b][4 * n4] = b4[0]; acc[par][ci][b][4 * n4 + 1] = b4[1]; acc[par][ci][b][4 * n4 + 2] = b4[2]; acc[par][ci][b][4 * n4 + 3] = b4[3];
              }
            }
          }
        if (pair == 1) {                  // registers this layer never used
#pragma unroll
          for (int r = 0; r < R_NA; ++r)
            if ((r & 15) >= NK16) loadR(n_tag, r, img_x);
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const int L = pair * NK + ks;
          const V bf0 = bq[0][L & 3], bf1 = bq[1][L & 3];
          if (L + 4 < 2 * NK) {
            bq[0][L & 3] = bfrag(hin_l, xin_l, nkh, nx, 2 * ((L + 4) / NK), (L + 4) % NK);
            bq[1][L & 3] = bfrag(hin_l, xin_l, nkh, nx, 2 * ((L + 4) / NK) + 1, (L + 4) % NK);
          }
          const V a0 = ks < 16 ? A[ks] : AX[ks - 16], a1 = ks < 16 ? A[16 + ks] : AX[NKX + ks - 16];
          const bool z = !fwd && ks == 0;          // (a backward chain starts from the literal 0: a zero VECTOR gets hoisted and spilled)
          acc[par][0][0] = Op<OT>::mfma(a0, bf0, z ? zero16() : acc[par][0][0]);
          acc[par][0][1] = Op<OT>::mfma(a1, bf0, z ? zero16() : acc[par][0][1]);
          acc[par][1][0] = Op<OT>::mfma(a0, bf1, z ? zero16() : acc[par][1][0]);
          acc[par][1][1] = Op<OT>::mfma(a1, bf1, z ? zero16() : acc[par][1][1]);
          if (pair > 0 && ks < 16) { epi_piece(fwd, 0, 0, lay, tensor, hout_l, ks); epi_piece(fwd, 0, 1, lay, tensor, hout_l, ks); }
          if (pair == 1 && ks < 16) { loadR(n_tag, ks, img_x); loadR(n_tag, 16 + ks, img_x); }
          R256_FENCE();
        }
        if (pair > 0) {
#pragma unroll
          for (int j = NK; j < 16; ++j) { epi_piece(fwd, 0, 0, lay, tensor, hout_l, j); epi_piece(fwd, 0, 1, lay, tensor, hout_l, j); }
        }
        RTP(c == C_B5H ? 28 + pair : 22 + pair);
      }
      if (!defer) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { epi_piece(fwd, 1, 0, lay, tensor, hout_l, j); epi_piece(fwd, 1, 1, lay, tensor, hout_l, j); }
        RTP(26);
      }
    };
    // a slot pass (consumer c: nb row blocks, this wave's own column group): xacc[b] = sum over the k-steps; pend: the
    // deferred last epilogue of the layer before it (fwd, lay, tensor, hout of THAT layer) rides along
    auto rslot = [&](auto c_tag, auto n_tag, const char* hin, const char* xin, const int nkh, const int nx, auto&& sink,
                     const char* img_next, auto pend_tag, auto pfwd_tag, const int play, const int ptensor, char* phout)
        // (img_next: the image the NEXT consumer's fragments come from -- the next tile's object after the last pass)
        __attribute__((always_inline)) {
      constexpr int c = decltype(c_tag)::value;
      constexpr int NK = rc_nk(c), NB = rc_nb(c);
      constexpr bool pend = decltype(pend_tag)::value, pfwd = decltype(pfwd_tag)::value;
      const uint32_t hin_l = opaque_lds(hin), xin_l = opaque_lds(xin), phout_l = opaque_lds(phout);
      const char* img_c = cons_base(c_tag, img_k);
      const char* img_x = cons_base(n_tag, img_next);
      if constexpr (NB == 1) {
#pragma unroll
        for (int r = NK; r < R_NA; ++r) loadR(n_tag, r, img_x);
      }
      // Blocks 0 and 1 run TOGETHER (two chains each = four independent chains: a chain of two leaves the matrix core idle
      // for most of its ~110-cycle dependent-issue latency), a third block afterwards on four chains (k-step mod 4), its
      // fragments having replaced block 0's as those fell free; a single block (alpha, colour) runs on four chains too.
      constexpr int NB2 = NB >= 2 ? 2 : 1;                   // blocks of the first round
      constexpr int NCH = NB == 1 ? 4 : 2;                   // chains per block in it
      {
        f32x16 ch[NB2][NCH];
        V bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[i] = bfrag(hin_l, xin_l, nkh, nx, w, i);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const V bf = bq[ks & 3];
          if (ks + 4 < NK) bq[ks & 3] = bfrag(hin_l, xin_l, nkh, nx, w, ks + 4);
#pragma unroll
          for (int b = 0; b < NB2; ++b) {
            const int r = NB == 1 ? ks : b * 16 + ks;
            ch[b][ks % NCH] = Op<OT>::mfma(A[r], bf, ks < NCH ? zero16() : ch[b][ks % NCH]);
          }
          if (pend && ks < 16) { epi_piece(pfwd, 1, 0, play, ptensor, phout_l, ks); epi_piece(pfwd, 1, 1, play, ptensor, phout_l, ks); }
#pragma unroll
          for (int b = 0; b < NB2; ++b) {
            const int r = NB == 1 ? ks : b * 16 + ks;
            if (b + 2 < NB) frag_load(A[r], frag_base(c_tag, b + 2, ks, img_c));
            else loadR(n_tag, r, img_x);
          }
          R256_FENCE();
        }
        RTP(27);
#pragma unroll
        for (int b = 0; b < NB2; ++b) {
          if constexpr (NCH == 4) sink(b, (ch[b][0] + ch[b][1]) + (ch[b][2] + ch[b][3]));
          else sink(b, ch[b][0] + ch[b][1]);
        }
      }
      if constexpr (NB == 3) {
        f32x16 ch[4];
        V bq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[i] = bfrag(hin_l, xin_l, nkh, nx, w, i);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const V bf = bq[ks & 3];
          if (ks + 4 < NK) bq[ks & 3] = bfrag(hin_l, xin_l, nkh, nx, w, ks + 4);
          ch[ks & 3] = Op<OT>::mfma(A[ks], bf, ks < 4 ? zero16() : ch[ks & 3]);
          loadR(n_tag, ks, img_x);
          R256_FENCE();
        }
        RTP(27);
        sink(2, (ch[0] + ch[1]) + (ch[2] + ch[3]));
      }
    };

    // ------------------------------------------------------------------ forward
    rlayer(std::integral_constant<int, C_F1>{}, std::integral_constant<int, C_F2>{}, std::true_type{}, 0, 0, hb0, xb, 0, KS_X1, hb0, img_k, std::false_type{});                               // h1 -> buffer 0
    if (si == 0) { float* rp = s_ray + 5 * q; rp[0] = rg0; rp[1] = rg1; rp[2] = rg2; rp[3] = rg3; rp[4] = __int_as_float(rlab); }     // (landed under F1)
    if (live) load_point((int)next_obj, next_obj != k ? 0 : tile + 1, npx, npy, npz, nzv);               // the next tile's sample
    RT(2);
    RSYNC();
    rlayer(std::integral_constant<int, C_F2>{}, std::integral_constant<int, C_F3>{}, std::true_type{}, 1, 1, hb0, xb, KS_H, 1, hb1, img_k, std::false_type{});                                // h2 -> 1
    asm volatile("" : "+v"(npx), "+v"(npy), "+v"(npz), "+v"(nzv));      // (computed HERE, not at the next tile's top: 4 live values instead of 9)
    RT(3);
    RSYNC();
    rlayer(std::integral_constant<int, C_F3>{}, std::integral_constant<int, C_F4>{}, std::true_type{}, 2, 2, hb1, xb, KS_H, KS_X1, hb0, img_k, std::false_type{});                            // h3 = f([h2 | x1]) -> 0
    RT(4);
    RSYNC();
    {   // x1 is dead: its bytes take x2 (octaves 4, 5: slot u = 2 dd + (f - 4)), the strips and later the head gradients
      float vh[11], vl[11];
      float u0, u1, u2;
      load_t(u0, u1, u2);
      project(vh, vl, u0, u1, u2);
      GV* xp = (GV*)(ws_obj + a.wl.off_x2 + ((sg_own * KS_X2) << 10) + lane_off);
#pragma unroll
      for (int t = 0; t < KS_X2; ++t) {
        V xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = 8 * t + j, dd = u >> 1, f = 4 + (u & 1);
          float v = 0.0f;
          if (dd < 10) v = rev_sin(vh[dd], vl[dd], (float)(1 << f));
          else if (dd == 10) v = h == 0 ? rev_sin(vh[10], vl[10], (float)(1 << f)) : 0.0f;
          xf[j] = Op<OT>::cvt(v);
        }
        *reinterpret_cast<V*>(xb + ((w * KS_X2 + t) << 10) + lane_off) = xf;
        xp[t * 64] = xf;
      }
      s_z[st_idx] = z_own;
    }
    rlayer(std::integral_constant<int, C_F4>{}, std::integral_constant<int, C_F5>{}, std::true_type{}, 3, 3, hb0, xb, KS_H, 1, hb1, img_k, std::false_type{});                                // h4 -> 1
    RT(5);
    RSYNC();
    rlayer(std::integral_constant<int, C_F5>{}, std::integral_constant<int, C_AL>{}, std::true_type{}, 4, 4, hb1, xb, KS_H, KS_X2, hb0, img_k, std::true_type{});                             // hc = f([h4 | x2]) -> 0
    {   // F5 block 8 on this wave's own samples: row 0 = w_alpha . h4 (raw density, model.py:81)
      rslot(std::integral_constant<int, C_AL>{}, std::integral_constant<int, C_F6>{}, hb1, xb, KS_H, KS_X2,
            [&](const int, const f32x16& r) __attribute__((always_inline)) { if (h == 0) s_raw[st_idx] = r[0] + ba; },
            img_k, std::true_type{}, std::true_type{}, 4, 4, hb0);
    RT(6);
    }
    RSYNC();
    {   // F6: colour head on hc (model.py:95)
      rslot(std::integral_constant<int, C_F6>{}, std::integral_constant<int, C_B6>{}, hb0, xb, KS_H, 1,
            [&](const int, const f32x16& r) __attribute__((always_inline)) {
              if (h == 0) { s_col[st_idx] = r[0] + boc0; s_col[TSAMP + st_idx] = r[1] + boc1; s_col[2 * TSAMP + st_idx] = r[2] + boc2; }
            }, img_k, std::false_type{}, std::false_type{}, 0, 0, hb0);
    RT(8);
    }
    RSYNC();

    RT(19);
    // ------------------------------------------------------------------ compositing + losses (loss.py:27-101)
    {
      constexpr int LPR = S < 64 ? S : 64;               // lanes per ray
      constexpr int SPL = S / LPR;                       // samples per lane
      constexpr int RPP = 64 / LPR;                      // rays per wave pass
      constexpr int NPASS = (TR + RPP - 1) / RPP;
      for (int ps = w; ps < NPASS; ps += NWAVE) {
        const int ql = lane / LPR, li = lane - ql * LPR;
        const int qq = ps * RPP + ql;
        const long rayq = tile * TR + qq;
        const bool on = (qq < TR) && (rayq < a.R);
        float gtd = 0.f, gr = 0.f, gg = 0.f, gb = 0.f;
        int lab = 2;
        if (on) { const float* rp = s_ray + 5 * qq; gtd = rp[0]; gr = rp[1]; gg = rp[2]; gb = rp[3]; lab = __float_as_int(rp[4]); }
        float occ[SPL], fr[SPL], zz[SPL], c0[SPL], c1[SPL], c2[SPL], Tn[SPL], wgt[SPL];
        float lp = 1.0f;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
          const int sl = qq * S + li * SPL + e;
          float al = 0.f;
          zz[e] = 0.f; c0[e] = c1[e] = c2[e] = 0.f;
          if (on) {
            al = 10.0f * s_raw[sl];                                     // model.py:88
            c0[e] = sigmoid_acc(s_col[sl]); c1[e] = sigmoid_acc(s_col[TSAMP + sl]); c2[e] = sigmoid_acc(s_col[2 * TSAMP + sl]);
            zz[e] = s_z[sl];
          }
          occ[e] = on ? sigmoid_acc(al) : 0.0f;                         // render_rays.py:13
          fr[e] = on ? (1.0f - occ[e]) + 1e-10f : 1.0f;                 // :38
          lp *= fr[e];
        }
        float inc = lp;
#pragma unroll
        for (int d = 1; d < LPR; d <<= 1) { const float t = __shfl_up(inc, d, LPR); if (li >= d) inc *= t; }
        float ex = __shfl_up(inc, 1, LPR);
        if (li == 0) ex = 1.0f;
        float Dl = 0.f, Ol = 0.f, C0l = 0.f, C1l = 0.f, C2l = 0.f;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
          Tn[e] = ex; ex *= fr[e];
          wgt[e] = occ[e] * Tn[e];                                      // :43
          Dl += wgt[e] * zz[e]; Ol += wgt[e]; C0l += wgt[e] * c0[e]; C1l += wgt[e] * c1[e]; C2l += wgt[e] * c2[e];
        }
        const float D = seg_sum<LPR>(Dl), Oo = seg_sum<LPR>(Ol);
        const float C0 = seg_sum<LPR>(C0l), C1 = seg_sum<LPR>(C1l), C2 = seg_sum<LPR>(C2l);
        float Vl = 0.f;
#pragma unroll
        for (int e = 0; e < SPL; ++e) { const float dz = zz[e] - D; Vl += wgt[e] * (dz * dz); }
        const float Vv = seg_sum<LPR>(Vl);                               // loss.py:32-33
        const float m1 = (lab == 1) ? 1.0f : 0.0f, m2 = (lab != 2) ? 1.0f : 0.0f, tgt = (lab != 0) ? 1.0f : 0.0f;
        const float info = 1.0f / (sqrtf(Vv) + 1e-4f);                   // render_rays.py:96-100
        const float rd = D - gtd, r0 = C0 - gr, r1 = C1 - gg, r2 = C2 - gb, ro = Oo - tgt;
        auto sgn = [](float x) { return x > 0.f ? 1.0f : (x < 0.f ? -1.0f : 0.0f); };
        const float gD = m1 * sgn(rd) * info * inv1;
        const float gC0 = a.color_scaling * m1 * sgn(r0) * inv1;
        const float gC1 = a.color_scaling * m1 * sgn(r1) * inv1;
        const float gC2 = a.color_scaling * m1 * sgn(r2) * inv1;
        const float gO = a.opacity_scaling * m2 * sgn(ro) * inv2;
        if (on && li == 0) {
          l_d += m1 * fabsf(rd) * info * inv1;
          l_c += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
          l_o += m2 * fabsf(ro) * inv2;
        }
        float dw[SPL], qv[SPL], ql_sum = 0.f;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
          dw[e] = gD * zz[e] + gO + gC0 * c0[e] + gC1 * c1[e] + gC2 * c2[e];
          qv[e] = dw[e] * wgt[e];
          ql_sum += qv[e];
        }
        float sinc = ql_sum;
#pragma unroll
        for (int d = 1; d < LPR; d <<= 1) { const float t = __shfl_down(sinc, d, LPR); if (li + d < LPR) sinc += t; }
        float after = sinc - ql_sum;                                   // sum over later lanes
#pragma unroll
        for (int e = SPL - 1; e >= 0; --e) {
          const float docc = dw[e] * Tn[e] - after / fr[e];
          after += qv[e];
          if (on) {
            const int sl = qq * S + li * SPL + e;
            s_raw[sl] = 10.0f * (docc * occ[e] * (1.0f - occ[e]));     // d / d raw (model.py:88)
            s_col[sl] = gC0 * wgt[e] * c0[e] * (1.0f - c0[e]);         // d / d colour pre-activation
            s_col[TSAMP + sl] = gC1 * wgt[e] * c1[e] * (1.0f - c1[e]);
            s_col[2 * TSAMP + sl] = gC2 * wgt[e] * c2[e] * (1.0f - c2[e]);
          }
        }
      }
    }
    RSYNC();
    RT(20);
    {   // head gradient piece of this wave's samples: slot (0, 0) = d raw, (0, 1 + c) = d colour pre-activation (x scale)
      V dh;
#pragma unroll
      for (int j = 0; j < 8; ++j) dh[j] = (OT)0.0f;
      if (h == 0 && valid) {
        dh[0] = Op<OT>::cvt(s_raw[st_idx] * gs);
        dh[1] = Op<OT>::cvt(s_col[st_idx] * gs);
        dh[2] = Op<OT>::cvt(s_col[TSAMP + st_idx] * gs);
        dh[3] = Op<OT>::cvt(s_col[2 * TSAMP + st_idx] * gs);
      }
      *reinterpret_cast<V*>(dhb + (w << 10) + lane_off) = dh;
      *(GV*)(ws_obj + a.wl.off_dhead + (sg_own << 10) + lane_off) = dh;
    }
    RSYNC();

    RT(21);
    // ------------------------------------------------------------------ backward
    rlayer(std::integral_constant<int, C_B6>{}, std::integral_constant<int, C_B5H>{}, std::false_type{}, 4, 9, hb0, dhb, 0, 1, hb1, img_k, std::false_type{});                                // d_pre5 = mask(W_oc^T d colour) -> 1
    RT(9);
    RSYNC();
    rlayer(std::integral_constant<int, C_B5H>{}, std::integral_constant<int, C_B5X>{}, std::false_type{}, 3, 8, hb1, dhb, KS_H, 1, hb0, img_k, std::true_type{});                             // d_pre4 -> 0
    // the embedding's chain rule (embedding.py:49-52), applied to a 16-slot block of gradients as it completes: OCT octaves
    // per direction starting at F0 (x1: 4 from 0, x2: 2 from 4), slot u = OCT dd + (f - F0).  The chain rule is linear in the
    // slot gradients, so every contribution goes straight into d B: d B[j][c] += d proj_j * t_c, summed over the half-wave's
    // samples (DPP row sums + one row swap) and added to the wave's LDS sums by lane 0 -- nothing is carried from pass to
    // pass (eleven registers through three layers otherwise, reloaded one by one from scratch at the tile's end).
    auto pe_bwd_block = [&](auto oct_tag, auto f0_tag, const int b, const f32x16& r) __attribute__((always_inline)) {
      constexpr int OCT = decltype(oct_tag)::value, F0 = decltype(f0_tag)::value;
      constexpr int DPB = 16 / OCT;                          // directions per block
      float u0, u1, u2;
      load_t(u0, u1, u2);
      // three batches over the block's directions -- projections (their table rows in flight together), angle ladders and
      // products, half-wave sums (independent DPP chains interleave) -- with a fence between the batches only: fenced per
      // direction every LDS round trip and every dependent DPP chain was exposed (~330 cycles x 33 directions per tile)
      constexpr int ND = (DPB == 8) ? 8 : 4;
      float vh[ND], vl[ND];
#pragma unroll
      for (int d2 = 0; d2 < ND; ++d2) {
        const int dd = b * DPB + d2;
        vh[d2] = vl[d2] = 0.0f;
        if (dd < 11) {
          uint32_t smo = R_SMALL + 132 * h + 4 * (dd < 10 ? 3 * dd : (h ? 27 : 30));
          asm volatile("" : "+v"(smo));
          const float* sr = reinterpret_cast<const float*>(lds + smo);
          const float p = fmaf(u2, sr[2], fmaf(u1, sr[1], u0 * sr[0]));
          const float a0 = p * OBJ_PI_F;
          vh[d2] = a0 * OBJ_INV2PI_HI_;
          vl[d2] = fmaf(a0, OBJ_INV2PI_LO_, fmaf(a0, OBJ_INV2PI_HI_, -vh[d2]));
        }
      }
      R256_FENCE();
      float g[ND][3];
#pragma unroll
      for (int d2 = 0; d2 < ND; ++d2) {
        const int dd = b * DPB + d2;
        g[d2][0] = g[d2][1] = g[d2][2] = 0.0f;
        if (dd < 11) {
          float sv[OCT], cv[OCT];
          rev_ladder<F0, OCT>(vh[d2], vl[d2], sv, cv);
          float dpf = 0.0f;
#pragma unroll
          for (int f = 0; f < OCT; ++f) dpf = fmaf(r[OCT * d2 + f], cv[f] * (OBJ_PI_F * (float)(1 << (F0 + f))), dpf);
          const float dp = (valid && (dd < 10 || h == 0)) ? dpf * inv_gs : 0.0f;
          g[d2][0] = dp * u0; g[d2][1] = dp * u1; g[d2][2] = dp * u2;
        }
      }
      R256_FENCE();
#pragma unroll
      for (int d2 = 0; d2 < ND; ++d2) {
        const int dd = b * DPB + d2;
        if (dd < 11) {
          const float g0 = wave_sum32(g[d2][0]), g1 = wave_sum32(g[d2][1]), g2 = wave_sum32(g[d2][2]);
          dbl += s == 3 * dd ? g0 : (s == 3 * dd + 1 ? g1 : (s == 3 * dd + 2 && dd < 10 ? g2 : 0.0f));
          if (dd == 10) dbl32 += g2;         // (half 1 has no eleventh direction: its dp is 0 there)
        }
      }
      R256_FENCE();
    };
    // B5X: d x2 of this wave's samples
    rslot(std::integral_constant<int, C_B5X>{}, std::integral_constant<int, C_B4>{}, hb1, xb, KS_H, 1,
          [&](const int b, const f32x16& r) __attribute__((always_inline)) { pe_bwd_block(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, b, r); },
          img_k, std::true_type{}, std::false_type{}, 3, 8, hb0);
    RT(11);
    RSYNC();
    rlayer(std::integral_constant<int, C_B4>{}, std::integral_constant<int, C_B3H>{}, std::false_type{}, 2, 7, hb0, xb, KS_H, 1, hb1, img_k, std::false_type{});                              // d_pre3 -> 1
    RT(12);
    RSYNC();
    rlayer(std::integral_constant<int, C_B3H>{}, std::integral_constant<int, C_B3X>{}, std::false_type{}, 1, 6, hb1, xb, KS_H, 1, hb0, img_k, std::true_type{});                              // d_pre2 -> 0
    // B3X: d x1 (cat layer's share) from d_pre3
    rslot(std::integral_constant<int, C_B3X>{}, std::integral_constant<int, C_B2>{}, hb1, xb, KS_H, 1,
          [&](const int b, const f32x16& r) __attribute__((always_inline)) { pe_bwd_block(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}, b, r); },
          img_k, std::true_type{}, std::false_type{}, 1, 6, hb0);
    RT(13);
    RSYNC();
    rlayer(std::integral_constant<int, C_B2>{}, std::integral_constant<int, C_B1>{}, std::false_type{}, 0, 5, hb0, xb, KS_H, 1, hb1, img_k, std::false_type{});                               // d_pre1 -> 1
    RT(15);
    RSYNC();
    // B1: d x1 += W_in^T d_pre1; its registers take the next tile's first layer as they fall free
    rslot(std::integral_constant<int, C_B1>{}, std::integral_constant<int, C_END>{}, hb1, xb, KS_H, 1,
          [&](const int b, const f32x16& r) __attribute__((always_inline)) { pe_bwd_block(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}, b, r); },
          img_n, std::false_type{}, std::false_type{}, 0, 0, hb0);
    RT(16);
    RT(18);
    {   // the next tile's first layer (requested HERE, behind the reduction: whatever the compiler reloads from scratch at the
        // tile boundary would otherwise queue behind these loads -- vmcnt completes in order)
      const char* cbn = cons_base(std::integral_constant<int, C_F1>{}, img_n);
#pragma unroll
      for (int r = 0; r < R_NA; ++r) loadR(std::integral_constant<int, C_F1>{}, r, cbn);
    }
    if (++tile_i == a.ntile) { tile_i = 0; ++k_i; }
  }
  flush_object();
#ifdef OBJ256_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("r256 (cycles of workgroup 0, wave 0): before barriers %llu in barriers %llu | F1 %llu F2 %llu F3 %llu F4 %llu F5 %llu alpha %llu colour %llu | B6 %llu B5H %llu "
           "B5X %llu B4 %llu B3H %llu B3X %llu B2 %llu B1 %llu | tile tail + head %llu\n", tm_[0], tm_[1], tm_[2], tm_[3], tm_[4], tm_[5], tm_[6], tm_[7], tm_[8],
           tm_[9], tm_[10], tm_[11], tm_[12], tm_[13], tm_[14], tm_[15], tm_[16], tm_[17]);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("r256 more: d B reduce %llu | layer phases 0..3 %llu %llu %llu %llu, exposed epilogue %llu | slot blocks %llu | B5H phases %llu %llu %llu %llu\n", tm_[18],
           tm_[22], tm_[23], tm_[24], tm_[25], tm_[26], tm_[27], tm_[28], tm_[29], tm_[30], tm_[31]);
#endif
}
