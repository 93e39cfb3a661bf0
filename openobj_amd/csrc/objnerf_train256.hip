// One training iteration of K stacked hidden-256 object networks in the 16-bit operand modes (BASELINE configs[4]:
// 512-object stress, 8192 rays x 128 samples, hidden 256, fp16) -- train.py:424-472 on model.py:61-103 /
// embedding.py:46-55 / loss.py:5-103, gfx950 only.
//
// Why this path has the shape it has.  The layer-wise chain it replaces moved 25 KB of HBM per sample (every layer's
// activations written, re-read by the next layer, by the backward GEMM and by the weight gradient).  What bounds a
// fused design at this width is ON-CHIP CAPACITY, not arithmetic: one object's weight gradient is 1.24 MB of fp32
// accumulators, a CU holds 512 KB of registers + 160 KB of LDS, so no workgroup can keep dW resident while it sweeps
// samples, and flushing a partial dW per tile costs more HBM traffic (2.5 MB per tile) than storing the operands.
// The minimum-traffic split is therefore two kernels:
//
//   kernel A  (fwd256_kernel)   forward + compositing + losses + the whole input-gradient chain of a 256-sample tile
//                               with every activation in REGISTERS: a wave owns 32 samples, a layer's 32 x 32
//                               accumulator block is -- converted to the operand type in place -- the next layer's MFMA
//                               B operand (v_mfma_f32_32x32x16: rows = features on the registers, column = sample on
//                               the lane), the weights stream L2 -> LDS as pre-packed A operands (pack256_kernel: one
//                               ds_read_b128 per MFMA, no conversion, a three-slot ring two stages ahead of the
//                               MFMAs).  It leaves exactly what the weight gradient needs: h1..h4, hc and the five
//                               pre-activation gradients as 16-bit MFMA fragments, x1 / x2 and the head gradients
//                               (5.4 KB per sample, each 1-KB fragment one fully coalesced wave store), and d B.
//   kernel B  (wgrad256_kernel) dW = d_pre^T . input over the samples: fragments back through LDS (8-byte stores into a
//                               72-byte-pitch [sample][feature] image, ds_read_b64_tr_b16 turns them into operands with
//                               the SAMPLE as contraction index), split over sample ranges, partial tiles into slabs,
//                               summed in slab order by finalize256_kernel (no atomics: bit-reproducible).
//
// Algorithmic traffic is 16 B per sample; this path moves 5.4 KB written + 6.4 KB read per sample.
// DESIGN.md section 4.9 has the capacity argument and the measured numbers.
//
// Arithmetic = the operand-rounded specification of the 16-bit modes (oracle.mlp_forward_stacked_16 with act16,
// round_head_weights, round_head_grads): fp32 accumulation, biases, compositing, losses and master weights; operands
// (weights, embedding, stored activations, stored pre-activation gradients -- fp16 pre-scaled by 2^(floor(log2 R)+3))
// rounded to the operand type.  Not the reference's fp32 arithmetic: opt-in, PSNR-gated.
#include <type_traits>
#include "objnerf_device.h"
#include "objnerf_generic.h"

namespace obj256 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int HID = 256, NWAVE = 8, NTHR = 512;     // (kernel B and the widest kernel A; kernel A: NW waves x 32 samples)
constexpr int KS_H = 16, KS_X1 = 6, KS_X2 = 3;       // k-steps (16 features each) of a hidden vector, of x1, of x2
constexpr int PIECE = 1024;                          // one MFMA operand of a whole wave: 64 lanes x 8 x 16 bit

template <typename OT> struct Op;
template <> struct Op<__bf16> {
  typedef bf16x8 V;
  static __device__ __forceinline__ __bf16 cvt(float x) { return (__bf16)x; }
  static __device__ __forceinline__ f32x16 mfma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Op<_Float16> {
  typedef f16x8 V;
  static __device__ __forceinline__ _Float16 cvt(float x) { return (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f); }
  static __device__ __forceinline__ f32x16 mfma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// ---------------------------------------------------------------------------------------------------------------
// Index maps.  MFMA 32x32x16: A lane (r = l & 31, h = l >> 5) element j = A[row r][k = 8 h + j]; B lane (s, h) element
// j = B[k = 8 h + j][col s]; D lane (s, h) register n = D[row (n & 3) + 8 (n >> 2) + 4 h][col s].
// A hidden vector as B operand: k-step ks, half h, element j  <->  feature hid_feat(ks, h, j) -- exactly where the
// accumulator block ks >> 1 of the producing layer holds it (register 8 (ks & 1) + j), so no lane ever moves.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int hid_feat(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
__host__ __device__ constexpr int acc_row(int n, int h) { return (n & 3) + 8 * (n >> 2) + 4 * h; }
constexpr int COL_ZERO = -1;
// Embedding slots.  Lane half h owns directions 11 h .. 11 h + 10 (h = 1: ten of them); x1 slot u = 8 t + j of half h:
// u = 4 dd + f (direction 11 h + dd, octave f < 4); half 1 carries x / scale in u = 40..42.  x2 slot u = 2 dd + (f - 4).
// -> column of the reference's embedding row (embedding.py:46-55: [x / scale | sin(2^f pi B x), band-major]) or COL_ZERO
__host__ __device__ constexpr int x1_slot_col(int h, int u) {
  const int dd = u >> 2, f = u & 3;
  if (dd < 10 || (dd == 10 && h == 0)) return 3 + OBJ_NDIR * f + 11 * h + dd;
  if (h == 1 && u >= 40 && u < 43) return u - 40;
  return COL_ZERO;
}
__host__ __device__ constexpr int x2_slot_col(int h, int u) {      // column inside emb[87:]
  const int dd = u >> 1, f = u & 1;
  if (dd < 10 || (dd == 10 && h == 0)) return OBJ_NDIR * f + 11 * h + dd;
  return COL_ZERO;
}

// ---------------------------------------------------------------------------------------------------------------
// The packed weight image of one object: the A operands of every MFMA of kernel A in consumption order.
// sequence      A rows (32 per block)                      contraction (k-steps)                blocks  k-steps
//  F1  h1       W_in                                       x1 slots                               8       6
//  F2  h2       W_m1                                       h1                                     8      16
//  F3  h3       W_cat                                      h2 | x1 slots                          8      22
//  F4  h4       W_m2                                       h3                                     8      16
//  F5  hc,raw   W_cl ; block 8 row 0 = w_alpha             h4 | x2 slots                          9      19
//  F6  colour   block row c < 3 = W_oc[c]                  hc                                     1      16
//  B6  d hc     rows = hc features: W_oc^T                 head slots (h = 0, j = 1 + c)          8       1
//  B5H d h4     W_cl[:, :H]^T ; k-step 16 slot (0,0) = w_alpha   d hc | head slot (0, 0)          8      17
//  B5X d x2     rows = x2 slots (2 blocks): W_cl[:, H:]^T  d hc                                   2      16
//  B4  d h3     W_m2^T                                     d h4                                   8      16
//  B3H d h2     W_cat[:, :H]^T                             d h3                                   8      16
//  B3X d x1     rows = x1 slots (3 blocks): W_cat[:, H:]^T d h3                                   3      16
//  B2  d h1     W_m1^T                                     d h2                                   8      16
//  B1  d x1 +=  rows = x1 slots: W_in^T                    d h1                                   3      16
// A "stage" of the LDS ring = one block of a sequence (its k-steps' pieces are contiguous).  Slot rows of the B?X
// sequences: row rho of block b belongs to lane half h = (rho >> 2) & 1 and is its register n = 4 (rho >> 3) + (rho & 3),
// slot u = 16 b + n -- every lane receives the gradients of ITS OWN slots.
// ---------------------------------------------------------------------------------------------------------------
enum Seq { F1, F2, F3, F4, F5, F6, B6, B5H, B5X, B4, B3H, B3X, B2, B1, NSEQ };
__host__ __device__ constexpr int seq_nb(int q) {
  return q == F5 ? 9 : q == F6 ? 1 : q == B5X ? 2 : (q == B3X || q == B1) ? 3 : 8;
}
__host__ __device__ constexpr int seq_nk(int q) {
  return q == F1 ? 6 : q == F3 ? 22 : q == F5 ? 19 : q == B6 ? 1 : q == B5H ? 17 : 16;
}
__host__ __device__ constexpr int seq_off(int q) {       // first piece of the sequence
  int o = 0;
  for (int i = 0; i < q; ++i) o += seq_nb(i) * seq_nk(i);
  return o;
}
constexpr int N_PIECES = seq_off(NSEQ);
constexpr int N_STAGES = 8 + 8 + 8 + 8 + 9 + 1 + 8 + 8 + 2 + 8 + 8 + 3 + 8 + 3;
static_assert(N_PIECES == 1323 && N_STAGES == 90, "image size");
constexpr long IMG_BYTES = (long)N_PIECES * PIECE;
constexpr int MAX_NK = 22;
constexpr int RING_SLOT = MAX_NK * PIECE;            // 22 KB
// stage -> (first piece, k-steps); one period, read with a wave-uniform index
struct StageTab { short off[N_STAGES + 2]; signed char nk[N_STAGES + 2]; };
__host__ __device__ constexpr StageTab make_stage_tab() {
  StageTab t{};
  int g = 0;
  for (int q = 0; q < NSEQ; ++q)
    for (int b = 0; b < seq_nb(q); ++b) { t.off[g] = (short)(seq_off(q) + b * seq_nk(q)); t.nk[g] = (signed char)seq_nk(q); ++g; }
  t.off[g] = t.off[0]; t.nk[g] = t.nk[0];
  t.off[g + 1] = t.off[1]; t.nk[g + 1] = t.nk[1];
  return t;
}
__device__ __constant__ StageTab c_stage = make_stage_tab();

struct Lay256 { int in_w, in_b, m1_w, m1_b, cat_w, cat_b, m2_w, m2_b, a_w, a_b, cl_w, cl_b, oc_w, oc_b, pe_b; };

template <typename OT>
__global__ __launch_bounds__(256) void pack256_kernel(int K, const float* __restrict__ params, long p_stride, Lay256 L,
                                                      OT* __restrict__ img) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // (object, piece, lane)
  if (idx >= (long)K * N_PIECES * 64) return;
  const int lane = (int)(idx & 63);
  const int piece = (int)((idx >> 6) % N_PIECES);
  const int k = (int)((idx >> 6) / N_PIECES);
  const float* P = params + (long)k * p_stride;
  int q = 0;
  while (q + 1 < NSEQ && piece >= seq_off(q + 1)) ++q;
  const int rel = piece - seq_off(q), nk = seq_nk(q);
  const int blk = rel / nk, ks = rel - blk * nk;
  const int r = lane & 31, h = lane >> 5;
  const int E1 = OBJ_E1, E2 = OBJ_E2;
  typename Op<OT>::V out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.0f;
    const int o = 32 * blk + r;                    // forward: output feature; backward: row index
    switch (q) {
      case F1: { const int c = x1_slot_col(h, 8 * ks + j); if (c >= 0) v = P[L.in_w + o * E1 + c]; break; }
      case F2: v = P[L.m1_w + o * HID + hid_feat(ks, h, j)]; break;
      case F3:
        if (ks < KS_H) v = P[L.cat_w + o * (HID + E1) + hid_feat(ks, h, j)];
        else { const int c = x1_slot_col(h, 8 * (ks - KS_H) + j); if (c >= 0) v = P[L.cat_w + o * (HID + E1) + HID + c]; }
        break;
      case F4: v = P[L.m2_w + o * HID + hid_feat(ks, h, j)]; break;
      case F5:
        if (blk < 8) {
          if (ks < KS_H) v = P[L.cl_w + o * (HID + E2) + hid_feat(ks, h, j)];
          else { const int c = x2_slot_col(h, 8 * (ks - KS_H) + j); if (c >= 0) v = P[L.cl_w + o * (HID + E2) + HID + c]; }
        } else if (r == 0 && ks < KS_H) v = P[L.a_w + hid_feat(ks, h, j)];
        break;
      case F6: if (r < 3) v = P[L.oc_w + r * HID + hid_feat(ks, h, j)]; break;
      case B6: if (h == 0 && j >= 1 && j < 4) v = P[L.oc_w + (j - 1) * HID + o]; break;
      case B5H:
        if (ks < KS_H) v = P[L.cl_w + hid_feat(ks, h, j) * (HID + E2) + o];
        else if (h == 0 && j == 0) v = P[L.a_w + o];
        break;
      case B5X: case B3X: case B1: {
        const int hr = (r >> 2) & 1, n = 4 * (r >> 3) + (r & 3), u = 16 * blk + n;
        const int kf = hid_feat(ks, h, j);
        if (q == B5X) { const int c = u < 24 ? x2_slot_col(hr, u) : COL_ZERO; if (c >= 0) v = P[L.cl_w + kf * (HID + E2) + HID + c]; }
        else { const int c = x1_slot_col(hr, u);
               if (c >= 0) v = (q == B3X) ? P[L.cat_w + kf * (HID + E1) + HID + c] : P[L.in_w + kf * E1 + c]; }
        break;
      }
      case B4: v = P[L.m2_w + hid_feat(ks, h, j) * HID + o]; break;
      case B3H: v = P[L.cat_w + hid_feat(ks, h, j) * (HID + E1) + o]; break;
      case B2: v = P[L.m1_w + hid_feat(ks, h, j) * HID + o]; break;
    }
    out[j] = Op<OT>::cvt(v);
  }
  reinterpret_cast<typename Op<OT>::V*>(img)[idx] = out;
}

// ---------------------------------------------------------------------------------------------------------------
// Workspace of one call (K objects, n = R S samples each, padded to whole tiles): per object
//   acts  [5][NSG][16][64][8] 16 bit   h1 h2 h3 h4 hc as B fragments (k-step ks of sample group sg: 1 KB)
//   dpre  [5][NSG][16][64][8]          d pre-activation of L1 L2 L3 L4 L5 (fp16: times the gradient scale)
//   x1    [NSG][6][64][8], x2 [NSG][3][64][8], dhead [NSG][64][8]  (head slot (0,0) = d raw alpha, (0,1..3) = d colour)
// ---------------------------------------------------------------------------------------------------------------
struct WsLay {
  long nsg;                 // sample groups of 32 per object
  long act_stride;          // bytes of one activation tensor of one object
  long obj_bytes;           // everything of one object
  long off_dpre, off_x1, off_x2, off_dhead;
  __host__ __device__ static WsLay make(long n, int tsamp) {
    WsLay w;
    const long ntile = (n + tsamp - 1) / tsamp;         // whole tiles of kernel A (tsamp samples each)
    w.nsg = ntile * (tsamp / 32);
    w.act_stride = w.nsg * KS_H * PIECE;
    w.off_dpre = 5 * w.act_stride;
    w.off_x1 = 10 * w.act_stride;
    w.off_x2 = w.off_x1 + w.nsg * KS_X1 * PIECE;
    w.off_dhead = w.off_x2 + w.nsg * KS_X2 * PIECE;
    w.obj_bytes = w.off_dhead + w.nsg * PIECE;
    return w;
  }
};

constexpr int PART_FLOATS = 72;         // per (object, workgroup): d B (63), loss terms (3), padding
constexpr int NWG_A = 256;

struct FwdArgs {
  int K, R, S, TR;                      // TR = rays per tile
  long ntile;                           // tiles per object
  float color_scaling, opacity_scaling, obj_center, grad_scale;
  const float* params; long p_stride; const float* scale;
  const float* pts; const float* origins; const float* dirs; const float* z;
  const float* gt_depth; const float* gt_rgb; const uint8_t* labels;
  const int* counts; const int* flags;
  const void* img;                      // packed weight images [K][IMG_BYTES]
  char* ws;                             // activation workspace
  float* part;                          // [K][NWG_A][PART_FLOATS]
  Lay256 L;
  WsLay wl;
};

// LDS of kernel A (NW waves, NW * 64 threads, tile of NW * 32 samples)
constexpr int L_RING = 0;                                   // 3 x 22 KB weight ring
constexpr int L_HBUF = L_RING + 3 * RING_SLOT;              // per wave 16 KB: the 16 operand fragments of the layer being produced
__host__ __device__ constexpr int l_mask(int NW) { return L_HBUF + NW * KS_H * PIECE; }            // [5 layers][4 block pairs][threads] ReLU bits
__host__ __device__ constexpr int l_bias(int NW) { return l_mask(NW) + 5 * 4 * NW * 64 * 4; }      // [5][8 blk][2 h][16] fp32
__host__ __device__ constexpr int l_strip(int NW) { return l_bias(NW) + 5 * 256 * 4; }             // raw alpha | colour pre [3] per sample
__host__ __device__ constexpr int l_small(int NW) { return l_strip(NW) + 4 * NW * 32 * 4; }        // B rows, ba, boc[3] (96 floats)
__host__ __device__ constexpr int l_db(int NW) { return l_small(NW) + 96 * 4; }                    // per-wave d B [64] + loss [4]
__host__ __device__ constexpr int l_total(int NW) { return (l_db(NW) + NW * 68 * 4 + 15) & ~15; }
static_assert(l_total(4) <= 163840, "LDS budget");

#define OBJ_INV2PI_HI_ 0.15915494f
#define OBJ_INV2PI_LO_ 4.4620826e-09f

// sin / cos of 2^f a (a = fp32(proj * pi), embedding.py:49-52: fl(fl(p 2^f) pi) == 2^f fl(p pi)) from the angle in
// revolutions vh + vl: the scaling by 2^f and the subtraction of the nearest integer are exact
__device__ __forceinline__ void rev_sincos(const float vh, const float vl, const float sc, float& s, float& c) {
  const float u = vh * sc;
  const float r = u - rintf(u);
  const float w = fmaf(vl, sc, r);
  s = __builtin_amdgcn_sinf(w);
  c = __builtin_amdgcn_cosf(w);
}
__device__ __forceinline__ float rev_sin(const float vh, const float vl, const float sc) {
  const float u = vh * sc;
  const float r = u - rintf(u);
  return __builtin_amdgcn_sinf(fmaf(vl, sc, r));
}

// ---- epilogues on PACKED words (two 16-bit values per register): the ReLU is a packed integer max with zero (a
// negative value has its sign bit set in either 16-bit format), the branch bit of each half a packed min with 1, and
// the backward mask a packed integer multiply by those 0 / 1 halves.  ~5 instructions per value pair instead of ~13.
__device__ __forceinline__ uint32_t pk_relu(uint32_t w) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(w));
  return r;
}
__device__ __forceinline__ uint32_t pk_nonzero(uint32_t w) {                          // 1 per non-zero half (0x00010001-style)
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(0x00010001u));
  return r;
}
__device__ __forceinline__ uint32_t pk_mask(uint32_t w, uint32_t m01) {               // halves of w times the 0 / 1 halves of m01
  uint32_t r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(m01));
  return r;
}
template <typename OT> __device__ __forceinline__ uint32_t pk_cvt(float a, float b);
template <> __device__ __forceinline__ uint32_t pk_cvt<__bf16>(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pk_cvt<_Float16>(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = {(_Float16)__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), (_Float16)__builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
  return __builtin_bit_cast(uint32_t, v);
}

#ifdef OBJ256_TIMING      // diagnostic build: cycles per part of a stage (s_memtime), printed by workgroup 0 / wave 0
#define T256_DECL unsigned long long tm_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tm_t = __builtin_amdgcn_s_memtime()
#define T256(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm_[i] += t_ - tm_t; tm_t = t_; } while (0)
#else
#define T256_DECL do {} while (0)
#define T256(i) do {} while (0)
#endif

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// segmented (width LPR) sum of the compositing pass
template <int LPR> __device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int d = LPR / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, LPR);
  return v;
}

template <typename OT>
struct KA {
  typedef typename Op<OT>::V V;
  typedef Op<OT> O;

  static __device__ __forceinline__ V zero_frag() {
    V v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (OT)0.0f;
    return v;
  }

  // -------------------------------------------------------------------------------------------------------------
  // ring: stage g of this workgroup's stream lives in slot g % 3.  During stage g the waves issue the LDS-DMA
  // (global_load_lds_dwordx4: 1 KB per wave instruction, no registers) of the pieces of stage g + 2 into the slot that
  // held stage g - 1, which every wave has left (they all passed the barrier that ended it).  A stage ends with a
  // COUNTED wait (stage_sync<C>): everything older than this stage's own C transfers has landed -- the pieces of the
  // stage computed next and the previous stage's fragment stores -- while the newest transfer stays in flight across
  // the barrier: two stages of latency are hidden.  (The count is wave-uniform and dynamic: the wait is picked by a
  // scalar switch -- s_waitcnt takes immediates only.)
  // -------------------------------------------------------------------------------------------------------------
  struct Ring {
    const char* img;          // packed images of all objects
    long tiles_per_obj;
    long tau_end;             // one past this workgroup's last tile
    long tau;                 // tile of the stage being computed
    long obj, tile;           // = tau / tiles_per_obj, tau % tiles_per_obj, kept incrementally (no division per stage)
    int st;                   // its stage index inside the tile (0 .. 89)
    int slot;                 // its ring slot
    int wave, nw;
    uint32_t lds_ring, voff;                // LDS byte address of the ring; lane * 16
    // the transfer being issued (scalars): this wave moves pieces wave, wave + nw, ... of the stage: cnt of them
    unsigned long long src; uint32_t dst; int cnt;

    // generic form (prologue): table lookups
    __device__ __forceinline__ void prepare(const int ahead) {
      int st2 = __builtin_amdgcn_readfirstlane(st) + ahead;
      setup(st2 >= N_STAGES, c_stage.off[st2 >= N_STAGES ? st2 - N_STAGES : st2], c_stage.nk[st2 >= N_STAGES ? st2 - N_STAGES : st2], ahead);
    }
    // the stage two ahead of block blk of sequence Q, from compile-time tables: no memory access, a few scalar selects
    template <int Q>
    __device__ __forceinline__ void prepare2(const int blk) {
      constexpr int NB = seq_nb(Q), NK = seq_nk(Q), OFF = seq_off(Q);
      constexpr int Q1 = (Q + 1) % NSEQ, Q2 = (Q + 2) % NSEQ, Q3 = (Q + 3) % NSEQ;
      // stage index (relative to the first block of Q) -> (sequence, block): at most three sequences ahead
      constexpr int NB1 = seq_nb(Q1), NB2 = seq_nb(Q2);
      const int e = blk + 2 - NB;                       // >= 0: past the end of Q
      int off, nk; bool wrap;
      if (e < 0) { off = OFF + (blk + 2) * NK; nk = NK; wrap = false; }
      else if (e < NB1) { off = seq_off(Q1) + e * seq_nk(Q1); nk = seq_nk(Q1); wrap = Q1 < Q; }
      else if (e - NB1 < NB2) { off = seq_off(Q2) + (e - NB1) * seq_nk(Q2); nk = seq_nk(Q2); wrap = Q2 < Q; }
      else { off = seq_off(Q3) + (e - NB1 - NB2) * seq_nk(Q3); nk = seq_nk(Q3); wrap = Q3 < Q; }
      setup(wrap, off, nk, 2);
    }
    __device__ __forceinline__ void setup(const bool wrap, const int off, const int nk, const int ahead) {
      long obj2 = obj;
      bool live = true;
      if (wrap) {
        live = tau + 1 < tau_end;
        if (live && tile + 1 == tiles_per_obj) obj2 += 1;
      }
      int s2 = __builtin_amdgcn_readfirstlane(slot) + ahead; if (s2 >= 3) s2 -= 3;
      const int npiece = live ? nk : 0;
      cnt = __builtin_amdgcn_readfirstlane(npiece > wave ? (npiece - wave + nw - 1) / nw : 0);
      const unsigned long long sa = (unsigned long long)(img + obj2 * IMG_BYTES + ((long)off + wave) * PIECE);
      // scalar registers, provably (readfirstlane returns int: through uint32_t, or the low half is sign-extended)
      src = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa >> 32)) << 32) |
            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa);
      dst = __builtin_amdgcn_readfirstlane(lds_ring + s2 * RING_SLOT + wave * PIECE);
    }
    template <int I>
    __device__ __forceinline__ void issue() const {                     // the I-th piece of this wave, if it has one
      if (I < cnt) {
        const unsigned long long sa = src + (unsigned long long)I * nw * PIECE;
        const uint32_t da = dst + I * nw * PIECE;
        // M0 (the LDS destination) belongs to the compiler: saved and restored inside the statement that uses it.
        // Operands are scalar-ALU results of values made scalar in prepare(), long before: no VALU-written SGPR reaches
        // the VMEM instruction inside its 5 wait states (hipcc pads nothing for inline asm).
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(da), "v"(voff), "s"(sa) : "memory");
      }
    }
    __device__ __forceinline__ void advance() {
      if (++slot == 3) slot = 0;
      if (++st == N_STAGES) { st = 0; ++tau; if (++tile == tiles_per_obj) { tile = 0; ++obj; } }
    }
  };
  // end of a stage in which this wave issued cnt transfers (raw s_barrier: __syncthreads() would drain the DMA)
  static __device__ __forceinline__ void stage_sync(const int cnt) {
    switch (cnt) {
      case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  }
  // DMA instructions per wave for a stage issued from sequence q: the stage two ahead lies in q, q + 1 or q + 2
  static constexpr int dma_count(int q, int nw) {
    int m = seq_nk(q);
    const int q1 = (q + 1) % NSEQ, q2 = (q + 2) % NSEQ;
    if (seq_nk(q1) > m) m = seq_nk(q1);
    if (seq_nk(q2) > m) m = seq_nk(q2);
    return (m + nw - 1) / nw;
  }

  // one block of a sequence: acc += sum over the stage's k-steps of piece(ks) x bfrag(ks).  The A operands are read
  // DEPTH k-steps ahead by hand (inline ds_read_b128 with literal offsets + counted lgkmcnt waits tied to the registers
  // they release).  side(ks) is called after the MFMA of k-step ks has been issued: work that does not depend on the
  // accumulator (the DMA of a later stage, the previous block's fragment stores) goes into the MFMA's shadow there --
  // with one wave per SIMD nothing else would overlap it.
  static constexpr int DEPTH = 6;
  template <int OFF>
  static __device__ __forceinline__ void rd(V& dst, const uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
  }
  template <int N>
  static __device__ __forceinline__ void wait_for(V& x) {       // at most N LDS reads still outstanding
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "i"(N));
  }
  template <int NK, int KS, class BF, class SD>
  static __device__ __forceinline__ void step(f32x16& acc, V (&a)[DEPTH], const uint32_t addr, const BF& bfrag, SD& side) {
    if constexpr (KS < NK) {
      constexpr int inflight = (NK - 1 - KS) < (DEPTH - 1) ? (NK - 1 - KS) : (DEPTH - 1);   // reads issued after read KS
      wait_for<inflight>(a[KS % DEPTH]);
      acc = O::mfma(a[KS % DEPTH], bfrag(KS), acc);
      if constexpr (KS + DEPTH < NK) rd<(KS + DEPTH) * PIECE>(a[KS % DEPTH], addr);
      side(std::integral_constant<int, KS>{});
      step<NK, KS + 1>(acc, a, addr, bfrag, side);
    }
  }
  template <int NK, class BF, class SD>
  static __device__ __forceinline__ void block_mma(f32x16& acc, const uint32_t addr, const BF& bfrag, SD& side) {
    V a[DEPTH];
    rd<0>(a[0], addr);
    if constexpr (NK > 1) rd<PIECE>(a[1], addr);
    if constexpr (NK > 2) rd<2 * PIECE>(a[2], addr);
    if constexpr (NK > 3) rd<3 * PIECE>(a[3], addr);
    if constexpr (NK > 4) rd<4 * PIECE>(a[4], addr);
    if constexpr (NK > 5) rd<5 * PIECE>(a[5], addr);
    step<NK, 0>(acc, a, addr, bfrag, side);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// kernel A
// ---------------------------------------------------------------------------------------------------------------
template <typename OT, int S, int NW>
__global__ __launch_bounds__(NW * 64) void fwd256_kernel(const FwdArgs a) {
  constexpr int NTHR = NW * 64, TSAMP = NW * 32, NWAVE = NW;
  constexpr int L_MASK = l_mask(NW), L_BIAS = l_bias(NW), L_STRIP = l_strip(NW), L_SMALL = l_small(NW), L_DB = l_db(NW);
  static_assert(TSAMP % S == 0, "whole rays per tile");
  typedef KA<OT> KT;
  typedef typename Op<OT>::V V;
  typedef __attribute__((address_space(1))) V GV;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = lane & 31, h = lane >> 5;
  constexpr int TR = TSAMP / S;                   // rays per tile
  // workgroups b and b + 8 share an XCD (round-robin dispatch: a speed assumption only); give each XCD a contiguous
  // range of the flat (object, tile) space so that it streams at most two objects' weight images through its L2
  const int nwg = gridDim.x;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const long T = (long)a.K * a.ntile;
  const long tau0 = T * wg / nwg, tau1 = T * (wg + 1) / nwg;
  if (tau0 >= tau1) return;

  uint32_t* s_mask = reinterpret_cast<uint32_t*>(lds + L_MASK);
  float* s_bias = reinterpret_cast<float*>(lds + L_BIAS);
  float* s_raw = reinterpret_cast<float*>(lds + L_STRIP);
  float* s_col = s_raw + TSAMP;
  float* s_small = reinterpret_cast<float*>(lds + L_SMALL);
  float* s_db = reinterpret_cast<float*>(lds + L_DB);
  char* hbuf = lds + L_HBUF + w * (KS_H * PIECE) + lane * 16;       // this lane's 16 bytes of fragment ks at + ks * PIECE
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;

  typename KT::Ring ring;
  ring.img = (const char*)a.img; ring.tiles_per_obj = a.ntile; ring.tau_end = tau1; ring.tau = tau0; ring.st = 0; ring.slot = 0;
  ring.obj = tau0 / a.ntile; ring.tile = tau0 - ring.obj * a.ntile;
  ring.wave = w; ring.nw = NW; ring.voff = lane * 16;
  ring.lds_ring = __builtin_amdgcn_readfirstlane(lds0 + L_RING);
  // prologue: stages 0 and 1
  ring.prepare(0);
  ring.template issue<0>(); ring.template issue<1>(); ring.template issue<2>(); ring.template issue<3>(); ring.template issue<4>(); ring.template issue<5>();
  ring.prepare(1);
  ring.template issue<0>(); ring.template issue<1>(); ring.template issue<2>(); ring.template issue<3>(); ring.template issue<4>(); ring.template issue<5>();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const float gs = a.grad_scale, inv_gs = 1.0f / a.grad_scale;
  T256_DECL;
  int cur_obj = -1;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f;
  float dbacc[11][3];               // d B of this lane's directions, summed over this lane's samples of the object
#pragma unroll
  for (int dd = 0; dd < 11; ++dd) dbacc[dd][0] = dbacc[dd][1] = dbacc[dd][2] = 0.f;
  float scale = 1.0f, inv1 = 0.f, inv2 = 0.f, ba = 0.f, boc0 = 0.f, boc1 = 0.f, boc2 = 0.f;

  auto flush_object = [&]() {       // partial d B and loss terms of (cur_obj, this workgroup)
    __syncthreads();
    float* pw = s_db + w * 68;
#pragma unroll
    for (int dd = 0; dd < 11; ++dd) {        // each 32-lane half owns its directions: sum over its lanes (samples)
      float g0 = dbacc[dd][0], g1 = dbacc[dd][1], g2 = dbacc[dd][2];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) { g0 += __shfl_xor(g0, d, 32); g1 += __shfl_xor(g1, d, 32); g2 += __shfl_xor(g2, d, 32); }
      if (s == 0 && (dd < 10 || h == 0)) {
        const int jd = 11 * h + dd;
        pw[3 * jd] = g0; pw[3 * jd + 1] = g1; pw[3 * jd + 2] = g2;
      }
      dbacc[dd][0] = dbacc[dd][1] = dbacc[dd][2] = 0.f;
    }
    l_d = seg_sum<64>(l_d); l_c = seg_sum<64>(l_c); l_o = seg_sum<64>(l_o);
    if (lane == 0) { pw[64] = l_d; pw[65] = l_c; pw[66] = l_o; }
    __syncthreads();
    if (tid < 67) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) v += s_db[ww * 68 + tid];
      a.part[((long)cur_obj * NWG_A + wg) * PART_FLOATS + tid] = v;
    }
    __syncthreads();
  };

  for (long tau = tau0; tau < tau1; ++tau) {
    const int k = (int)ring.obj;              // (the ring is at stage 0 of this tile)
    const long tile = ring.tile;
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
      scale = a.scale[k];
      const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
      inv1 = a.flags[0] ? 0.0f : 1.0f / (n1 + 1e-10f);         // render_rays.py:89-94 early return / :103 mean
      inv2 = a.flags[1] ? 0.0f : 1.0f / (n2 + 1e-10f);
    }
    __syncthreads();                                            // ring prologue / tables visible; previous tile done
    ba = s_small[64]; boc0 = s_small[65]; boc1 = s_small[66]; boc2 = s_small[67];

    char* ws_obj = a.ws + (long)k * a.wl.obj_bytes;
    const long sg = tile * (TSAMP / 32) + w;                    // this wave's sample group
    // fragment (tensor, k-step ks) of this wave's sample group; tensor 0..4 h1..hc, 5..9 d_pre1..5: a wave-uniform base
    // (scalar arithmetic) + this lane's 32-bit offset
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto act_base = [&](int tensor, int ks) __attribute__((always_inline)) -> GV* {
      return (GV*)(ws_obj + (long)tensor * a.wl.act_stride + (sg * KS_H + ks) * (64 * 16) + lane_off);
    };

    // ------------------------------------------------------------------ sample point of this lane (vmap.py:548-551)
    const int st_idx = 32 * w + s;                               // sample index inside the tile
    const int q = st_idx / S, si = st_idx - q * S;
    const long ray = tile * TR + q;
    const bool valid = ray < a.R;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
      const long rr = (long)k * a.R + ray;
      if (a.pts) {
        const float* p = a.pts + (rr * S + si) * 3;
        px = p[0]; py = p[1]; pz = p[2];
      } else {
        const float zz = a.z[rr * S + si];
        const float* o = a.origins + rr * 3;
        const float* d = a.dirs + rr * 3;
        px = (o[0] + d[0] * zz) - a.obj_center;
        py = (o[1] + d[1] * zz) - a.obj_center;
        pz = (o[2] + d[2] * zz) - a.obj_center;
      }
    }
    const float t0 = px / scale, t1 = py / scale, t2 = pz / scale;       // embedding.py:47
    // projections of this half's directions as revolutions (hi + lo); recomputed where needed (22 registers otherwise)
    auto project = [&](float (&vh)[11], float (&vl)[11]) __attribute__((always_inline)) {
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) {
        const int jd = min(11 * h + dd, OBJ_NDIR - 1);
        const float p = fmaf(t2, s_small[3 * jd + 2], fmaf(t1, s_small[3 * jd + 1], t0 * s_small[3 * jd]));     // :48
        const float a0 = p * OBJ_PI_F;                                                                        // :52
        const float v = a0 * OBJ_INV2PI_HI_;
        vh[dd] = v;
        vl[dd] = fmaf(a0, OBJ_INV2PI_LO_, fmaf(a0, OBJ_INV2PI_HI_, -v));
      }
    };
    // x1 fragments: slot u = 4 dd + f
    V x1f[KS_X1];
    {
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int t = 0; t < KS_X1; ++t) {
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
          x1f[t][j] = Op<OT>::cvt(v);
        }
      }
    }
    {   // x1 of this sample group -> workspace (the in-layer's and the cat layer's weight gradients read it)
      GV* xp = (GV*)(ws_obj + a.wl.off_x1 + (sg * KS_X1) * (64 * 16) + lane_off);
#pragma unroll
      for (int t = 0; t < KS_X1; ++t) xp[t * 64] = x1f[t];
    }

    V hin[KS_H];
    // fragment stores of a block are issued in the shadow of the NEXT block's MFMAs
    V pend0, pend1;
    GV* pend_ptr = nullptr;
    auto flush_pending = [&]() __attribute__((always_inline)) {
      if (pend_ptr) {      // non-temporal: 5 KB per sample stream through the L2 that also has to keep serving the weight ring
        __builtin_nontemporal_store(pend0, pend_ptr);
        __builtin_nontemporal_store(pend1, pend_ptr + 64);
        pend_ptr = nullptr;
      }
    };
    // the side work of a stage that issues CD transfers: stores after the first MFMA, one transfer every other MFMA,
    // whatever is left after the last one
    auto side_work = [&](auto cd_tag, auto nk_tag) __attribute__((always_inline)) {
      return [&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int CD = decltype(cd_tag)::value, NK = decltype(nk_tag)::value, KS = decltype(ks_tag)::value;
        if constexpr (KS == 0) flush_pending();
        // one wave issues one instruction per ~4 cycles: the side work is spread, a transfer behind every other MFMA
        // (transfer i after MFMA 2 i + 1), so that no gap between two MFMAs holds more than the MFMA's own 32 cycles;
        // whatever has no such slot goes after the last MFMA
        if constexpr (KS == NK - 1) {
          if constexpr (1 >= NK - 1 && 0 < CD) ring.template issue<0>();
          if constexpr (3 >= NK - 1 && 1 < CD) ring.template issue<1>();
          if constexpr (5 >= NK - 1 && 2 < CD) ring.template issue<2>();
          if constexpr (7 >= NK - 1 && 3 < CD) ring.template issue<3>();
          if constexpr (9 >= NK - 1 && 4 < CD) ring.template issue<4>();
          if constexpr (11 >= NK - 1 && 5 < CD) ring.template issue<5>();
        } else if constexpr ((KS & 1) && KS / 2 < CD) ring.template issue<KS / 2>();
      };
    };
    auto reload = [&]() __attribute__((always_inline)) {          // hin <- the fragments the layer just produced
#pragma unroll
      for (int ks = 0; ks < KS_H; ++ks) hin[ks] = *reinterpret_cast<const V*>(hbuf + ks * PIECE);
    };
    auto ring_addr = [&]() __attribute__((always_inline)) -> uint32_t {
      return lds0 + L_RING + lane * 16 + ring.slot * RING_SLOT;
    };

    // forward hidden layer: NK k-steps from bsel(ks); blocks 0..7 -> fragments (LDS hand-off buffer + workspace tensor
    // `layer`), ReLU bits.  The block loop is ROLLED: one body per layer keeps the tile body inside the instruction cache.
    auto fwd_layer = [&](auto seq_tag, auto nk_tag, const int layer, auto&& bsel) __attribute__((always_inline)) {
      constexpr int NK = decltype(nk_tag)::value;
      constexpr int CD = KT::dma_count(decltype(seq_tag)::value, NW);
      uint32_t mprev = 0;
#pragma unroll 1
      for (int blk = 0; blk < 8; ++blk) {
        T256(5);
        ring.template prepare2<decltype(seq_tag)::value>(blk);
        f32x16 acc;
        const float* bp = s_bias + layer * 256 + blk * 32 + h * 16;
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
          const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bp + 4 * n4);
          acc[4 * n4] = b4[0]; acc[4 * n4 + 1] = b4[1]; acc[4 * n4 + 2] = b4[2]; acc[4 * n4 + 3] = b4[3];
        }
        T256(0);
        auto sd = side_work(std::integral_constant<int, CD>{}, nk_tag);
        KT::template block_mma<NK>(acc, ring_addr(), bsel, sd);
        T256(1);
        uint32_t wd[8], bits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          wd[i] = pk_relu(pk_cvt<OT>(acc[2 * i], acc[2 * i + 1]));
          bits |= pk_nonzero(wd[i]) << i;            // bit i: value 2 i, bit 16 + i: value 2 i + 1
        }
        const uint4 u0 = make_uint4(wd[0], wd[1], wd[2], wd[3]), u1 = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
        *reinterpret_cast<V*>(hbuf + (2 * blk) * PIECE) = f0;
        *reinterpret_cast<V*>(hbuf + (2 * blk + 1) * PIECE) = f1;
        pend0 = f0; pend1 = f1; pend_ptr = act_base(layer, 2 * blk);
        if (blk & 1) s_mask[(layer * 4 + (blk >> 1)) * NTHR + tid] = mprev | (bits << 8);
        mprev = bits;
        T256(2);
        KT::stage_sync(ring.cnt);
        T256(3);
        ring.advance();
      }
    };
    // ------------------------------------------------------------------ forward
    fwd_layer(std::integral_constant<int, F1>{}, std::integral_constant<int, 6>{}, 0, [&](int ks) -> V { return x1f[ks]; });   // h1
    reload();
    fwd_layer(std::integral_constant<int, F2>{}, std::integral_constant<int, 16>{}, 1, [&](int ks) -> V { return hin[ks]; });  // h2
    reload();
    fwd_layer(std::integral_constant<int, F3>{}, std::integral_constant<int, 22>{}, 2,
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : x1f[ks - KS_H]; });                                              // h3
    reload();
    fwd_layer(std::integral_constant<int, F4>{}, std::integral_constant<int, 16>{}, 3, [&](int ks) -> V { return hin[ks]; });  // h4
    reload();
    // x2 fragments (octaves 4, 5): slot u = 2 dd + (f - 4)
    V x2f[KS_X2];
    {
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int t = 0; t < KS_X2; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = 8 * t + j, dd = u >> 1, f = 4 + (u & 1);
          float v = 0.0f;
          if (dd < 10) v = rev_sin(vh[dd], vl[dd], (float)(1 << f));
          else if (dd == 10) v = h == 0 ? rev_sin(vh[10], vl[10], (float)(1 << f)) : 0.0f;
          x2f[t][j] = Op<OT>::cvt(v);
        }
      }
    }
    {
      GV* xp = (GV*)(ws_obj + a.wl.off_x2 + (sg * KS_X2) * (64 * 16) + lane_off);
#pragma unroll
      for (int t = 0; t < KS_X2; ++t) xp[t * 64] = x2f[t];
    }
    fwd_layer(std::integral_constant<int, F5>{}, std::integral_constant<int, 19>{}, 4,
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : x2f[ks - KS_H]; });                                              // hc
    {   // F5 block 8: row 0 = w_alpha . h4  (raw density, model.py:81)
      constexpr int CD = KT::dma_count(F5, NW);
      ring.template prepare2<F5>(8);
      f32x16 acc = zero16();
      auto sd = side_work(std::integral_constant<int, CD>{}, std::integral_constant<int, 19>{});
      KT::template block_mma<19>(acc, ring_addr(), [&](int ks) -> V { return ks < KS_H ? hin[ks] : x2f[ks - KS_H]; }, sd);
      if (h == 0) s_raw[st_idx] = acc[0] + ba;
      KT::stage_sync(ring.cnt);
      ring.advance();
    }
    reload();                                  // hin = hc
    {   // F6: colour head on hc (model.py:95)
      constexpr int CD = KT::dma_count(F6, NW);
      ring.template prepare2<F6>(0);
      f32x16 acc = zero16();
      auto sd = side_work(std::integral_constant<int, CD>{}, std::integral_constant<int, 16>{});
      KT::template block_mma<16>(acc, ring_addr(), [&](int ks) -> V { return hin[ks]; }, sd);
      if (h == 0) { s_col[st_idx] = acc[0] + boc0; s_col[TSAMP + st_idx] = acc[1] + boc1; s_col[2 * TSAMP + st_idx] = acc[2] + boc2; }
      KT::stage_sync(ring.cnt);           // (also publishes the strips to the compositing waves)
      ring.advance();
    }
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
        const long rr = (long)k * a.R + (on ? rayq : 0);
        float gtd = 0.f, gr = 0.f, gg = 0.f, gb = 0.f;
        int lab = 2;
        if (on) { gtd = a.gt_depth[rr]; gr = a.gt_rgb[rr * 3]; gg = a.gt_rgb[rr * 3 + 1]; gb = a.gt_rgb[rr * 3 + 2]; lab = a.labels[rr]; }
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
            zz[e] = a.z[rr * S + li * SPL + e];
          }
          occ[e] = on ? sigmoid_acc(al) : 0.0f;                         // render_rays.py:13
          fr[e] = on ? (1.0f - occ[e]) + 1e-10f : 1.0f;                 // :38
          lp *= fr[e];
        }
        // exclusive product over the lanes of the ray, then inside the lane
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
        // suffix sums: lanes after this one, then samples after e inside the lane
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
    __syncthreads();
    // head gradient fragment: slot (0, 0) = d raw, (0, 1 + c) = d colour pre-activation; everything times the scale
    V dh = KT::zero_frag();
    if (h == 0 && valid) {
      dh[0] = Op<OT>::cvt(s_raw[st_idx] * gs);
      dh[1] = Op<OT>::cvt(s_col[st_idx] * gs);
      dh[2] = Op<OT>::cvt(s_col[TSAMP + st_idx] * gs);
      dh[3] = Op<OT>::cvt(s_col[2 * TSAMP + st_idx] * gs);
    }
    *(GV*)(ws_obj + a.wl.off_dhead + sg * (64 * 16) + lane_off) = dh;

    // ------------------------------------------------------------------ backward
    // hidden input-gradient layer: d(input features) from NK k-steps of bsel, masked by the ReLU bits of the PRODUCING
    // layer `mlayer`, stored as the weight-gradient operand `5 + mlayer` and handed to the next GEMM through hbuf
    auto bwd_layer = [&](auto seq_tag, auto nk_tag, const int mlayer, auto&& bsel) __attribute__((always_inline)) {
      constexpr int NK = decltype(nk_tag)::value;
      constexpr int CD = KT::dma_count(decltype(seq_tag)::value, NW);
#pragma unroll 1
      for (int blk = 0; blk < 8; ++blk) {
        T256(5);
        ring.template prepare2<decltype(seq_tag)::value>(blk);
        const uint32_t bits = s_mask[(mlayer * 4 + (blk >> 1)) * NTHR + tid] >> (8 * (blk & 1));
        f32x16 acc = zero16();
        auto sd = side_work(std::integral_constant<int, CD>{}, nk_tag);
        KT::template block_mma<NK>(acc, ring_addr(), bsel, sd);
        uint32_t wd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wd[i] = pk_mask(pk_cvt<OT>(acc[2 * i], acc[2 * i + 1]), (bits >> i) & 0x00010001u);
        const uint4 u0 = make_uint4(wd[0], wd[1], wd[2], wd[3]), u1 = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
        *reinterpret_cast<V*>(hbuf + (2 * blk) * PIECE) = f0;
        *reinterpret_cast<V*>(hbuf + (2 * blk + 1) * PIECE) = f1;
        pend0 = f0; pend1 = f1; pend_ptr = act_base(5 + mlayer, 2 * blk);
        KT::stage_sync(ring.cnt);
        ring.advance();
        T256(4);
      }
    };
    // slot-gradient blocks (d x1 / d x2): NB blocks accumulated into xacc[b]
    auto bwd_slots = [&](auto seq_tag, auto nb_tag, f32x16* xacc) __attribute__((always_inline)) {
      constexpr int NB = decltype(nb_tag)::value;
      constexpr int CD = KT::dma_count(decltype(seq_tag)::value, NW);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        ring.template prepare2<decltype(seq_tag)::value>(b);
        auto sd = side_work(std::integral_constant<int, CD>{}, std::integral_constant<int, 16>{});
        KT::template block_mma<16>(xacc[b], ring_addr(), [&](int ks) -> V { return hin[ks]; }, sd);
        KT::stage_sync(ring.cnt);
        ring.advance();
      }
    };
    float dproj[11];
#pragma unroll
    for (int dd = 0; dd < 11; ++dd) dproj[dd] = 0.f;

    bwd_layer(std::integral_constant<int, B6>{}, std::integral_constant<int, 1>{}, 4, [&](int) -> V { return dh; });          // d hc
    reload();                                                                                                                  // hin = d hc
    bwd_layer(std::integral_constant<int, B5H>{}, std::integral_constant<int, 17>{}, 3,
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : dh; });                                                          // d h4
    {                                                                                                                          // B5X: d x2
      f32x16 xa[2] = {zero16(), zero16()};
      bwd_slots(std::integral_constant<int, B5X>{}, std::integral_constant<int, 2>{}, xa);
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int u = 0; u < 22; ++u) {                        // slot u = 2 dd + (f - 4)
        const int dd = u >> 1, f = 4 + (u & 1);
        float sv, cv;
        rev_sincos(vh[dd], vl[dd], (float)(1 << f), sv, cv);
        dproj[dd] = fmaf(xa[u >> 4][u & 15], (cv * OBJ_PI_F) * (float)(1 << f), dproj[dd]);     // embedding.py:49-52
      }
    }
    reload();                                                                                                                  // hin = d h4
    bwd_layer(std::integral_constant<int, B4>{}, std::integral_constant<int, 16>{}, 2, [&](int ks) -> V { return hin[ks]; });  // d h3
    reload();
    bwd_layer(std::integral_constant<int, B3H>{}, std::integral_constant<int, 16>{}, 1, [&](int ks) -> V { return hin[ks]; }); // d h2
    auto pe_bwd_x1 = [&](const f32x16* x1a) __attribute__((always_inline)) {
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int u = 0; u < 44; ++u) {                          // slot u = 4 dd + f
        const int dd = u >> 2, f = u & 3;
        float sv, cv;
        rev_sincos(vh[dd], vl[dd], (float)(1 << f), sv, cv);
        dproj[dd] = fmaf(x1a[u >> 4][u & 15], (cv * OBJ_PI_F) * (float)(1 << f), dproj[dd]);
      }
    };
    {                                                                                                                          // B3X: d x1
      f32x16 x1a[3] = {zero16(), zero16(), zero16()};
      bwd_slots(std::integral_constant<int, B3X>{}, std::integral_constant<int, 3>{}, x1a);
      pe_bwd_x1(x1a);          // (the chain rule is linear in d x1: applied per contribution)
    }
    reload();                                                                                                                  // hin = d h2
    bwd_layer(std::integral_constant<int, B2>{}, std::integral_constant<int, 16>{}, 0, [&](int ks) -> V { return hin[ks]; });  // d h1
    reload();
    {                                                                                                                          // B1: d x1 +=
      f32x16 x1a[3] = {zero16(), zero16(), zero16()};
      bwd_slots(std::integral_constant<int, B1>{}, std::integral_constant<int, 3>{}, x1a);
      pe_bwd_x1(x1a);
    }
    flush_pending();
    // d B[j][c] += d proj_j * t_c: per-lane sums, reduced over the lanes once per object (flush_object)
#pragma unroll
    for (int dd = 0; dd < 11; ++dd) {
      const bool has = valid && (dd < 10 || h == 0);
      const float dp = has ? dproj[dd] * inv_gs : 0.0f;
      dbacc[dd][0] = fmaf(dp, t0, dbacc[dd][0]);
      dbacc[dd][1] = fmaf(dp, t1, dbacc[dd][1]);
      dbacc[dd][2] = fmaf(dp, t2, dbacc[dd][2]);
    }
  }
  flush_object();
#ifdef OBJ256_TIMING
  T256(5);
  if (blockIdx.x == 0 && tid == 0)
    printf("t256 fwd-layer stages: issue %llu  mma %llu  epilogue %llu  sync %llu  | bwd-layer stages %llu | everything else %llu\n",
           tm_[0], tm_[1], tm_[2], tm_[3], tm_[4], tm_[5]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// kernel B: weight gradients.  Task types (A rows x B columns, both read back from kernel A's fragments):
//   0 L2   d_pre2 x h1                 2 L4  d_pre4 x h3                                   (8 x 8 blocks)
//   1 L3   d_pre3 x [h2 | x1]                                                              (8 x 11: one pass over d_pre3)
//   3 L5   [d_pre5 ; d_head] x [h4 | x2]      (8 x 10, + the head row block x h4: one pass over d_pre5 and h4)
//   4 L1   d_pre1 x x1                                                                     (8 x 3)
//   5 col  d_head x hc                                                                     (1 x 8)
// Every task also yields the row sums of its A operand (bias gradients).  A workgroup takes one (object, type, part)
// = a range of sample groups, and writes its partial tiles [.][32][32] + row sums into its slab.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NTYPE = 6;
__host__ __device__ constexpr int type_tiles(int t) { return t == 1 ? 88 : t == 3 ? 88 : t == 4 ? 24 : t == 5 ? 8 : 64; }
__host__ __device__ constexpr int type_rows(int t) { return t == 3 ? 9 * 32 : t == 5 ? 32 : 8 * 32; }      // row sums
__host__ __device__ constexpr int type_slab(int t) { return type_tiles(t) * 1024 + type_rows(t); }
__host__ __device__ constexpr int type_pieces(int t) {      // 1-KB pieces per sample group
  return t == 1 ? 16 + 16 + KS_X1 : t == 3 ? 16 + 1 + 16 + KS_X2 : t == 4 ? 16 + KS_X1 : t == 5 ? 1 + 16 : 32;
}
struct WgArgs {
  int K;
  int parts[NTYPE];
  int prefix[NTYPE + 1];        // workgroups per object before type t
  long slab_prefix[NTYPE + 1];  // floats per object before type t's slabs
  const char* ws; float* slabs;
  WsLay wl;
};
constexpr int IMG_PITCH = 72;                       // bytes per sample row of a 32-feature block image
constexpr int IMG_BLK = 32 * IMG_PITCH;             // 2304

__device__ __forceinline__ uint2 lds_tr16(const uint32_t addr) {
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

// RB / CB: row / column blocks of the task; NRG x NCG waves; AK / BK: 0 hidden fragments (2 per block), 1 head, 2 x slots
template <typename OT, int RB, int CB, int NRG, int NCG, int AK, int BK, int KSB>
__device__ __forceinline__ void wgrad_task(const WgArgs& a, const char* Asrc, const char* Bsrc, const long sg0, const long sg1,
                                           float* slab, char* lds) {
  typedef typename Op<OT>::V V;
  constexpr int RBW = RB / NRG, CBW = CB / NCG;
  constexpr int NPA = AK == 0 ? 2 * RB : 1, NPB = BK == 0 ? 2 * CB : KSB;
  constexpr int CH = 2;                                        // sample groups per chunk
  constexpr int NP = CH * (NPA + NPB);
  constexpr int PPW = (NP + NWAVE - 1) / NWAVE;                // pieces per wave and chunk
  constexpr int BUF = CH * (RB + CB) * IMG_BLK;
  static_assert(2 * BUF <= 163840, "LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = w % NRG, cg = w / NRG;
  const int s = lane & 31, h = lane >> 5;
  f32x16 acc[RBW][CBW];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int j = 0; j < CBW; ++j) acc[i][j] = zero16();
  float rs[RBW];
#pragma unroll
  for (int i = 0; i < RBW; ++i) rs[i] = 0.f;

  uint4 stg[PPW];
  auto load_chunk = [&](const long sgc) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = w + NWAVE * i;
      stg[i] = make_uint4(0, 0, 0, 0);
      if (p < NP) {
        const int sgi = p / (NPA + NPB), pp = p - sgi * (NPA + NPB);
        const long sg = sgc + sgi;
        if (sg < sg1) {
          const char* src = pp < NPA ? Asrc + ((sg * NPA + pp) * 64 + lane) * 16 : Bsrc + ((sg * NPB + (pp - NPA)) * 64 + lane) * 16;
          stg[i] = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
  };
  auto store_chunk = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = w + NWAVE * i;
      if (p < NP) {
        const int sgi = p / (NPA + NPB), pp = p - sgi * (NPA + NPB);
        const bool isA = pp < NPA;
        const int kind = isA ? AK : BK;
        const int ks = isA ? pp : pp - NPA;
        char* base = buf + (sgi * (RB + CB) + (isA ? 0 : RB)) * IMG_BLK + s * IMG_PITCH;
        const uint2 lo = make_uint2(stg[i].x, stg[i].y), hi = make_uint2(stg[i].z, stg[i].w);
        if (kind == 0) {             // hidden fragment: features 16 sub + 4 h + 0..3 and 16 sub + 8 + 4 h + 0..3 of block ks >> 1
          char* d = base + (ks >> 1) * IMG_BLK + (16 * (ks & 1) + 4 * h) * 2;
          *reinterpret_cast<uint2*>(d) = lo;
          *reinterpret_cast<uint2*>(d + 16) = hi;
        } else {                     // head / x slots: column 16 (t & 1) + 8 h + j of block t >> 1
          char* d = base + (ks >> 1) * IMG_BLK + (16 * (ks & 1) + 8 * h) * 2;
          *reinterpret_cast<uint2*>(d) = lo;
          *reinterpret_cast<uint2*>(d + 8) = hi;
        }
      }
    }
  };
  // operand of block image `img` for k-step kk (samples 16 kk + 8 h + 0..7): lane (c = l & 31) <- column c
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const int grp = (lane >> 4) & 1, li = lane & 15;
  const uint32_t tr_lane = (uint32_t)(((li >> 2) + 8 * h) * IMG_PITCH + (16 * grp + 4 * (li & 3)) * 2);

  load_chunk(sg0);
  store_chunk(lds);
  __syncthreads();
  int cur = 0;
  for (long sgc = sg0; sgc < sg1; sgc += CH) {
    const bool more = sgc + CH < sg1;
    if (more) load_chunk(sgc + CH);
    const uint32_t buf = lds0 + cur * BUF;
#pragma unroll
    for (int sgi = 0; sgi < CH; ++sgi) {
      if (sgc + sgi < sg1) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          uint2 alo[RBW], ahi[RBW], blo[CBW], bhi[CBW];
          const uint32_t rowoff = buf + sgi * (RB + CB) * IMG_BLK + kk * 16 * IMG_PITCH + tr_lane;
#pragma unroll
          for (int i = 0; i < RBW; ++i) {
            const uint32_t ad = rowoff + (rg * RBW + i) * IMG_BLK;
            alo[i] = lds_tr16(ad); ahi[i] = lds_tr16(ad + 4 * IMG_PITCH);
          }
#pragma unroll
          for (int j = 0; j < CBW; ++j) {
            const uint32_t ad = rowoff + (RB + cg * CBW + j) * IMG_BLK;
            blo[j] = lds_tr16(ad); bhi[j] = lds_tr16(ad + 4 * IMG_PITCH);
          }
          // one wait for the whole batch, tied to the registers the reads write
          if constexpr (RBW == 2 && CBW == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(blo[0]), "+v"(bhi[0]),
                         "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3])::"memory");
          else if constexpr (CBW == 3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]),
                         "+v"(blo[2]), "+v"(bhi[2])::"memory");
          else if constexpr (CBW == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1])::"memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(blo[0]), "+v"(bhi[0])::"memory");
          __builtin_amdgcn_sched_barrier(0);
          V af[RBW], bf[CBW];
#pragma unroll
          for (int i = 0; i < RBW; ++i) {
            const uint4 u = make_uint4(alo[i].x, alo[i].y, ahi[i].x, ahi[i].y);
            af[i] = *reinterpret_cast<const V*>(&u);
            if (cg == 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) rs[i] += (float)af[i][j];
            }
          }
#pragma unroll
          for (int j = 0; j < CBW; ++j) {
            const uint4 u = make_uint4(blo[j].x, blo[j].y, bhi[j].x, bhi[j].y);
            bf[j] = *reinterpret_cast<const V*>(&u);
          }
#pragma unroll
          for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int j = 0; j < CBW; ++j) acc[i][j] = Op<OT>::mfma(af[i], bf[j], acc[i][j]);
        }
      }
    }
    if (more) store_chunk(lds + (cur ^ 1) * BUF);
    __syncthreads();
    cur ^= 1;
  }
  // partial tiles -> slab: tile (rb, cb) at (rb * CB + cb) * 1024, element [row][col]
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int j = 0; j < CBW; ++j) {
      float* t = slab + ((rg * RBW + i) * CB + (cg * CBW + j)) * 1024;
#pragma unroll
      for (int n = 0; n < 16; ++n) t[acc_row(n, h) * 32 + s] = acc[i][j][n];
    }
  if (cg == 0) {
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
      const float v = rs[i] + __shfl_xor(rs[i], 32, 64);
      if (h == 0) slab[RB * CB * 1024 + (rg * RBW + i) * 32 + s] = v;
    }
  }
}

// The concatenated layers in ONE pass over their d_pre: rows = the 8 blocks of d_pre (wave w owns block w), columns =
// [hidden input (8 blocks) | x slots (CBX blocks)]; HEAD: the d_head row block x the hidden input on top (wave w: its
// column block w) -- the density head's weight gradient shares the read of h4.  One sample group per chunk.
template <typename OT, int KSX, bool HEAD>
__device__ __forceinline__ void wgrad_cat(const char* Asrc, const char* Hsrc, const char* Bsrc, const char* Xsrc, const long sg0,
                                          const long sg1, float* slab, char* lds) {
  typedef typename Op<OT>::V V;
  constexpr int CBX = (KSX + 1) / 2, CB = 8 + CBX;
  constexpr int NA = 8 + (HEAD ? 1 : 0);                       // block images of the A side
  constexpr int NP = 16 + (HEAD ? 1 : 0) + 16 + KSX;           // pieces per sample group
  constexpr int PPW = (NP + NWAVE - 1) / NWAVE;
  constexpr int BUF = (NA + CB) * IMG_BLK;
  static_assert(2 * BUF <= 163840, "LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = lane & 31, h = lane >> 5;
  f32x16 acc[CB], hacc = zero16();
#pragma unroll
  for (int j = 0; j < CB; ++j) acc[j] = zero16();
  float rs = 0.f, rsh = 0.f;

  uint4 stg[PPW];
  auto load_chunk = [&](const long sg) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = w + NWAVE * i;
      stg[i] = make_uint4(0, 0, 0, 0);
      if (p < NP && sg < sg1) {
        const char* src;
        if (p < 16) src = Asrc + ((sg * 16 + p) * 64 + lane) * 16;
        else if (HEAD && p == 16) src = Hsrc + (sg * 64 + lane) * 16;
        else if (p < NP - KSX) src = Bsrc + ((sg * 16 + (p - (NP - KSX - 16))) * 64 + lane) * 16;
        else src = Xsrc + ((sg * KSX + (p - (NP - KSX))) * 64 + lane) * 16;
        stg[i] = *reinterpret_cast<const uint4*>(src);
      }
    }
  };
  auto store_chunk = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = w + NWAVE * i;
      if (p < NP) {
        const uint2 lo = make_uint2(stg[i].x, stg[i].y), hi = make_uint2(stg[i].z, stg[i].w);
        int blk, ks; bool hidden;
        if (p < 16) { blk = p >> 1; ks = p; hidden = true; }
        else if (HEAD && p == 16) { blk = 8; ks = 0; hidden = false; }
        else if (p < NP - KSX) { ks = p - (NP - KSX - 16); blk = NA + (ks >> 1); hidden = true; }
        else { ks = p - (NP - KSX); blk = NA + 8 + (ks >> 1); hidden = false; }
        char* base = buf + blk * IMG_BLK + s * IMG_PITCH;
        if (hidden) {
          char* d = base + (16 * (ks & 1) + 4 * h) * 2;
          *reinterpret_cast<uint2*>(d) = lo;
          *reinterpret_cast<uint2*>(d + 16) = hi;
        } else {
          char* d = base + (16 * (ks & 1) + 8 * h) * 2;
          *reinterpret_cast<uint2*>(d) = lo;
          *reinterpret_cast<uint2*>(d + 8) = hi;
        }
      }
    }
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const int grp = (lane >> 4) & 1, li = lane & 15;
  const uint32_t tr_lane = (uint32_t)(((li >> 2) + 8 * h) * IMG_PITCH + (16 * grp + 4 * (li & 3)) * 2);
  auto frag = [&](const uint32_t ad) __attribute__((always_inline)) -> V {
    uint2 lo = lds_tr16(ad), hi = lds_tr16(ad + 4 * IMG_PITCH);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)::"memory");
    const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return *reinterpret_cast<const V*>(&u);
  };

  load_chunk(sg0);
  store_chunk(lds);
  __syncthreads();
  int cur = 0;
  for (long sg = sg0; sg < sg1; ++sg) {
    const bool more = sg + 1 < sg1;
    if (more) load_chunk(sg + 1);
    const uint32_t buf = lds0 + cur * BUF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t rowoff = buf + kk * 16 * IMG_PITCH + tr_lane;
      // the A operand (and the head rows), then the columns four at a time (the tr reads of a batch share one wait;
      // all eleven at once do not fit the 256 registers next to 176 accumulators)
      uint2 alo = lds_tr16(rowoff + w * IMG_BLK), ahi = lds_tr16(rowoff + w * IMG_BLK + 4 * IMG_PITCH);
      uint2 hlo = alo, hhi = ahi;
      if (HEAD) { hlo = lds_tr16(rowoff + 8 * IMG_BLK); hhi = lds_tr16(rowoff + 8 * IMG_BLK + 4 * IMG_PITCH); }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo), "+v"(ahi), "+v"(hlo), "+v"(hhi)::"memory");
      const uint4 ua = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
      const V af = *reinterpret_cast<const V*>(&ua);
#pragma unroll
      for (int j = 0; j < 8; ++j) rs += (float)af[j];
#pragma unroll
      for (int j0 = 0; j0 < CB; j0 += 4) {
        uint2 blo[4], bhi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j0 + j < CB) { blo[j] = lds_tr16(rowoff + (NA + j0 + j) * IMG_BLK); bhi[j] = lds_tr16(rowoff + (NA + j0 + j) * IMG_BLK + 4 * IMG_PITCH); }
          else { blo[j] = blo[0]; bhi[j] = bhi[0]; }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]),
                     "+v"(blo[3]), "+v"(bhi[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j0 + j < CB) {
            const uint4 ub = make_uint4(blo[j].x, blo[j].y, bhi[j].x, bhi[j].y);
            acc[j0 + j] = Op<OT>::mfma(af, *reinterpret_cast<const V*>(&ub), acc[j0 + j]);
          }
      }
      if (HEAD) {
        const uint4 uh = make_uint4(hlo.x, hlo.y, hhi.x, hhi.y);
        const V hf = *reinterpret_cast<const V*>(&uh);
        if (w == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) rsh += (float)hf[j];
        }
        // this wave's column block w of the hidden input
        V bw;
        {
          const uint32_t ad = rowoff + (NA + w) * IMG_BLK;
          uint2 lo = lds_tr16(ad), hi = lds_tr16(ad + 4 * IMG_PITCH);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)::"memory");
          const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
          bw = *reinterpret_cast<const V*>(&u);
        }
        hacc = Op<OT>::mfma(hf, bw, hacc);
      }
    }
    (void)frag;
    if (more) store_chunk(lds + (cur ^ 1) * BUF);
    __syncthreads();
    cur ^= 1;
  }
  // slab: tiles (rb = wave, cb) at (w * CB + cb) * 1024; HEAD: head tiles at (8 * CB + w) * 1024; row sums behind the tiles
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    float* t = slab + (w * CB + j) * 1024;
#pragma unroll
    for (int n = 0; n < 16; ++n) t[acc_row(n, h) * 32 + s] = acc[j][n];
  }
  constexpr int NT = 8 * CB + (HEAD ? 8 : 0);
  if (HEAD) {
    float* t = slab + (8 * CB + w) * 1024;
#pragma unroll
    for (int n = 0; n < 16; ++n) t[acc_row(n, h) * 32 + s] = hacc[n];
  }
  {
    const float v = rs + __shfl_xor(rs, 32, 64);
    if (h == 0) slab[NT * 1024 + w * 32 + s] = v;
    if (HEAD && w == 0) {
      const float vh = rsh + __shfl_xor(rsh, 32, 64);
      if (h == 0) slab[NT * 1024 + 256 + s] = vh;
    }
  }
}

template <typename OT>
__global__ __launch_bounds__(NTHR) void wgrad256_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int per_obj = a.prefix[NTYPE];
  const int k = blockIdx.x / per_obj, r = blockIdx.x - k * per_obj;
  int t = 0;
  while (t + 1 < NTYPE && r >= a.prefix[t + 1]) ++t;
  const int part = r - a.prefix[t], nparts = a.parts[t];
  const long sg0 = a.wl.nsg * part / nparts, sg1 = a.wl.nsg * (part + 1) / nparts;
  const char* ws = a.ws + (long)k * a.wl.obj_bytes;
  float* slab = a.slabs + (long)k * a.slab_prefix[NTYPE] + a.slab_prefix[t] + (long)part * type_slab(t);
  auto act = [&](int i) { return ws + (long)i * a.wl.act_stride; };
  const char* x1 = ws + a.wl.off_x1; const char* x2 = ws + a.wl.off_x2; const char* dh = ws + a.wl.off_dhead;
  switch (t) {
    case 0: wgrad_task<OT, 8, 8, 4, 2, 0, 0, 0>(a, act(6), act(0), sg0, sg1, slab, lds); break;
    case 1: wgrad_cat<OT, KS_X1, false>(act(7), dh, act(1), x1, sg0, sg1, slab, lds); break;
    case 2: wgrad_task<OT, 8, 8, 4, 2, 0, 0, 0>(a, act(8), act(2), sg0, sg1, slab, lds); break;
    case 3: wgrad_cat<OT, KS_X2, true>(act(9), dh, act(3), x2, sg0, sg1, slab, lds); break;
    case 4: wgrad_task<OT, 8, 3, 8, 1, 0, 2, KS_X1>(a, act(5), x1, sg0, sg1, slab, lds); break;
    default: wgrad_task<OT, 1, 8, 1, 8, 1, 0, 0>(a, dh, act(4), sg0, sg1, slab, lds); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// finalize: grads[k][i] = (1 / grad_scale) * sum over the parts' slabs, in part order; d B and the loss terms from
// kernel A's per-workgroup partials, in workgroup order.
// ---------------------------------------------------------------------------------------------------------------
struct FinArgs {
  int K; long P, p_stride;
  int parts[NTYPE]; long slab_prefix[NTYPE + 1];
  const float* slabs; const float* part;
  float* grads; float* loss_terms; int* status;
  float inv_gs;
  Lay256 L;
};
__device__ __forceinline__ float slab_sum(const FinArgs& a, int k, int t, long elem) {
  const float* base = a.slabs + (long)k * a.slab_prefix[NTYPE] + a.slab_prefix[t] + elem;
  float v = 0.f;
  for (int p = 0; p < a.parts[t]; ++p) v += base[(long)p * type_slab(t)];
  return v;
}
// (x1 / x2 reference column) -> (column block, column) of the slot images: block t >> 1, column 16 (t & 1) + 8 h + j
__device__ __forceinline__ int x1_col_pos(int col) {
  int h, u;
  if (col < 3) { h = 1; u = 40 + col; }
  else { const int f = (col - 3) / OBJ_NDIR, j = (col - 3) % OBJ_NDIR; h = j >= 11 ? 1 : 0; u = 4 * (j - 11 * h) + f; }
  const int t = u >> 3, jj = u & 7;
  return (t >> 1) * 1024 + 16 * (t & 1) + 8 * h + jj;          // cb * 1024 + column
}
__device__ __forceinline__ int x2_col_pos(int col) {
  const int f = col / OBJ_NDIR, j = col % OBJ_NDIR, h = j >= 11 ? 1 : 0, u = 2 * (j - 11 * h) + f;
  const int t = u >> 3, jj = u & 7;
  return (t >> 1) * 1024 + 16 * (t & 1) + 8 * h + jj;
}
__global__ __launch_bounds__(256) void finalize256_kernel(const FinArgs a) {
  const int k = blockIdx.y;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const Lay256& L = a.L;
  const int E1 = OBJ_E1, E2 = OBJ_E2;
  if (i < a.P) {
    float v = 0.f;
    bool has = true;
    // tile element: tile * 1024 + row * 32 + col; row sums behind a task's tiles
    auto tile = [&](int t, int tl, int row, int col) { return slab_sum(a, k, t, (long)tl * 1024 + row * 32 + col); };
    auto rowsum = [&](int t, int o) { return slab_sum(a, k, t, (long)type_tiles(t) * 1024 + o); };
    if (i < L.in_b) { const int o = (int)(i - L.in_w) / E1, c = (int)(i - L.in_w) % E1; const int pos = x1_col_pos(c);
                      v = tile(4, (o >> 5) * 3 + (pos >> 10), o & 31, pos & 1023); }
    else if (i < L.m1_w) v = rowsum(4, (int)(i - L.in_b));
    else if (i < L.m1_b) { const int e = (int)(i - L.m1_w), o = e / HID, c = e % HID; v = tile(0, (o >> 5) * 8 + (c >> 5), o & 31, c & 31); }
    else if (i < L.cat_w) v = rowsum(0, (int)(i - L.m1_b));
    else if (i < L.cat_b) { const int e = (int)(i - L.cat_w), o = e / (HID + E1), c = e % (HID + E1);
                            if (c < HID) v = tile(1, (o >> 5) * 11 + (c >> 5), o & 31, c & 31);
                            else { const int pos = x1_col_pos(c - HID); v = tile(1, (o >> 5) * 11 + 8 + (pos >> 10), o & 31, pos & 1023); } }
    else if (i < L.m2_w) v = rowsum(1, (int)(i - L.cat_b));
    else if (i < L.m2_b) { const int e = (int)(i - L.m2_w), o = e / HID, c = e % HID; v = tile(2, (o >> 5) * 8 + (c >> 5), o & 31, c & 31); }
    else if (i < L.a_w) v = rowsum(2, (int)(i - L.m2_b));
    else if (i < L.a_b) { const int c = (int)(i - L.a_w); v = tile(3, 80 + (c >> 5), 0, c & 31); }
    else if (i < L.cl_w) v = rowsum(3, 256);
    else if (i < L.cl_b) { const int e = (int)(i - L.cl_w), o = e / (HID + E2), c = e % (HID + E2);
                           if (c < HID) v = tile(3, (o >> 5) * 10 + (c >> 5), o & 31, c & 31);
                           else { const int pos = x2_col_pos(c - HID); v = tile(3, (o >> 5) * 10 + 8 + (pos >> 10), o & 31, pos & 1023); } }
    else if (i < L.oc_w) v = rowsum(3, (int)(i - L.cl_b));
    else if (i < L.oc_b) { const int e = (int)(i - L.oc_w), ch = e / HID, c = e % HID; v = tile(5, c >> 5, 1 + ch, c & 31); }
    else if (i < L.oc_b + 3) v = rowsum(5, 1 + (int)(i - L.oc_b));
    else if (i >= L.pe_b && i < L.pe_b + 63) {
      float sdb = 0.f;
      for (int g = 0; g < NWG_A; ++g) sdb += a.part[((long)k * NWG_A + g) * PART_FLOATS + (i - L.pe_b)];
      a.grads[(long)k * a.p_stride + i] = sdb;          // (kernel A already removed the gradient scale)
      has = false;
    } else has = false;                                  // feature branch: no gradient without gt_feat
    if (has) a.grads[(long)k * a.p_stride + i] = v * a.inv_gs;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    float sl = 0.f;
    if (threadIdx.x < 3)
      for (int g = 0; g < NWG_A; ++g) sl += a.part[((long)k * NWG_A + g) * PART_FLOATS + 64 + threadIdx.x];
    a.loss_terms[k * 4 + threadIdx.x] = sl;
    if (sl > 100000.0f) atomicOr(a.status, 1);           // render_rays.py:109-111
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
  WsLay wl;
  int parts[NTYPE]; int prefix[NTYPE + 1]; long slab_prefix[NTYPE + 1];
  size_t off_img, off_part, off_slabs, off_ws, bytes;
};
#ifndef OBJ256_NW
#define OBJ256_NW 4          // waves per workgroup of kernel A where the ray length allows (S <= 32 NW)
#endif
static int waves_for(int) { return OBJ256_NW; }
static Plan make_plan(int K, long n, int S) {
  Plan p;
  p.wl = WsLay::make(n, 32 * waves_for(S));
  const long target = 48L << 20;                      // bytes one weight-gradient workgroup streams
  p.prefix[0] = 0; p.slab_prefix[0] = 0;
  for (int t = 0; t < NTYPE; ++t) {
    const long bytes = p.wl.nsg * type_pieces(t) * PIECE;
    long parts = (bytes + target - 1) / target;
    if (parts < 1) parts = 1;
    if (parts > p.wl.nsg / 2) parts = p.wl.nsg / 2 > 0 ? p.wl.nsg / 2 : 1;
    p.parts[t] = (int)parts;
    p.prefix[t + 1] = p.prefix[t] + (int)parts;
    p.slab_prefix[t + 1] = p.slab_prefix[t] + parts * type_slab(t);
  }
  size_t o = 0;
  p.off_img = o; o += al256((size_t)K * IMG_BYTES);
  p.off_part = o; o += al256((size_t)K * NWG_A * PART_FLOATS * 4);
  p.off_slabs = o; o += al256((size_t)K * p.slab_prefix[NTYPE] * 4);
  p.off_ws = o; o += al256((size_t)K * p.wl.obj_bytes);
  p.bytes = o;
  return p;
}

bool applicable(const objnerf_net* net, const objnerf_train_args* a) {
  if (net->hidden != HID || net->n_freqs != 6) return false;
  if (!(a->mode & (OBJNERF_TRAIN_BF16 | OBJNERF_TRAIN_FP16))) return false;
  if (a->gt_feat || a->relu_masks || a->emb_debug) return false;
  const int S = a->S;
  return S == 32 || S == 64 || S == 128;
}
size_t workspace_bytes(int K, int R, int S) { return make_plan(K, (long)R * S, S).bytes + 256; }

template <typename OT, int S, int NW>
static void launch_fwd(const FwdArgs& fa, hipStream_t st) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)fwd256_kernel<OT, S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, l_total(NW));
  });
  hipLaunchKernelGGL((fwd256_kernel<OT, S, NW>), dim3(NWG_A), dim3(NW * 64), l_total(NW), st, fa);
}
template <typename OT>
static int run(const objnerf_net* net, const objnerf_train_args* a, hipStream_t st) {
  const int K = a->K;
  const long n = (long)a->R * a->S;
  const Plan p = make_plan(K, n, a->S);
  if (a->workspace_bytes < p.bytes) return OBJNERF_EINVAL;
  char* base = (char*)a->workspace;
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  Lay256 L;
  L.in_w = (int)off[0]; L.in_b = (int)off[1]; L.m1_w = (int)off[2]; L.m1_b = (int)off[3]; L.cat_w = (int)off[4];
  L.cat_b = (int)off[5]; L.m2_w = (int)off[6]; L.m2_b = (int)off[7]; L.a_w = (int)off[8]; L.a_b = (int)off[9];
  L.cl_w = (int)off[10]; L.cl_b = (int)off[11]; L.oc_w = (int)off[12]; L.oc_b = (int)off[13]; L.pe_b = (int)off[18];
  OT* img = (OT*)(base + p.off_img);
  float* part = (float*)(base + p.off_part);
  float* slabs = (float*)(base + p.off_slabs);
  char* ws = base + p.off_ws;
  (void)hipMemsetAsync(a->status, 0, sizeof(int), st);
  (void)hipMemsetAsync(part, 0, (size_t)K * NWG_A * PART_FLOATS * 4, st);
  {
    const long tot = (long)K * N_PIECES * 64;
    hipLaunchKernelGGL((pack256_kernel<OT>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, K, a->params,
                       (long)a->p_stride, L, img);
  }
  const float gs = (a->mode & OBJNERF_TRAIN_FP16) ? exp2f(floorf(log2f((float)a->R)) + 3.0f) : 1.0f;
  FwdArgs fa;
  fa.K = K; fa.R = a->R; fa.S = a->S; fa.TR = 0;
  const int nw = waves_for(a->S);
  const int tsamp = 32 * nw;
  fa.ntile = (n + tsamp - 1) / tsamp;
  (void)nw;
  fa.color_scaling = a->color_scaling; fa.opacity_scaling = a->opacity_scaling; fa.obj_center = a->obj_center;
  fa.grad_scale = gs;
  fa.params = a->params; fa.p_stride = a->p_stride; fa.scale = a->scale;
  fa.pts = a->pts; fa.origins = a->origins; fa.dirs = a->dirs; fa.z = a->z;
  fa.gt_depth = a->gt_depth; fa.gt_rgb = a->gt_rgb; fa.labels = a->labels;
  fa.counts = a->counts; fa.flags = a->flags;
  fa.img = img; fa.ws = ws; fa.part = part; fa.L = L; fa.wl = p.wl;
#ifdef OBJ256_ONE          // diagnostic builds: one instantiation (compile time)
  launch_fwd<OT, 128, OBJ256_NW>(fa, st);
#else
  switch (a->S) {
    case 32: launch_fwd<OT, 32, OBJ256_NW>(fa, st); break;
    case 64: launch_fwd<OT, 64, OBJ256_NW>(fa, st); break;
    default: launch_fwd<OT, 128, OBJ256_NW>(fa, st); break;
  }
#endif
  WgArgs wa;
  wa.K = K;
  for (int t = 0; t < NTYPE; ++t) wa.parts[t] = p.parts[t];
  for (int t = 0; t <= NTYPE; ++t) { wa.prefix[t] = p.prefix[t]; wa.slab_prefix[t] = p.slab_prefix[t]; }
  wa.ws = ws; wa.slabs = slabs; wa.wl = p.wl;
  constexpr int WG_LDS = 2 * 2 * 16 * IMG_BLK;            // the widest task: 2 buffers x 2 sample groups x 16 block images
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)wgrad256_kernel<OT>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);
  });
  hipLaunchKernelGGL((wgrad256_kernel<OT>), dim3((unsigned)(K * p.prefix[NTYPE])), dim3(NTHR), WG_LDS, st, wa);
  FinArgs fn;
  fn.K = K; fn.P = off[OBJNERF_N_TENSORS]; fn.p_stride = a->p_stride;
  for (int t = 0; t < NTYPE; ++t) fn.parts[t] = p.parts[t];
  for (int t = 0; t <= NTYPE; ++t) fn.slab_prefix[t] = p.slab_prefix[t];
  fn.slabs = slabs; fn.part = part; fn.grads = a->grads; fn.loss_terms = a->loss_terms; fn.status = a->status;
  fn.inv_gs = 1.0f / gs; fn.L = L;
  hipLaunchKernelGGL(finalize256_kernel, dim3((unsigned)((fn.P + 255) / 256), (unsigned)K), dim3(256), 0, st, fn);
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream) {
#ifndef OBJ256_ONE
  if (a->mode & OBJNERF_TRAIN_FP16) return run<_Float16>(net, a, (hipStream_t)stream);
#endif
  return run<__bf16>(net, a, (hipStream_t)stream);
}

}  // namespace obj256
