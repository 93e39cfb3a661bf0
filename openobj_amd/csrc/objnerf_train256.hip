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
#include <cstdlib>
#include <type_traits>
#include "objnerf_device.h"
#include "objnerf_generic.h"

namespace obj256 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

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
// With the feature-distillation loss (gt_feat != NULL; SQ<true>) three sequences join, in consumption order between F6 and B6:
//  F7  hf       W_fl ; x2 slot (h = 0, u = 22) = b_fl      h4 | x2 slots (slot 22 holds 1.0)      8      19
//  B7H d h4 (partial, unmasked)  W_fl[:, :H]^T             d hf                                   8      16
//  B7X d x2     rows = x2 slots: W_fl[:, H:]^T             d hf                                   2      16
enum Seq { F1, F2, F3, F4, F5, F6, B6, B5H, B5X, B4, B3H, B3X, B2, B1, F7, B7H, B7X, NSEQ_ALL };
__host__ __device__ constexpr int seq_nb(int q) {
  return q == F5 ? 9 : q == F6 ? 1 : (q == B5X || q == B7X) ? 2 : (q == B3X || q == B1) ? 3 : 8;
}
__host__ __device__ constexpr int seq_nk(int q) {
  return q == F1 ? 6 : q == F3 ? 22 : (q == F5 || q == F7) ? 19 : q == B6 ? 1 : q == B5H ? 17 : 16;
}
constexpr int MAX_NK = 22;
constexpr int RING_SLOT = MAX_NK * PIECE;            // 22 KB
// Ring stages: one per block of a sequence -- except B6, whose eight one-piece blocks form ONE stage.  A tile's stage
// schedule is a compile-time object: every stage site knows how many pieces the stage two ahead of it has, so the
// transfer set-up is a handful of scalar additions (no table look-ups, no division, no selects).
__host__ __device__ constexpr int stg_nb(int q) { return q == B6 ? 1 : seq_nb(q); }
__host__ __device__ constexpr int stg_np(int q) { return q == B6 ? 8 : seq_nk(q); }      // pieces per stage
template <bool FEAT>
struct SQ {
  static constexpr int NSEQ = FEAT ? 17 : 14;
  // the sequences in CONSUMPTION order (the image is read front to back, the stages follow each other in this order)
  __host__ __device__ static constexpr int at(int i) {
    if (!FEAT) return i;                                 // F1 .. F6, B6 .. B1: the enum's own order
    return i < 6 ? i : i == 6 ? F7 : i == 7 ? B7H : i == 8 ? B7X : i - 3;      // .. F6, F7, B7H, B7X, B6, B5H ..
  }
  __host__ __device__ static constexpr int pos(int q) {
    for (int i = 0; i < NSEQ; ++i) if (at(i) == q) return i;
    return NSEQ;
  }
  __host__ __device__ static constexpr int seq_off(int q) {       // first piece of sequence q (q == NSEQ_ALL: the image's end)
    int o = 0;
    const int n = q == NSEQ_ALL ? NSEQ : pos(q);
    for (int i = 0; i < n; ++i) o += seq_nb(at(i)) * seq_nk(at(i));
    return o;
  }
  __host__ __device__ static constexpr int stg_base(int q) {
    int g = 0;
    const int n = q == NSEQ_ALL ? NSEQ : pos(q);
    for (int i = 0; i < n; ++i) g += stg_nb(at(i));
    return g;
  }
  static constexpr int N_PIECES = seq_off(NSEQ_ALL);
  static constexpr long IMG_BYTES = (long)N_PIECES * PIECE;
  static constexpr int N_STAGES = stg_base(NSEQ_ALL);
  __host__ __device__ static constexpr int stage_pieces(int g) {        // pieces of stage g (mod N_STAGES)
    g %= N_STAGES;
    for (int i = 0; i < NSEQ; ++i) {
      if (g < stg_nb(at(i))) return stg_np(at(i));
      g -= stg_nb(at(i));
    }
    return 0;
  }
  // the sequence of piece `piece` of the image and its first piece
  __host__ __device__ static constexpr int seq_of_piece(int piece, int& first) {
    int o = 0;
    for (int i = 0; i < NSEQ; ++i) {
      const int n = seq_nb(at(i)) * seq_nk(at(i));
      if (piece < o + n) { first = o; return at(i); }
      o += n;
    }
    first = o;
    return NSEQ_ALL;
  }
};
static_assert(SQ<false>::N_PIECES == 1323 && SQ<false>::N_STAGES == 83, "image / stage schedule without the feature loss");
static_assert(SQ<true>::N_PIECES == 1323 + 8 * 19 + 8 * 16 + 2 * 16 && SQ<true>::N_STAGES == 83 + 18, "... with it");
// piece-count classes of a stage; a wave moves a CONTIGUOUS share of a stage: quota pieces (the last wave what is left)
constexpr int NCLS = 6;
__host__ __device__ constexpr int cls_of(int np) { return np == 6 ? 0 : np == 8 ? 1 : np == 16 ? 2 : np == 17 ? 3 : np == 19 ? 4 : 5; }
__host__ __device__ constexpr int cls_np(int c) { return c == 0 ? 6 : c == 1 ? 8 : c == 2 ? 16 : c == 3 ? 17 : c == 4 ? 19 : 22; }
__host__ __device__ constexpr int quota(int np, int nw) { return (np + nw - 1) / nw; }
__host__ __device__ constexpr int wave_cnt(int np, int nw, int w) {
  const int q = quota(np, nw), r = np - w * q;
  return r < 0 ? 0 : (r > q ? q : r);
}

struct Lay256 { int in_w, in_b, m1_w, m1_b, cat_w, cat_b, m2_w, m2_b, a_w, a_b, cl_w, cl_b, oc_w, oc_b, pe_b, fl_w, fl_b; };

constexpr int X2_BIAS_SLOT = 22;        // x2 slot (h = 0, u = 22) is unused by the embedding: with the feature loss it carries the
                                        // constant 1 and the feature layer's image its bias there (no bias table for F7)
template <typename OT, bool FEAT>
__global__ __launch_bounds__(256) void pack256_kernel(int K, const float* __restrict__ params, long p_stride, Lay256 L,
                                                      OT* __restrict__ img) {
  constexpr int N_PIECES = SQ<FEAT>::N_PIECES;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // (object, piece, lane)
  if (idx >= (long)K * N_PIECES * 64) return;
  const int lane = (int)(idx & 63);
  const int piece = (int)((idx >> 6) % N_PIECES);
  const int k = (int)((idx >> 6) / N_PIECES);
  const float* P = params + (long)k * p_stride;
  int first = 0;
  const int q = SQ<FEAT>::seq_of_piece(piece, first);
  const int rel = piece - first, nk = seq_nk(q);
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
      case F7:
        if (ks < KS_H) v = P[L.fl_w + o * (HID + E2) + hid_feat(ks, h, j)];
        else {
          const int u = 8 * (ks - KS_H) + j;
          const int c = x2_slot_col(h, u);
          if (c >= 0) v = P[L.fl_w + o * (HID + E2) + HID + c];
          else if (h == 0 && u == X2_BIAS_SLOT) v = P[L.fl_b + o];
        }
        break;
      case B7H: v = P[L.fl_w + hid_feat(ks, h, j) * (HID + E2) + o]; break;
      case B7X: {
        const int hr = (r >> 2) & 1, n = 4 * (r >> 3) + (r & 3), u = 16 * blk + n;
        const int c = u < 24 ? x2_slot_col(hr, u) : COL_ZERO;
        if (c >= 0) v = P[L.fl_w + hid_feat(ks, h, j) * (HID + E2) + HID + c];
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

// The Gram matrix G = W_of^T W_of of the hoisted feature head as an A-operand image [8 row blocks][16 k-steps][64 lanes][8]
// in the operand type (128 KB per object): kernel A's feature step multiplies it by the rays' fh on the matrix cores.
constexpr int GIMG_PIECES = 8 * KS_H;
template <typename OT>
__global__ __launch_bounds__(256) void packg256_kernel(int K, const float* __restrict__ gram, long gstride, OT* __restrict__ gimg) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // (object, piece, lane)
  if (idx >= (long)K * GIMG_PIECES * 64) return;
  const int lane = (int)(idx & 63);
  const int piece = (int)((idx >> 6) % GIMG_PIECES);
  const int k = (int)((idx >> 6) / GIMG_PIECES);
  const int blk = piece / KS_H, ks = piece - blk * KS_H;
  const int r = lane & 31, h = lane >> 5;
  const float* G = gram + (long)k * gstride + (long)(32 * blk + r) * HID;
  typename Op<OT>::V out;
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = Op<OT>::cvt(G[hid_feat(ks, h, j)]);
  reinterpret_cast<typename Op<OT>::V*>(gimg)[idx] = out;
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
  // feat: two more tensors behind the ten -- 10 = hf (the feature hidden), 11 = d pre-activation of the feature layer
  __host__ __device__ static WsLay make(long n, int tsamp, bool feat = false) {
    WsLay w;
    const long ntile = (n + tsamp - 1) / tsamp;         // whole tiles of kernel A (tsamp samples each)
    w.nsg = ntile * (tsamp / 32);
    w.act_stride = w.nsg * KS_H * PIECE;
    w.off_dpre = 5 * w.act_stride;
    w.off_x1 = (feat ? 12 : 10) * w.act_stride;
    w.off_x2 = w.off_x1 + w.nsg * KS_X1 * PIECE;
    w.off_dhead = w.off_x2 + w.nsg * KS_X2 * PIECE;
    w.obj_bytes = w.off_dhead + w.nsg * PIECE;
    return w;
  }
};

constexpr int PART_FLOATS = 72;         // per (object, workgroup): d B (63), loss terms (3; 4 with the feature term), padding
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
  char* dummy;                          // [NWG_A][8 waves][2 KB]: where a layer's first stage parks its (stale) fragment stores
  float* part;                          // [K][NWG_A][PART_FLOATS]
  Lay256 L;
  WsLay wl;
  // feature-distillation loss (FEAT kernels; DESIGN.md 4.3: the 512-d head hoisted past the compositing)
  float feat_scaling;
  const float* rayin;                   // [K][R][HID + 2]  u = W_of^T g, beta = b_of . g, |g|
  const float* gram;                    // [K][HID HID + HID + 1]  G = W_of^T W_of (symmetric), wb, bb
  const void* gimg;                     // [K][GIMG_PIECES][64][8] operand type: G as an A-operand image (packg256_kernel)
  float* rayfeat;                       // [K][R][HID + 3]  fh, O, a, c   (-> the head's moment GEMMs)
  float *X1, *X2;                       // [K][R][HID + 1]  [a fh | a O], [c fh | c O]
};

// LDS of kernel A (NW waves, NW * 64 threads, tile of NW * 32 samples)
constexpr int L_RING = 0;                                   // 3 x 22 KB weight ring
constexpr int L_HBUF = L_RING + 3 * RING_SLOT;              // per wave 16 KB: the 16 operand fragments of the layer being produced
__host__ __device__ constexpr int l_mask(int NW) { return L_HBUF + NW * KS_H * PIECE; }            // [5 layers][4 block pairs][threads] ReLU bits
__host__ __device__ constexpr int l_bias(int NW) { return l_mask(NW) + 5 * 4 * NW * 64 * 4; }      // [5][8 blk][2 h][16] fp32
__host__ __device__ constexpr int l_strip(int NW) { return l_bias(NW) + 5 * 256 * 4; }             // raw alpha | colour pre [3] per sample
__host__ __device__ constexpr int l_small(int NW) { return l_strip(NW) + 5 * NW * 32 * 4 + 8 * 8 * 4; }   // (+ z per sample, 8 floats per ray) B rows, ba, boc[3] (96 floats)
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

// sin / cos of octaves F0 .. F0 + N - 1 of one direction: the lowest octave from its own exact reduction, the others by
// angle doubling (sin 2x = 2 sin x cos x, cos 2x = 1 - 2 sin^2 x: 3 instructions instead of a reduction + two
// transcendentals).  The error grows ~4x per octave (v_sin: ~1e-6 -> ~1e-4 at the third doubling): used for the
// BACKWARD factors only (cos times a 16-bit-rounded gradient); the forward embedding values are MFMA operands and keep
// their own reductions (measured: the ladder there moved the mid-layer weight gradients from 2e-4 to 1e-3 of the
// specification).
template <int F0, int N>
__device__ __forceinline__ void rev_ladder(const float vh, const float vl, float (&sv)[N], float (&cv)[N]) {
  rev_sincos(vh, vl, (float)(1 << F0), sv[0], cv[0]);
#pragma unroll
  for (int i = 1; i < N; ++i) {
    const float t = sv[i - 1] + sv[i - 1];
    sv[i] = t * cv[i - 1];
    cv[i] = fmaf(-t, sv[i - 1], 1.0f);
  }
}

// ---- epilogues on PACKED words (two 16-bit values per register): the ReLU is a packed integer max with zero (a
// negative value has its sign bit set in either 16-bit format), the branch bit of each half its (inverted) sign bit --
// a pre-activation that rounds to +0 counts as passed: its activation is 0 either way -- and the backward mask a packed
// integer multiply by those 0 / 1 halves.  ~6 instructions per value pair instead of ~13, all native (an inline-asm
// VALU instruction costs an s_nop on either side).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_relu(uint32_t w) {                             // v_pk_max_i16
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}
__device__ __forceinline__ uint32_t pk_nonzero(uint32_t w) {                          // 1 per non-zero half (0x00010001-style)
  uint32_t r;                                                                         // (the vector min is scalarised by hipcc)
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(0x00010001u));
  return r;
}
__device__ __forceinline__ uint32_t pk_mask(uint32_t w, uint32_t m01) {               // halves of w times the 0 / 1 halves of m01
  return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, w) * __builtin_bit_cast(u16x2, m01)));
}
template <typename OT> __device__ __forceinline__ uint32_t pk_cvt(float a, float b);
template <> __device__ __forceinline__ uint32_t pk_cvt<__bf16>(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, b2));      // one v_cvt_pk_bf16_f32
}
template <> __device__ __forceinline__ uint32_t pk_cvt<_Float16>(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x = {a, b};
  const h2 lim = {(_Float16)65504.0f, (_Float16)65504.0f};
  h2 v = __builtin_convertvector(x, h2);                       // (round to nearest even; an overflow becomes +-inf ...)
  v = __builtin_elementwise_max(__builtin_elementwise_min(v, lim), -lim);              // ... and is clamped here
  return __builtin_bit_cast(uint32_t, v);
}

// (a scheduling fence after each k-step's side work pins the epilogue pieces to their MFMA gaps; measured 1 % SLOWER than
// letting the compiler move them: -DOBJ256_SCHED_FENCE_ON restores it)
#ifdef OBJ256_SCHED_FENCE_ON
#define OBJ256_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define OBJ256_SCHED_FENCE() do {} while (0)
#endif
#ifdef OBJ256_TIMING      // diagnostic build: cycles per part of a stage (s_memtime), printed by workgroup 0 / wave 0
#define T256_DECL unsigned long long tm_[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tm_t = __builtin_amdgcn_s_memtime()
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

// objnerf_train256r.hip compiles this file a second time with OBJ256_ROWSPLIT_TU defined: everything above (types, tables,
// pack kernels, workspace layout, arguments, helpers) + the row-split kernel A, WITHOUT -amdgpu-mfma-vgpr-form (its
// accumulators are natural AGPR residents; with that flag this compiler's AGPR rewrite pass crashes on it).
#ifndef OBJ256_ROWSPLIT_TU
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
  // The weight stream: stage g of this workgroup lives in ring slot g % 3.  During stage g the waves issue the LDS-DMA
  // (global_load_lds_dwordx4: 1 KB per wave instruction, no registers; its instruction offset moves BOTH the source and
  // the LDS address -- tools/ubench_glds_off.hip -- so the pieces of a wave's contiguous share need one M0 value and
  // one source pointer) of the pieces of stage g + 2 into the slot that held stage g - 1, which every wave has left
  // (they all passed the barrier that ended it).  A stage ends with a COUNTED wait: everything older than the C
  // vector-memory operations this wave issued during the stage (its transfers and its fragment stores; vmcnt counts
  // loads, stores and LDS-DMA together, in issue order) has landed -- in particular the pieces of the stage computed
  // next -- while the stage's own transfers and stores stay in flight across the barrier: two stages of latency are
  // hidden.  C may under-count (older stragglers are then waited for too), never over-count.  All of it is scalar
  // arithmetic on a compile-time schedule: the image is consumed front to back, so the source pointer just advances.
  // -------------------------------------------------------------------------------------------------------------
  template <int NW>
  struct Stream {
    unsigned long long srcp;      // (scalar) image address of the first piece of the stage TWO ahead of the one computed
    uint32_t dst2;                // (scalar) LDS address of that stage's ring slot
    uint32_t rd;                  // (scalar) LDS address of the slot being computed
    uint32_t lo;                  // (scalar) LDS address of the ring
    uint32_t voff[NCLS];          // (vector) lane * 16 + this wave's share offset inside a stage of class c
    uint32_t woff[NCLS];          // (scalar) this wave's share offset: wave * quota * PIECE
    int cw[NCLS];                 // (scalar) transfers of this wave per stage of class c
    bool last;                    // (scalar) the last wave (the one whose share may be short)

    __device__ __forceinline__ void init(const uint32_t lds_ring, const int wave, const int lane) {
      lo = lds_ring; dst2 = lds_ring; rd = lds_ring;
      last = wave == NW - 1;
#pragma unroll
      for (int c = 0; c < NCLS; ++c) {
        const int q = quota(cls_np(c), NW);
        int r = cls_np(c) - wave * q;
        r = r < 0 ? 0 : (r > q ? q : r);
        cw[c] = __builtin_amdgcn_readfirstlane(r);
        woff[c] = __builtin_amdgcn_readfirstlane(wave * q * PIECE);
        voff[c] = (uint32_t)lane * 16u + woff[c];
      }
    }
    __device__ __forceinline__ void set_source(const char* p) {
      const unsigned long long sa = (unsigned long long)p;
      // scalar registers, provably (readfirstlane returns int: through uint32_t, or the low half is sign-extended)
      srcp = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa >> 32)) << 32) |
             (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa);
    }
    // transfer I of this wave's share of a stage of NP pieces (first piece at srcp, slot at dst2); ALLC: every transfer
    // is conditional on the run-time count `cnt` (the stages that wrap to the next tile)
    template <int NP, int I, bool ALLC>
    __device__ __forceinline__ void issue(const int cnt) const {
      constexpr int Q4 = quota(NP, NW), CMIN = wave_cnt(NP, NW, NW - 1), C = cls_of(NP);
      if constexpr (I < Q4) {
        auto go = [&]() __attribute__((always_inline)) {
          // M0 = the LDS destination; operands are scalar-ALU results of values made scalar long before: no
          // VALU-written SGPR reaches the VMEM instruction inside its wait states (hipcc pads nothing for inline asm)
          const uint32_t d = dst2 + woff[C] + (I >= 4 ? 4 * PIECE : 0);
          const unsigned long long sa = srcp + (I >= 4 ? 4 * PIECE : 0);
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                       :: "s"(d), "v"(voff[C]), "s"(sa), "n"((I & 3) * PIECE) : "memory", "m0");
        };
        if constexpr (!ALLC && I < CMIN) go();
        else if (I < cnt) go();
      }
    }
    // the stage two ahead has been requested: on to the next one
    template <int NP>
    __device__ __forceinline__ void advance() {
      srcp += (unsigned long long)NP * PIECE;
      dst2 += RING_SLOT; if (dst2 == lo + 3 * RING_SLOT) dst2 = lo;
      rd += RING_SLOT; if (rd == lo + 3 * RING_SLOT) rd = lo;
    }
  };
  // end of a stage (raw s_barrier: __syncthreads() would drain the DMA): at most N vector-memory operations of this
  // wave stay in flight
  template <int N>
  static __device__ __forceinline__ void sync_imm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N < 63 ? N : 63) : "memory");
  }
  // ... of a stage that requested a stage of NP pieces and issued ST stores of its own
  template <int NW, int NP, int ST>
  static __device__ __forceinline__ void stage_sync(const bool last_wave) {
    constexpr int Q4 = quota(NP, NW), CMIN = wave_cnt(NP, NW, NW - 1);
    static_assert(NW < 3 || wave_cnt(NP, NW, NW - 2) == Q4, "only the last wave's share is short");
    if constexpr (Q4 == CMIN) sync_imm<Q4 + ST>();
    else if (last_wave) sync_imm<CMIN + ST>();
    else sync_imm<Q4 + ST>();
  }

  // one block of a sequence: acc += sum over the stage's k-steps of piece(ks) x bfrag(ks).  The A operands are read
  // DEPTH k-steps ahead by hand (program order of `rd`; round 4: plain LDS loads whose LGKM waits the compiler counts --
  // rounds 3's inline ds_read_b128 + hand-counted waits remain under -DOBJ256_ASM_LDS_READS).  side(ks) is called after the MFMA of k-step ks has been issued: work that does not depend on the
  // accumulator (the DMA of a later stage, the previous block's fragment stores) goes into the MFMA's shadow there --
  // with one wave per SIMD nothing else would overlap it.
  static constexpr int DEPTH = 6;
#ifdef OBJ256_ASM_LDS_READS   // rounds 3's form: bare inline asm reads + hand-counted waits tied to the destination registers
  template <int OFF>
  static __device__ __forceinline__ void rd(V& dst, const uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
  }
  template <int N>
  static __device__ __forceinline__ void wait_for(V& x) {       // at most N LDS reads still outstanding
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
  }
  template <int N>
  static __device__ __forceinline__ void wait_for2(V& x, V& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
  }
#else
  // compiler-visible LDS loads: the compiler counts the outstanding LGKM operations itself and waits where a value is
  // first used (the same counted s_waitcnt, placed by the tool that also places the copies and spills)
  template <int OFF>
  static __device__ __forceinline__ void rd(V& dst, const uint32_t addr) {
    dst = *reinterpret_cast<const __attribute__((address_space(3))) V*>((uintptr_t)(addr + OFF));
  }
  template <int N>
  static __device__ __forceinline__ void wait_for(V&) {}
  template <int N>
  static __device__ __forceinline__ void wait_for2(V&, V&) {}
#endif
  // TWO accumulator chains (even / odd k-steps): a v_mfma_f32_32x32x16 that depends on the one issued right before it
  // starts ~110 cycles after it, an independent one after ~34 (tools/ubench_mfma32.hip: 645 against 1708 TFLOP/s on
  // the whole chip) -- with one wave per SIMD there is no other wave to fill that gap, so the block's contraction is
  // split over two accumulators that the epilogue adds.  One counted wait serves a PAIR of k-steps (every instruction
  // of the single wave costs an issue slot).
  // Z0 / Z1: the chain starts from zero (the MFMA takes the constant as its C operand: no register initialisation).
  template <int NK, int KS, bool Z0, bool Z1, class BF, class SD>
  static __device__ __forceinline__ void step(f32x16& acc0, f32x16& acc1, V (&a)[DEPTH], const uint32_t addr, const BF& bfrag, SD& side) {
    if constexpr (KS < NK) {
      constexpr int inflight = (NK - 1 - KS) < (DEPTH - 1) ? (NK - 1 - KS) : (DEPTH - 1);   // reads issued after read KS
      if constexpr ((KS & 1) == 0) {
        if constexpr (KS + 1 < NK) wait_for2<(inflight > 0 ? inflight - 1 : 0)>(a[KS % DEPTH], a[(KS + 1) % DEPTH]);
        else wait_for<inflight>(a[KS % DEPTH]);
        if constexpr (KS == 0 && Z0) acc0 = O::mfma(a[KS % DEPTH], bfrag(KS), zero16());
        else acc0 = O::mfma(a[KS % DEPTH], bfrag(KS), acc0);
      } else {
        if constexpr (KS == 1 && Z1) acc1 = O::mfma(a[KS % DEPTH], bfrag(KS), zero16());
        else acc1 = O::mfma(a[KS % DEPTH], bfrag(KS), acc1);
      }
      if constexpr (KS + DEPTH < NK) rd<(KS + DEPTH) * PIECE>(a[KS % DEPTH], addr);
      side(std::integral_constant<int, KS>{});
      step<NK, KS + 1, Z0, Z1>(acc0, acc1, a, addr, bfrag, side);
    }
  }
  template <int NK, bool Z0, bool Z1, class BF, class SD>
  static __device__ __forceinline__ void block_mma(f32x16& acc0, f32x16& acc1, const uint32_t addr, const BF& bfrag, SD& side) {
    static_assert(NK >= 2 || !Z1, "chain 1 needs a k-step to start from");
    V a[DEPTH];
    rd<0>(a[0], addr);
    if constexpr (NK > 1) rd<PIECE>(a[1], addr);
    if constexpr (NK > 2) rd<2 * PIECE>(a[2], addr);
    if constexpr (NK > 3) rd<3 * PIECE>(a[3], addr);
    if constexpr (NK > 4) rd<4 * PIECE>(a[4], addr);
    if constexpr (NK > 5) rd<5 * PIECE>(a[5], addr);
    step<NK, 0, Z0, Z1>(acc0, acc1, a, addr, bfrag, side);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// kernel A
// ---------------------------------------------------------------------------------------------------------------
template <typename OT> struct Unpack;
template <> struct Unpack<__bf16> {          // the two 16-bit halves of a packed word as floats
  static __device__ __forceinline__ void get(uint32_t w, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, w << 16); hi = __builtin_bit_cast(float, w & 0xffff0000u);
  }
};
template <> struct Unpack<_Float16> {
  static __device__ __forceinline__ void get(uint32_t w, float& lo, float& hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, w);
    lo = (float)v[0]; hi = (float)v[1];
  }
};

template <typename OT, int S, int NW, bool FEAT>
__global__ __launch_bounds__(NW * 64) void fwd256_kernel(const FwdArgs a) {
  typedef SQ<FEAT> SQT;
  constexpr int N_STAGES = SQT::N_STAGES;
  constexpr long IMG_BYTES = SQT::IMG_BYTES;
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
  float* s_z = s_raw + 4 * TSAMP;              // depth of every sample of the tile, staged early (compositing reads it)
  float* s_ray = s_z + TSAMP;                  // per ray of the tile: gt depth, gt rgb, label (8 floats)
  float* s_small = reinterpret_cast<float*>(lds + L_SMALL);
  float* s_db = reinterpret_cast<float*>(lds + L_DB);
  char* hbuf = lds + L_HBUF + w * (KS_H * PIECE) + lane * 16;       // this lane's 16 bytes of fragment ks at + ks * PIECE
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;

  typename KT::template Stream<NW> strm;
  strm.init(__builtin_amdgcn_readfirstlane(lds0 + L_RING), w, lane);
  long tile_i; int k_i;
  {
    const long obj0 = tau0 / a.ntile;
    k_i = (int)obj0; tile_i = tau0 - obj0 * a.ntile;
    strm.set_source((const char*)a.img + obj0 * IMG_BYTES);
    // prologue: stages 0 and 1 (F1 blocks 0 and 1) into slots 0 and 1; the stream then points at stage 2 / slot 2
    constexpr int NP0 = SQT::stage_pieces(0), NP1 = SQT::stage_pieces(1);
    strm.template issue<NP0, 0, true>(strm.cw[cls_of(NP0)]); strm.template issue<NP0, 1, true>(strm.cw[cls_of(NP0)]);
    strm.srcp += (unsigned long long)NP0 * PIECE; strm.dst2 += RING_SLOT;
    strm.template issue<NP1, 0, true>(strm.cw[cls_of(NP1)]); strm.template issue<NP1, 1, true>(strm.cw[cls_of(NP1)]);
    strm.srcp += (unsigned long long)NP1 * PIECE; strm.dst2 += RING_SLOT;
    static_assert(quota(NP0, NW) <= 2 && quota(NP1, NW) <= 2, "prologue transfers");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  typedef __attribute__((address_space(1))) typename Op<OT>::V GVp;
  GVp* const park = (GVp*)(a.dummy + ((long)wg * 8 + w) * 2048 + lane * 16);

  const float gs = a.grad_scale, inv_gs = 1.0f / a.grad_scale;
  T256_DECL;
  int cur_obj = -1;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f, l_f = 0.f;
  // d B of this lane's directions, summed over this lane's samples of the object.  With the feature loss there is no room
  // for 33 registers that live through every layer loop (the compiler spilled them and more, and its AGPR rewrite pass
  // crashes on some spilled forms): there each tile's contribution is reduced over the lanes at once and added to the
  // wave's sums in LDS (db_add below)
  constexpr int NDB = FEAT ? 1 : 11;
  float dbacc[NDB][3];
#pragma unroll
  for (int dd = 0; dd < NDB; ++dd) dbacc[dd][0] = dbacc[dd][1] = dbacc[dd][2] = 0.f;
  float scale = 1.0f, inv1 = 0.f, inv2 = 0.f, ba = 0.f, boc0 = 0.f, boc1 = 0.f, boc2 = 0.f;

  auto flush_object = [&]() {       // partial d B and loss terms of (cur_obj, this workgroup)
    __syncthreads();
    float* pw = s_db + w * 68;
#pragma unroll
    for (int dd = 0; dd < NDB; ++dd) {       // each 32-lane half owns its directions: sum over its lanes (samples)
      if constexpr (FEAT) break;             // (already in pw: db_add)
      float g0 = dbacc[dd][0], g1 = dbacc[dd][1], g2 = dbacc[dd][2];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) { g0 += __shfl_xor(g0, d, 32); g1 += __shfl_xor(g1, d, 32); g2 += __shfl_xor(g2, d, 32); }
      if (s == 0 && (dd < 10 || h == 0)) {
        const int jd = 11 * h + dd;
        pw[3 * jd] = g0; pw[3 * jd + 1] = g1; pw[3 * jd + 2] = g2;
      }
      dbacc[dd][0] = dbacc[dd][1] = dbacc[dd][2] = 0.f;
    }
    l_d = seg_sum<64>(l_d); l_c = seg_sum<64>(l_c); l_o = seg_sum<64>(l_o); l_f = seg_sum<64>(l_f);
    if (lane == 0) { pw[64] = l_d; pw[65] = l_c; pw[66] = l_o; pw[67] = l_f; }
    __syncthreads();
    if (tid < 68) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) v += s_db[ww * 68 + tid];
      a.part[((long)cur_obj * NWG_A + wg) * PART_FLOATS + tid] = v;
    }
    __syncthreads();
  };

  // the sample point (and depth) of this lane in tile (k_, tile_): requested one tile ahead
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

  const int tid_k = tid;
  char* const hbuf_k = hbuf;
  for (long tau = tau0; tau < tau1; ++tau) {
    // With the feature loss the lane-derived constants are re-derived per tile from an opaque copy of the thread index:
    // left loop-invariant, the compiler hoists some 70 of them (LDS addresses of every fragment slot, vector bases,
    // lane predicates) ahead of the tile loop and then spills them, and this compiler's AGPR rewrite pass crashes on
    // the spilled form.  (Without the feature loss the kernel fits as it is, and its code is left alone.)
    int tid = tid_k;
    if constexpr (FEAT) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, s = lane & 31, h = lane >> 5;
    char* const hbuf = FEAT ? lds + L_HBUF + w * (KS_H * PIECE) + lane * 16 : hbuf_k;
    const int k = k_i;                        // (the stream is at stage 0 of this tile)
    const long tile = tile_i;
    const bool live = tau + 1 < tau1;         // another tile follows: the last two stages request its first two
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
      l_d = l_c = l_o = l_f = 0.f;
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
    // (the lane offset is made opaque at every use: left visible, the compiler hoists one 64-bit vector base per tensor
    // out of the tile loop and then spills them -- with it hidden the base stays scalar and the lane offset rides in the
    // access's own offset operand)
    auto act_base = [&](int tensor, int ks) __attribute__((always_inline)) -> GV* {
      uint32_t lo = lane_off;
      asm volatile("" : "+v"(lo));
      return (GV*)(ws_obj + (long)tensor * a.wl.act_stride + (sg * KS_H + ks) * (64 * 16) + lo);
    };

    // ------------------------------------------------------------------ sample point of this lane (vmap.py:548-551)
    const int st_idx = 32 * w + s;                               // sample index inside the tile
    const int q = st_idx / S, si = st_idx - q * S;
    const long ray = tile * TR + q;
    const bool valid = ray < a.R;
    const float px = npx, py = npy, pz = npz, z_own = nzv;      // (requested during the previous tile)
    // per-ray targets of the tile: one lane per ray requests them now and parks them in LDS after the first layer
    float rg0 = 0.f, rg1 = 0.f, rg2 = 0.f, rg3 = 0.f; int rlab = 2;
    if (valid && si == 0) {
      const long rr = (long)k * a.R + ray;
      rg0 = a.gt_depth[rr]; rg1 = a.gt_rgb[rr * 3]; rg2 = a.gt_rgb[rr * 3 + 1]; rg3 = a.gt_rgb[rr * 3 + 2]; rlab = a.labels[rr];
    }
    const float t0 = px / scale, t1 = py / scale, t2 = pz / scale;       // embedding.py:47
    // d B[j][c] += d proj_j * t_c of this lane's sample (dpv: d loss / d projection of this half's directions, unscaled).
    // FEAT: the 33 products are reduced over the 32 lanes of the half together (33 independent shuffle chains) and lane 0
    // adds them to the wave's sums
    auto db_add = [&](const float (&dpv)[11]) __attribute__((always_inline)) {
      if constexpr (!FEAT) {
#pragma unroll
        for (int dd = 0; dd < 11; ++dd) {
          dbacc[dd][0] = fmaf(dpv[dd], t0, dbacc[dd][0]);
          dbacc[dd][1] = fmaf(dpv[dd], t1, dbacc[dd][1]);
          dbacc[dd][2] = fmaf(dpv[dd], t2, dbacc[dd][2]);
        }
      } else {
        float g[33];
#pragma unroll
        for (int dd = 0; dd < 11; ++dd) { g[3 * dd] = dpv[dd] * t0; g[3 * dd + 1] = dpv[dd] * t1; g[3 * dd + 2] = dpv[dd] * t2; }
#pragma unroll
        for (int i = 0; i < 33; ++i) g[i] = wave_sum32(g[i]);      // DPP row sums + one row swap: no LDS permutes (165 of them cost ~6 k cycles)
        if (s == 0) {                       // (all reads, then all writes: 33 dependent read-add-write round trips otherwise)
          float* pj = s_db + w * 68 + 33 * h;
          float o[33];
#pragma unroll
          for (int i = 0; i < 33; ++i) o[i] = pj[i];
#pragma unroll
          for (int i = 0; i < 33; ++i)
            if (i < 30 || h == 0) pj[i] = o[i] + g[i];
        }
      }
    };
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
      for (int t = 0; t < KS_X1; ++t) {        // (every octave from its own exact reduction: these values ARE operands --
#pragma unroll                               //  a doubling ladder's 1e-4 moves 2 % of their bf16 roundings)
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
    // the transfers of a stage that requests a stage of NP2 pieces, spread behind the MFMAs of its NK k-steps (transfer
    // i after MFMA 2 i + 1: one wave issues one instruction per ~4 cycles, so no gap between two MFMAs should hold more
    // than the MFMA's own cycles), whatever has no such slot after the last one
    auto dma_side = [&](auto np2_tag, auto nk_tag, auto ks_tag, auto allc_tag, const int cnt) __attribute__((always_inline)) {
      constexpr int NP2 = decltype(np2_tag)::value, NK = decltype(nk_tag)::value, KS = decltype(ks_tag)::value;
      constexpr bool AC = decltype(allc_tag)::value;
      constexpr int CD = quota(NP2, NW);
      static_assert(CD <= 6, "transfers per wave and stage");
      if constexpr (KS == NK - 1) {
        if constexpr (1 >= NK - 1 && 0 < CD) strm.template issue<NP2, 0, AC>(cnt);
        if constexpr (3 >= NK - 1 && 1 < CD) strm.template issue<NP2, 1, AC>(cnt);
        if constexpr (5 >= NK - 1 && 2 < CD) strm.template issue<NP2, 2, AC>(cnt);
        if constexpr (7 >= NK - 1 && 3 < CD) strm.template issue<NP2, 3, AC>(cnt);
        if constexpr (9 >= NK - 1 && 4 < CD) strm.template issue<NP2, 4, AC>(cnt);
        if constexpr (11 >= NK - 1 && 5 < CD) strm.template issue<NP2, 5, AC>(cnt);
      } else if constexpr ((KS & 1) && KS / 2 < CD) strm.template issue<NP2, KS / 2, AC>(cnt);
    };
    auto reload = [&]() __attribute__((always_inline)) {          // hin <- the fragments the layer just produced
#pragma unroll
      for (int ks = 0; ks < KS_H; ++ks) hin[ks] = *reinterpret_cast<const V*>(hbuf + ks * PIECE);
    };
    auto ring_addr = [&]() __attribute__((always_inline)) -> uint32_t { return strm.rd + (uint32_t)lane * 16u; };
    // a stage without an epilogue of its own: the block's MFMAs (two chains) + the transfers; the caller ends it
    auto plain_stage = [&](auto np2_tag, auto nk_tag, auto allc_tag, auto keep_tag, const int cnt, f32x16& acc0, f32x16& acc1, auto&& bsel)
        __attribute__((always_inline)) {
      constexpr int NK = decltype(nk_tag)::value;
      auto sd = [&](auto ks_tag) __attribute__((always_inline)) { dma_side(np2_tag, nk_tag, ks_tag, allc_tag, cnt); };
      KT::template block_mma<NK, !decltype(keep_tag)::value, true>(acc0, acc1, ring_addr(), bsel, sd);
    };
#define NP2_OF(Q_, BLK_) SQT::stage_pieces(SQT::stg_base(Q_) + (BLK_) + 2)

    // Hidden layers are SOFTWARE-PIPELINED over their eight blocks on two accumulators: with one wave per SIMD nothing
    // overlaps an MFMA but this wave's own independent instructions, so the epilogue of block b - 1 (convert, ReLU /
    // mask, pack, hand-off and fragment stores) runs piecewise in the shadow of block b's MFMAs (one value pair behind
    // each of the first eight), and so do the scalar set-up of the ring transfer and the bias rows of block b + 1.  Only
    // the last block's epilogue is exposed.  Stage 0 runs the same code on a stale accumulator: its stores are parked
    // in hand-off pieces 14 / 15 and mask word 3, which blocks 6 / 7 overwrite later, and its global stores are skipped.
    // forward hidden layer: NK k-steps from bsel(ks); blocks 0..7 -> fragments (LDS hand-off buffer + workspace tensor
    // `layer`), ReLU bits.  The block loop is rolled over block PAIRS (the accumulators swap roles) for blocks 0..5,
    // whose stage two ahead lies in the same sequence; blocks 6 and 7 are peeled (theirs is the next sequence's).
    auto fwd_layer = [&](auto seq_tag, auto nk_tag, const int layer, auto&& bsel, auto plain_tag, const int tensor)
        __attribute__((always_inline)) {
      // plain_tag: the feature layer -- its bias rides in the image (x2 slot 22), and its ReLU bits are not kept (d hf is
      // formed from the values themselves): no bias table row, no mask words
      constexpr bool PLAIN = decltype(plain_tag)::value;
      constexpr int NK = decltype(nk_tag)::value, Q = decltype(seq_tag)::value;
      constexpr int SK = NK - 1 < 8 ? NK - 1 : 8;          // the k-step whose shadow finishes the previous block's epilogue
      uint32_t mprev = 0;
      auto load_bias = [&](f32x16& acc, const int blk) __attribute__((always_inline)) {
        if constexpr (PLAIN) { acc = zero16(); return; }
        const float* bp = s_bias + layer * 256 + blk * 32 + h * 16;
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
          const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bp + 4 * n4);
          acc[4 * n4] = b4[0]; acc[4 * n4 + 1] = b4[1]; acc[4 * n4 + 2] = b4[2]; acc[4 * n4 + 3] = b4[3];
        }
      };
      auto finish = [&](const int pb, const uint32_t (&wd)[8], const uint32_t bits, GV* dst) __attribute__((always_inline)) {
        const uint4 u0 = make_uint4(wd[0], wd[1], wd[2], wd[3]), u1 = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
        *reinterpret_cast<V*>(hbuf + (2 * pb) * PIECE) = f0;
        *reinterpret_cast<V*>(hbuf + (2 * pb + 1) * PIECE) = f1;
        if constexpr (!PLAIN) s_mask[(layer * 4 + (pb >> 1)) * NTHR + tid] = (pb & 1) ? (mprev | (bits << 8)) : bits;
        mprev = bits;
        // non-temporal: 5 KB per sample stream through the L2 that also has to keep serving the weight ring
        __builtin_nontemporal_store(f0, dst);
        __builtin_nontemporal_store(f1, dst + 64);
      };
      auto stage = [&](auto np2_tag, const int blk, f32x16& cur, f32x16& cur1, f32x16& prv, f32x16& prv1) __attribute__((always_inline)) {
        constexpr int NP2 = decltype(np2_tag)::value;
        uint32_t wd[8], bits = 0;
        const int pb = (blk + 7) & 7;                      // the block whose epilogue rides along
        auto piece = [&](const int i) __attribute__((always_inline)) {
          const uint32_t c = pk_cvt<OT>(prv[2 * i] + prv1[2 * i], prv[2 * i + 1] + prv1[2 * i + 1]);
          wd[i] = pk_relu(c);
          bits = ((c >> (15 - i)) & (0x00010001u << i)) | bits;      // SIGN bits: bit i value 2 i, bit 16 + i value 2 i + 1
        };
        auto sd = [&](auto ks_tag) __attribute__((always_inline)) {
          constexpr int KS = decltype(ks_tag)::value;
          if constexpr (KS < SK) piece(KS);
          if constexpr (KS == SK) {
#pragma unroll
            for (int i = SK; i < 8; ++i) piece(i);
            finish(pb, wd, ~bits & 0x00ff00ffu, blk > 0 ? act_base(tensor, 2 * pb) : (GV*)park);
            load_bias(prv, (blk + 1) & 7);                 // prv becomes the next block's first chain (the second starts from 0)
          }
          dma_side(np2_tag, nk_tag, ks_tag, std::false_type{}, strm.cw[cls_of(NP2)]);
          OBJ256_SCHED_FENCE();
        };
        T256(0);
        KT::template block_mma<NK, false, true>(cur, cur1, ring_addr(), bsel, sd);
        asm volatile("" : "+v"(prv));                      // (the compiler's wait for the bias rows lands here, not in the next stage)
        T256(1);
        KT::template stage_sync<NW, NP2, 2>(strm.last);
        T256(2);
        strm.template advance<NP2>();
      };
      f32x16 accA, accA1 = zero16(), accB = zero16(), accB1 = zero16();
      load_bias(accA, 0);
#pragma unroll 1
      for (int blk = 0; blk < 6; blk += 2) {
        stage(nk_tag, blk, accA, accA1, accB, accB1);
        stage(nk_tag, blk + 1, accB, accB1, accA, accA1);
      }
      stage(std::integral_constant<int, NP2_OF(Q, 6)>{}, 6, accA, accA1, accB, accB1);
      stage(std::integral_constant<int, NP2_OF(Q, 7)>{}, 7, accB, accB1, accA, accA1);
      {   // block 7's epilogue
        uint32_t wd[8], bits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t c = pk_cvt<OT>(accB[2 * i] + accB1[2 * i], accB[2 * i + 1] + accB1[2 * i + 1]);
          wd[i] = pk_relu(c);
          bits = ((c >> (15 - i)) & (0x00010001u << i)) | bits;
        }
        finish(7, wd, ~bits & 0x00ff00ffu, act_base(tensor, 14));
      }
    };
    // ------------------------------------------------------------------ forward
    T256(6);
    fwd_layer(std::integral_constant<int, F1>{}, std::integral_constant<int, 6>{}, 0, [&](int ks) -> V { return x1f[ks]; }, std::false_type{}, 0);   // h1
    {   // the loads requested at the top of the tile have landed under the first layer: park them for the compositing
      s_z[st_idx] = z_own;
      if (si == 0) { float* rp = s_ray + 8 * q; rp[0] = rg0; rp[1] = rg1; rp[2] = rg2; rp[3] = rg3; rp[4] = __int_as_float(rlab); }
      if (live) load_point((int)next_obj, next_obj != k ? 0 : tile + 1, npx, npy, npz, nzv);      // the next tile's sample
    }
    reload();
    fwd_layer(std::integral_constant<int, F2>{}, std::integral_constant<int, 16>{}, 1, [&](int ks) -> V { return hin[ks]; }, std::false_type{}, 1);  // h2
    reload();
    fwd_layer(std::integral_constant<int, F3>{}, std::integral_constant<int, 22>{}, 2,
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : x1f[ks - KS_H]; }, std::false_type{}, 2);                        // h3
    reload();
    fwd_layer(std::integral_constant<int, F4>{}, std::integral_constant<int, 16>{}, 3, [&](int ks) -> V { return hin[ks]; }, std::false_type{}, 3);  // h4
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
          if (FEAT && u == X2_BIAS_SLOT && h == 0) v = 1.0f;       // the feature layer's bias column (pack256_kernel, F7)
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
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : x2f[ks - KS_H]; }, std::false_type{}, 4);                        // hc
    T256(7);
    {   // F5 block 8: row 0 = w_alpha . h4  (raw density, model.py:81)
      constexpr int NP2 = NP2_OF(F5, 8);
      f32x16 acc = zero16(), acc1 = zero16();
      plain_stage(std::integral_constant<int, NP2>{}, std::integral_constant<int, 19>{}, std::false_type{}, std::false_type{}, strm.cw[cls_of(NP2)], acc, acc1,
                  [&](int ks) -> V { return ks < KS_H ? hin[ks] : x2f[ks - KS_H]; });
      if (h == 0) s_raw[st_idx] = (acc[0] + acc1[0]) + ba;
      KT::template stage_sync<NW, NP2, 0>(strm.last);
      strm.template advance<NP2>();
    }
    // (with the feature loss h4 stays in `hin` for the feature layer: the colour head then reads hc's fragments straight
    // from the hand-off buffer, one LDS read per k-step)
    if constexpr (!FEAT) reload();             // hin = hc
    {   // F6: colour head on hc (model.py:95)
      constexpr int NP2 = NP2_OF(F6, 0);
      f32x16 acc = zero16(), acc1 = zero16();
      plain_stage(std::integral_constant<int, NP2>{}, std::integral_constant<int, 16>{}, std::false_type{}, std::false_type{}, strm.cw[cls_of(NP2)], acc, acc1,
                  [&](int ks) -> V { if constexpr (FEAT) return *reinterpret_cast<const V*>(hbuf + ks * PIECE); else return hin[ks]; });
      if (h == 0) {
        s_col[st_idx] = (acc[0] + acc1[0]) + boc0; s_col[TSAMP + st_idx] = (acc[1] + acc1[1]) + boc1;
        s_col[2 * TSAMP + st_idx] = (acc[2] + acc1[2]) + boc2;
      }
      KT::template stage_sync<NW, NP2, 0>(strm.last);           // (also publishes the strips to the compositing waves)
      strm.template advance<NP2>();
    }
    T256(8);
    // ------------------------------------------------------------------ feature layer (FEAT): hf = relu([h4 | x2] W_fl^T + b_fl)
    // scratch of the feature step: the ring slot of the stage just computed is free until the NEXT stage starts its
    // transfers (its barrier has passed; the two other slots hold / receive the two stages ahead)
    float* fx = nullptr;
    if constexpr (FEAT) {
      static_assert(!FEAT || NTHR == HID, "the feature step maps a thread to a hidden feature");
      fwd_layer(std::integral_constant<int, F7>{}, std::integral_constant<int, 19>{}, 0,
                [&](int ks) -> V { return ks < KS_H ? hin[ks] : x2f[ks - KS_H]; }, std::true_type{}, 10);                        // hf
      T256(15);
      const uint32_t free_slot = strm.rd == strm.lo ? strm.lo + 2 * RING_SLOT : strm.rd - RING_SLOT;
      fx = reinterpret_cast<float*>(lds + (free_slot - lds0));
    }
    constexpr int FX_X = 0, FX_FH = FX_X + NW * HID, FX_DFH = FX_FH + 4 * HID, FX_W = FX_DFH + 4 * HID, FX_DWV = FX_W + TSAMP,
                  FX_RQ = FX_DWV + TSAMP, FX_RED = FX_RQ + 4 * 8, FX_END = FX_RED + NW * 4;
    static_assert(FX_END * 4 <= RING_SLOT, "feature scratch");
    // ------------------------------------------------------------------ compositing + losses (loss.py:27-101)
    // pass 0: the whole pass (no feature loss).  With it the pass runs twice: 1 = ray weights, loss terms, opacity (nothing
    // of which depends on the feature term), 2 = the gradients, with the feature term's d loss / d weight added.
    auto composite = [&](auto pass_tag) __attribute__((always_inline)) {
      constexpr int PASS = decltype(pass_tag)::value;
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
        if (on) { const float* rp = s_ray + 8 * qq; gtd = rp[0]; gr = rp[1]; gg = rp[2]; gb = rp[3]; lab = __float_as_int(rp[4]); }
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
        if (PASS != 2 && on && li == 0) {
          l_d += m1 * fabsf(rd) * info * inv1;
          l_c += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
          l_o += m2 * fabsf(ro) * inv2;
        }
        if constexpr (PASS == 1) {                                      // ray weights / opacity for the feature step
#pragma unroll
          for (int e = 0; e < SPL; ++e)
            if (qq < TR) fx[FX_W + qq * S + li * SPL + e] = on ? wgt[e] : 0.0f;
          if (qq < TR && li == 0) { fx[FX_RQ + 8 * qq] = on ? Oo : 0.0f; fx[FX_RQ + 8 * qq + 1] = on ? m1 : 0.0f; }
          continue;
        }
        float dw[SPL], qv[SPL], ql_sum = 0.f;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
          dw[e] = gD * zz[e] + gO + gC0 * c0[e] + gC1 * c1[e] + gC2 * c2[e];
          if constexpr (PASS == 2) dw[e] += on ? fx[FX_DWV + qq * S + li * SPL + e] : 0.0f;      // the feature term's part
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
    };
    if constexpr (!FEAT) {
      composite(std::integral_constant<int, 0>{});
    } else {
      // ================================================================ feature-distillation term (loss.py:81-99 in the
      // hoisted form of DESIGN.md 4.3: F = W_of fh + b_of O with fh = sum_s w_s hf_s per ray; the cosine needs fh . u,
      // fh^T G fh, fh . wb only)
      for (int i = tid; i < 4 * HID; i += NTHR) fx[FX_DFH + i] = 0.0f;
      composite(std::integral_constant<int, 1>{});
      reload();                                                           // hin = hf (this wave's 32 samples)
      __syncthreads();
      T256(18);
      const int qw = (32 * w) / S;                                        // the tile's ray this wave's samples belong to
      const float wsm = valid ? fx[FX_W + st_idx] : 0.0f;
      // G's fragments for the product G fh below (wave w: row blocks 2 w, 2 w + 1) are requested NOW, from the image in
      // global memory (L2-resident: every workgroup of the object reads the same 128 KB), 16 bytes per lane and fragment:
      // their latency passes under the reduce-scatter
      V GA0[KS_H], GA1[KS_H];
      {
        const GV* gp = (const GV*)((const char*)a.gimg + ((long)k * GIMG_PIECES + 2 * w * KS_H) * PIECE) + lane;
#pragma unroll
        for (int i = 0; i < KS_H; ++i) { GA0[i] = gp[i * 64]; GA1[i] = gp[(KS_H + i) * 64]; }
      }
      {
        // fh partial of this wave: sum over its 32 samples (the lanes of a half) of w_s hf -- a reduce-scatter: after the
        // five exchange steps lane s holds the complete sums of entries 4 s .. 4 s + 3 (entry 8 ks + j <-> feature
        // hid_feat(ks, h, j)), i.e. four consecutive features
        // (in two halves of 64 entries -- k-steps 0..7 and 8..15: 128 live sums next to the 64 fragment registers spilled)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float v[64];
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * ks + j] = wsm * (float)hin[8 * half + ks][j];
#pragma unroll
          for (int nn = 32, d = 16; nn >= 2; nn >>= 1, d >>= 1) {
            const bool up = (s & d) != 0;
#pragma unroll
            for (int i = 0; i < nn; ++i) {
              const float keep = up ? v[i + nn] : v[i], send = up ? v[i] : v[i + nn];
              v[i] = keep + __shfl_xor(send, d, 32);
            }
          }
          // lane s holds entries 2 s, 2 s + 1 of the half: k-step 8 half + (s >> 2), elements j = 2 (s & 3) + {0, 1}
          const int ks = 8 * half + (s >> 2), j = 2 * (s & 3);
          const int f0 = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);      // hid_feat(ks, h, j)
          fx[FX_X + w * HID + f0] = v[0];
          fx[FX_X + w * HID + f0 + 1] = v[1];
        }
      }
      __syncthreads();
      T256(16);
      const float* Gk = a.gram + (long)k * ((long)HID * HID + HID + 1);
      const float* wbv = Gk + (long)HID * HID;
      // fh of every ray of the tile (thread = feature)
#pragma unroll
      for (int q2 = 0; q2 < TR; ++q2) {
        float fhv = 0.0f;
#pragma unroll
        for (int ww = 0; ww < S / 32; ++ww) fhv += fx[FX_X + (q2 * (S / 32) + ww) * HID + tid];
        fx[FX_FH + q2 * HID + tid] = fhv;
      }
      __syncthreads();
      {
        // G fh on the matrix cores: wave w owns the rows 64 w .. 64 w + 63 (two 32-row blocks) of G, column c of the B
        // operand is ray c of the tile (c < TR; the other columns are zero) -- the thread-per-row dot product this replaces
        // read 256 dependent-latency words per thread and ray and took more than the rest of the tile together.  (G
        // enters in the operand type, like W_of and the targets entered its GEMM.)
        f32x16 g0 = zero16(), g1 = zero16();
        const int c = s;
        const float* fb = fx + FX_FH + (c < TR ? c : 0) * HID + 4 * h;
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) {
          const float* dp = fb + 32 * (ks >> 1) + 16 * (ks & 1);
          const f32x4v d0 = *reinterpret_cast<const f32x4v*>(dp), d1 = *reinterpret_cast<const f32x4v*>(dp + 8);
          V bf;
#pragma unroll
          for (int j = 0; j < 8; ++j) bf[j] = Op<OT>::cvt(c < TR ? (j < 4 ? d0[j] : d1[j - 4]) : 0.0f);
          g0 = Op<OT>::mfma(GA0[ks], bf, g0);
          g1 = Op<OT>::mfma(GA1[ks], bf, g1);
        }
        if (c < TR) {            // (the partial sums in FX_X are consumed: G fh of ray c goes there)
#pragma unroll
          for (int n = 0; n < 16; ++n) {
            fx[FX_X + c * HID + 64 * w + acc_row(n, h)] = g0[n];
            fx[FX_X + c * HID + 64 * w + 32 + acc_row(n, h)] = g1[n];
          }
        }
      }
      __syncthreads();
      T256(17);
#pragma unroll 1
      for (int q2 = 0; q2 < TR; ++q2) {
        const long rayq = tile * TR + q2;
        if (rayq >= a.R) break;                                            // (wave-uniform: padding rays of the last tile)
        const long rr = (long)k * a.R + rayq;
        const float fhv = fx[FX_FH + q2 * HID + tid];
        const float gf = fx[FX_X + q2 * HID + tid];                        // (G fh)[tid]
        const float* rin = a.rayin + rr * (HID + 2);
        const float uu = rin[tid], wbf = wbv[tid];
        const float su = seg_sum<64>(fhv * uu), sg_ = seg_sum<64>(fhv * gf), sw = seg_sum<64>(fhv * wbf);
        if (lane == 0) { fx[FX_RED + 4 * w] = su; fx[FX_RED + 4 * w + 1] = sg_; fx[FX_RED + 4 * w + 2] = sw; }
        __syncthreads();
        float fu = 0.f, fGf = 0.f, fwb = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { fu += fx[FX_RED + 4 * ww]; fGf += fx[FX_RED + 4 * ww + 1]; fwb += fx[FX_RED + 4 * ww + 2]; }
        const float Oq = fx[FX_RQ + 8 * q2], m1 = fx[FX_RQ + 8 * q2 + 1];
        const float bb = wbv[HID], beta = rin[HID], ngv = rin[HID + 1];
        const float dotFg = fu + Oq * beta;
        const float nF2 = fmaxf(fGf + 2.0f * Oq * fwb + Oq * Oq * bb, 0.0f);
        const float nF = fmaxf(sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
        const float cosv = dotFg / (nF * ngc);
        const float gam = -a.feat_scaling * m1 * inv1;                    // d total / d cos
        const float ar = gam / (nF * ngc), cr = -gam * cosv / (nF * nF);
        fx[FX_DFH + q2 * HID + tid] = ar * uu + cr * (gf + Oq * wbf);      // d total / d fh
        a.rayfeat[rr * (HID + 3) + tid] = fhv;
        a.X1[rr * (HID + 1) + tid] = ar * fhv;
        a.X2[rr * (HID + 1) + tid] = cr * fhv;
        if (tid == 0) {
          float* rf = a.rayfeat + rr * (HID + 3);
          rf[HID] = Oq; rf[HID + 1] = ar; rf[HID + 2] = cr;
          a.X1[rr * (HID + 1) + HID] = ar * Oq;
          a.X2[rr * (HID + 1) + HID] = cr * Oq;
          fx[FX_RQ + 8 * q2 + 2] = ar * beta + cr * (fwb + Oq * bb);       // d total / d opacity (feature part)
          l_f += m1 * (1.0f - cosv) * inv1;
        }
        __syncthreads();
      }
      __syncthreads();
      T256(19);
      {
        // per sample: the feature part of d loss / d weight = gof + d fh . hf, and hf -> the pre-activation gradient of the
        // feature layer relu'(hf) w_s d fh (times the gradient scale) IN PLACE: `hin` becomes B7's operand
        float pp = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) {
          const float* dp = fx + FX_DFH + qw * HID + 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
          const f32x4v d0 = *reinterpret_cast<const f32x4v*>(dp), d1 = *reinterpret_cast<const f32x4v*>(dp + 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float hv = (float)hin[ks][j];
            const float dv = j < 4 ? d0[j] : d1[j - 4];
            pp = fmaf(dv, hv, pp);
            hin[ks][j] = Op<OT>::cvt(hv > 0.0f ? (wsm * dv) * gs : 0.0f);
          }
        }
        pp += __shfl_xor(pp, 32, 64);
        if (h == 0 && valid) fx[FX_DWV + st_idx] = fx[FX_RQ + 8 * qw + 2] + pp;
        else if (h == 0) fx[FX_DWV + st_idx] = 0.0f;
        // d hf goes back to the hand-off buffer (hf is dead): B7H / B7X read their operand from there, one LDS read per
        // k-step -- held in registers next to their accumulators it spilled (and this compiler's AGPR rewrite crashes
        // on spills); B7H does not write the buffer, so it survives until B6 fills it with d hc
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) {
          *reinterpret_cast<V*>(hbuf + ks * PIECE) = hin[ks];
          __builtin_nontemporal_store(hin[ks], act_base(11, ks));
        }
      }
      __syncthreads();
      T256(20);
      composite(std::integral_constant<int, 2>{});
    }
    __syncthreads();
    T256(9);
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
    // layer `mlayer`, stored as the weight-gradient operand `5 + mlayer` and handed to the next GEMM through hbuf;
    // pipelined like the forward layers (the mask word of a block is fetched one stage before its epilogue)
    // mode_tag (feature loss): 0 = as described; 1 = B7H, the feature layer's share of d h4 -- NOT masked (the colour layer's
    // share is still to come), not handed over (the hand-off buffer is B6's next), parked in the workspace slot of d_pre4
    // with ordinary stores; 2 = B5H with that share added back in the epilogue (read one stage ahead, like the mask word,
    // from the very addresses the block's result then overwrites).
    auto bwd_layer = [&](auto seq_tag, auto nk_tag, const int mlayer, auto&& bsel, auto mode_tag) __attribute__((always_inline)) {
      constexpr int NK = decltype(nk_tag)::value, Q = decltype(seq_tag)::value, MODE = decltype(mode_tag)::value;
      constexpr int SK = NK - 1 < 8 ? NK - 1 : 8;
      auto mask_bits = [&](const int b) __attribute__((always_inline)) -> uint32_t {
        if constexpr (MODE == 1) return 0x00ff00ffu;
        return s_mask[(mlayer * 4 + (b >> 1)) * NTHR + tid] >> (8 * (b & 1));
      };
      auto finish = [&](const int pb, const uint32_t (&wd)[8], GV* dst) __attribute__((always_inline)) {
        const uint4 u0 = make_uint4(wd[0], wd[1], wd[2], wd[3]), u1 = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
        if constexpr (MODE == 1) {
          *dst = f0;
          *(dst + 64) = f1;
        } else {
          *reinterpret_cast<V*>(hbuf + (2 * pb) * PIECE) = f0;
          *reinterpret_cast<V*>(hbuf + (2 * pb + 1) * PIECE) = f1;
          __builtin_nontemporal_store(f0, dst);
          __builtin_nontemporal_store(f1, dst + 64);
        }
      };
      uint32_t mb = 0;                                     // mask bits of the block whose epilogue comes next
      // MODE 2: block b's first accumulator chain STARTS from the parked share of block b (unpacked to fp32 into the
      // accumulator that has just been drained, as the forward layers start theirs from the bias rows): no register beyond
      // the accumulator itself carries it (a prefetched copy next to the epilogue words spilled, and this compiler's AGPR
      // rewrite crashes on spills)
      // (requested a stage and a half before its use -- eight registers of raw words -- : one stage ahead left ~1 K cycles
      // of the L2 round trip exposed in every stage)
      uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = make_uint4(0, 0, 0, 0);
      auto request_part = [&](const int b) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
          const GV* src = act_base(5 + mlayer, 2 * b);
          pf0 = __builtin_bit_cast(uint4, *src); pf1 = __builtin_bit_cast(uint4, *(src + 64));
        }
      };
      auto load_part = [&](f32x16& acc) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
          const uint4 g0 = pf0, g1 = pf1;
          float x0, x1;
          Unpack<OT>::get(g0.x, x0, x1); acc[0] = x0; acc[1] = x1;   Unpack<OT>::get(g0.y, x0, x1); acc[2] = x0; acc[3] = x1;
          Unpack<OT>::get(g0.z, x0, x1); acc[4] = x0; acc[5] = x1;   Unpack<OT>::get(g0.w, x0, x1); acc[6] = x0; acc[7] = x1;
          Unpack<OT>::get(g1.x, x0, x1); acc[8] = x0; acc[9] = x1;   Unpack<OT>::get(g1.y, x0, x1); acc[10] = x0; acc[11] = x1;
          Unpack<OT>::get(g1.z, x0, x1); acc[12] = x0; acc[13] = x1; Unpack<OT>::get(g1.w, x0, x1); acc[14] = x0; acc[15] = x1;
        }
      };
      auto stage = [&](auto np2_tag, const int blk, f32x16& cur, f32x16& cur1, f32x16& prv, f32x16& prv1) __attribute__((always_inline)) {
        constexpr int NP2 = decltype(np2_tag)::value;
        uint32_t wd[8];
        const int pb = (blk + 7) & 7;
        const uint32_t bits = mb;
        auto piece = [&](const int i) __attribute__((always_inline)) {
          wd[i] = pk_mask(pk_cvt<OT>(prv[2 * i] + prv1[2 * i], prv[2 * i + 1] + prv1[2 * i + 1]), (bits >> i) & 0x00010001u);
        };
        auto sd = [&](auto ks_tag) __attribute__((always_inline)) {
          constexpr int KS = decltype(ks_tag)::value;
          if constexpr (KS < SK) piece(KS);
          if constexpr (KS == SK) {
#pragma unroll
            for (int i = SK; i < 8; ++i) piece(i);
            finish(pb, wd, blk > 0 ? act_base(5 + mlayer, 2 * pb) : (GV*)park);
            mb = mask_bits(blk);
            if (blk < 7) load_part(prv);                   // prv becomes the next block's first chain
            if (blk < 6) request_part(blk + 2);
          }
          dma_side(np2_tag, nk_tag, ks_tag, std::false_type{}, strm.cw[cls_of(NP2)]);
          OBJ256_SCHED_FENCE();
        };
        T256(0);
        KT::template block_mma<NK, MODE != 2, true>(cur, cur1, ring_addr(), bsel, sd);
        asm volatile("" : "+v"(mb));
        T256(3);
        KT::template stage_sync<NW, NP2, 2>(strm.last);
        strm.template advance<NP2>();
        T256(4);
      };
      f32x16 accA, accA1, accB = zero16(), accB1 = zero16();
      request_part(0);
      load_part(accA);
      request_part(1);
#pragma unroll 1
      for (int blk = 0; blk < 6; blk += 2) {
        stage(nk_tag, blk, accA, accA1, accB, accB1);
        stage(nk_tag, blk + 1, accB, accB1, accA, accA1);
      }
      stage(std::integral_constant<int, NP2_OF(Q, 6)>{}, 6, accA, accA1, accB, accB1);
      stage(std::integral_constant<int, NP2_OF(Q, 7)>{}, 7, accB, accB1, accA, accA1);
      {
        uint32_t wd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          wd[i] = pk_mask(pk_cvt<OT>(accB[2 * i] + accB1[2 * i], accB[2 * i + 1] + accB1[2 * i + 1]), (mb >> i) & 0x00010001u);
        finish(7, wd, act_base(5 + mlayer, 14));
      }
    };
    // slot-gradient blocks (d x1 / d x2): NB blocks accumulated into xacc[b]
    auto bwd_slots = [&](auto seq_tag, auto nb_tag, auto keep_tag, f32x16* xacc, auto from_hbuf_tag) __attribute__((always_inline)) {
      // keep_tag: the chains continue from xacc instead of starting at zero
      constexpr int NB = decltype(nb_tag)::value, Q = decltype(seq_tag)::value;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 odd = zero16();
        auto bs = [&](int ks) -> V {
          if constexpr (decltype(from_hbuf_tag)::value) return *reinterpret_cast<const V*>(hbuf + ks * PIECE);
          else return hin[ks];
        };
        // (b is a constant after unrolling; the switch only turns it into one the templates can take)
        auto run = [&](auto b_tag) __attribute__((always_inline)) {
          constexpr int B = decltype(b_tag)::value;
          constexpr int G2 = SQT::stg_base(Q) + B + 2;           // the stage requested here
          constexpr int NP2 = SQT::stage_pieces(G2);
          if constexpr (G2 >= N_STAGES) {
            // the next tile's first stages: nothing after the workgroup's last tile; the image restarts (next object's
            // image after the object's last tile)
            if constexpr (G2 == N_STAGES) { if (live) strm.set_source((const char*)a.img + next_obj * IMG_BYTES); }
            const int cnt = live ? strm.cw[cls_of(NP2)] : 0;
            plain_stage(std::integral_constant<int, NP2>{}, std::integral_constant<int, 16>{}, std::true_type{}, keep_tag, cnt, xacc[b], odd, bs);
            xacc[b] += odd;
            if (live) KT::template stage_sync<NW, NP2, 0>(strm.last);
            else KT::template sync_imm<0>();
          } else {
            plain_stage(std::integral_constant<int, NP2>{}, std::integral_constant<int, 16>{}, std::false_type{}, keep_tag,
                        strm.cw[cls_of(NP2)], xacc[b], odd, bs);
            xacc[b] += odd;
            KT::template stage_sync<NW, NP2, 0>(strm.last);
          }
          strm.template advance<NP2>();
        };
        if (b == 0) run(std::integral_constant<int, 0>{});
        else if (b == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
      }
    };

    if constexpr (FEAT) {
      // B7H: the feature layer's share of d h4 = W_fl[:, :H]^T d hf (hin = d hf), parked unmasked in d_pre4's slot;
      // B7X: its share of d x2, whose chain rule (embedding.py:49-52, linear in d x2) is applied right away
      bwd_layer(std::integral_constant<int, B7H>{}, std::integral_constant<int, 16>{}, 3,
                [&](int ks) -> V { return *reinterpret_cast<const V*>(hbuf + ks * PIECE); }, std::integral_constant<int, 1>{});
      T256(21);
      f32x16 xa[2];
      bwd_slots(std::integral_constant<int, B7X>{}, std::integral_constant<int, 2>{}, std::false_type{}, xa, std::true_type{});
      float vh[11], vl[11], dpv[11];
      project(vh, vl);
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) {                     // slot u = 2 dd + (f - 4)
        float sv[2], cv[2];
        rev_ladder<4, 2>(vh[dd], vl[dd], sv, cv);
        float dpf = 0.0f;
#pragma unroll
        for (int f = 4; f < 6; ++f) {
          const int u = 2 * dd + (f - 4);
          dpf = fmaf(xa[u >> 4][u & 15], cv[f - 4] * (OBJ_PI_F * (float)(1 << f)), dpf);
        }
        dpv[dd] = (valid && (dd < 10 || h == 0)) ? dpf * inv_gs : 0.0f;
      }
      // straight into d B (linear in d x2): eleven more values alive across B6 and B5H (where the colour layer's share
      // of d x2 joins) spilled
      db_add(dpv);
      T256(22);
    }
    {   // B6 (ONE stage, eight one-piece blocks): d hc = W_oc^T d colour, masked by hc's ReLU bits -> hbuf, tensor 9
      constexpr int NP2 = NP2_OF(B6, 0);
      constexpr int CD = quota(NP2, NW);
      const uint32_t addr = ring_addr();
      V wa[8];
      KT::template rd<0>(wa[0], addr); KT::template rd<PIECE>(wa[1], addr); KT::template rd<2 * PIECE>(wa[2], addr);
      KT::template rd<3 * PIECE>(wa[3], addr); KT::template rd<4 * PIECE>(wa[4], addr); KT::template rd<5 * PIECE>(wa[5], addr);
      KT::template rd<6 * PIECE>(wa[6], addr); KT::template rd<7 * PIECE>(wa[7], addr);
#ifdef OBJ256_ASM_LDS_READS
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(wa[4]), "+v"(wa[5]), "+v"(wa[6]), "+v"(wa[7]));
#endif
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = Op<OT>::mfma(wa[4 * half + i], dh, zero16());
        if (half == 0) {
          strm.template issue<NP2, 0, false>(strm.cw[cls_of(NP2)]); strm.template issue<NP2, 1, false>(strm.cw[cls_of(NP2)]);
          strm.template issue<NP2, 2, false>(strm.cw[cls_of(NP2)]);
        } else {
          strm.template issue<NP2, 3, false>(strm.cw[cls_of(NP2)]); strm.template issue<NP2, 4, false>(strm.cw[cls_of(NP2)]);
          strm.template issue<NP2, 5, false>(strm.cw[cls_of(NP2)]);
        }
        static_assert(CD <= 6, "transfers of the B6 stage");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int blk = 4 * half + i;
          const uint32_t bits = s_mask[(4 * 4 + (blk >> 1)) * NTHR + tid] >> (8 * (blk & 1));
          uint32_t wd[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wd[e] = pk_mask(pk_cvt<OT>(c[i][2 * e], c[i][2 * e + 1]), (bits >> e) & 0x00010001u);
          const uint4 u0 = make_uint4(wd[0], wd[1], wd[2], wd[3]), u1 = make_uint4(wd[4], wd[5], wd[6], wd[7]);
          const V f0 = __builtin_bit_cast(V, u0), f1 = __builtin_bit_cast(V, u1);
          *reinterpret_cast<V*>(hbuf + (2 * blk) * PIECE) = f0;
          *reinterpret_cast<V*>(hbuf + (2 * blk + 1) * PIECE) = f1;
          GV* dst = act_base(9, 2 * blk);
          __builtin_nontemporal_store(f0, dst);
          __builtin_nontemporal_store(f1, dst + 64);
        }
      }
      KT::template stage_sync<NW, NP2, 16>(strm.last);
      strm.template advance<NP2>();
    }
    reload();                                                                                                                  // hin = d hc
    T256(10);
    bwd_layer(std::integral_constant<int, B5H>{}, std::integral_constant<int, 17>{}, 3,
              [&](int ks) -> V { return ks < KS_H ? hin[ks] : dh; }, std::integral_constant<int, FEAT ? 2 : 0>{});             // d h4
    T256(7);
    float dproj[11];
#pragma unroll
    for (int dd = 0; dd < 11; ++dd) dproj[dd] = 0.f;
    {                                                                                                                          // B5X: d x2
      f32x16 xa[2];
      bwd_slots(std::integral_constant<int, B5X>{}, std::integral_constant<int, 2>{}, std::false_type{}, xa, std::false_type{});
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) {                     // slot u = 2 dd + (f - 4)
        float sv[2], cv[2];
        rev_ladder<4, 2>(vh[dd], vl[dd], sv, cv);
#pragma unroll
        for (int f = 4; f < 6; ++f) {
          const int u = 2 * dd + (f - 4);
          dproj[dd] = fmaf(xa[u >> 4][u & 15], cv[f - 4] * (OBJ_PI_F * (float)(1 << f)), dproj[dd]);     // embedding.py:49-52
        }
      }
    }
    T256(11);
    reload();                                                                                                                  // hin = d h4
    bwd_layer(std::integral_constant<int, B4>{}, std::integral_constant<int, 16>{}, 2, [&](int ks) -> V { return hin[ks]; }, std::integral_constant<int, 0>{});  // d h3
    reload();
    bwd_layer(std::integral_constant<int, B3H>{}, std::integral_constant<int, 16>{}, 1, [&](int ks) -> V { return hin[ks]; }, std::integral_constant<int, 0>{}); // d h2
    auto pe_bwd_x1 = [&](const f32x16* x1a) __attribute__((always_inline)) {
      float vh[11], vl[11];
      project(vh, vl);
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) {                       // slot u = 4 dd + f
        float sv[4], cv[4];
        rev_ladder<0, 4>(vh[dd], vl[dd], sv, cv);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int u = 4 * dd + f;
          dproj[dd] = fmaf(x1a[u >> 4][u & 15], cv[f] * (OBJ_PI_F * (float)(1 << f)), dproj[dd]);
        }
      }
    };
    T256(7);
    {                                                                                                                          // B3X: d x1
      f32x16 x1a[3];
      bwd_slots(std::integral_constant<int, B3X>{}, std::integral_constant<int, 3>{}, std::false_type{}, x1a, std::false_type{});
      pe_bwd_x1(x1a);          // (the chain rule is linear in d x1: applied per contribution)
    }
    T256(12);
    reload();                                                                                                                  // hin = d h2
    bwd_layer(std::integral_constant<int, B2>{}, std::integral_constant<int, 16>{}, 0, [&](int ks) -> V { return hin[ks]; }, std::integral_constant<int, 0>{});  // d h1
    reload();
    T256(7);
    {                                                                                                                          // B1: d x1 +=
      f32x16 x1a[3];
      bwd_slots(std::integral_constant<int, B1>{}, std::integral_constant<int, 3>{}, std::false_type{}, x1a, std::false_type{});
      pe_bwd_x1(x1a);
    }
    T256(13);
    // d B[j][c] += d proj_j * t_c: per-lane sums, reduced over the lanes once per object (flush_object)
    {
      float dpv[11];
#pragma unroll
      for (int dd = 0; dd < 11; ++dd) dpv[dd] = (valid && (dd < 10 || h == 0)) ? dproj[dd] * inv_gs : 0.0f;
      db_add(dpv);
    }
    if (++tile_i == a.ntile) { tile_i = 0; ++k_i; }
    T256(14);
  }
  flush_object();
#ifdef OBJ256_TIMING
  T256(5);
  if (blockIdx.x == 0 && tid == 0)
    printf("t256 (ticks of workgroup 0, wave 0): tile head %llu | fwd-layer mma %llu sync %llu | bwd-layer mma %llu sync %llu | layer tails (last epilogue, reload) %llu | "
           "alpha + colour stages %llu | compositing %llu | dh + B6 %llu | B5X + pe %llu | B3X + pe %llu | B1 + pe %llu | tile tail %llu | rest %llu %llu\n",
           tm_[6], tm_[1], tm_[2], tm_[3], tm_[4], tm_[7], tm_[8], tm_[9], tm_[10], tm_[11], tm_[12], tm_[13], tm_[14], tm_[0], tm_[5]);
  if (FEAT && blockIdx.x == 0 && tid == 0)
    printf("t256 feat: F7 tail %llu | compositing 1 + reload %llu | reduce-scatter %llu | G fh %llu | ray loop %llu | d hf %llu | (compositing 2 = "
           "`compositing` above) | B7H tail %llu | B7X + pe %llu\n", tm_[15], tm_[18], tm_[16], tm_[17], tm_[19], tm_[20], tm_[21], tm_[22]);
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
//   6 Lf   d_pre_f x [h4 | x2]        (8 x 10; the feature layer, only with the feature loss: zero parts otherwise)
constexpr int NTYPE = 7;
__host__ __device__ constexpr int type_tiles(int t) { return t == 1 ? 88 : t == 3 ? 88 : t == 4 ? 24 : t == 5 ? 8 : t == 6 ? 80 : 64; }
__host__ __device__ constexpr int type_rows(int t) { return t == 3 ? 9 * 32 : t == 5 ? 32 : 8 * 32; }      // row sums
__host__ __device__ constexpr int type_slab(int t) { return type_tiles(t) * 1024 + type_rows(t); }
__host__ __device__ constexpr int type_pieces(int t) {      // 1-KB pieces per sample group
  return t == 1 ? 16 + 16 + KS_X1 : t == 3 ? 16 + 1 + 16 + KS_X2 : t == 4 ? 16 + KS_X1 : t == 5 ? 1 + 16 : t == 6 ? 16 + 16 + KS_X2 : 32;
}
struct WgArgs {
  int K;
  int parts[NTYPE];
  int prefix[NTYPE + 1];        // workgroups per object before type t
  long slab_prefix[NTYPE + 1];  // floats per object before type t's slabs
  const char* ws; float* slabs;
  WsLay wl;
};
#ifndef OBJ256_WG_CH
#define OBJ256_WG_CH 2      // sample groups per chunk of the plain weight-gradient tasks (1: half the bytes in flight)
#endif
constexpr int IMG_PITCH = 72;                       // bytes per sample row of a 32-feature block image
constexpr int IMG_BLK = 32 * IMG_PITCH;             // 2304

// the compiler's builtin: it tracks the LGKM return itself (no hand-counted s_waitcnt behind the reads any more)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const uint32_t addr) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(uintptr_t)addr);
  return __builtin_bit_cast(uint2, v);
}

// RB / CB: row / column blocks of the task; NRG x NCG waves; AK / BK: 0 hidden fragments (2 per block), 1 head, 2 x slots
template <typename OT, int RB, int CB, int NRG, int NCG, int AK, int BK, int KSB>
__device__ __forceinline__ void wgrad_task(const WgArgs& a, const char* Asrc, const char* Bsrc, const long sg0, const long sg1,
                                           float* slab, char* lds) {
  typedef typename Op<OT>::V V;
  constexpr int RBW = RB / NRG, CBW = CB / NCG;
  constexpr int NPA = AK == 0 ? 2 * RB : 1, NPB = BK == 0 ? 2 * CB : KSB;
  constexpr int CH = OBJ256_WG_CH;                             // sample groups per chunk
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
          { const u32x4v t_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(src)); stg[i] = make_uint4(t_[0], t_[1], t_[2], t_[3]); }   // streamed once
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
          // (the compiler waits for the batch of reads where their results are first used)
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
        { const u32x4v t_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(src)); stg[i] = make_uint4(t_[0], t_[1], t_[2], t_[3]); }   // streamed once
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
  while (t + 1 < NTYPE && r >= a.prefix[t + 1]) ++t;         // (a type without parts has an empty range: never selected)
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
    case 5: wgrad_task<OT, 1, 8, 1, 8, 1, 0, 0>(a, dh, act(4), sg0, sg1, slab, lds); break;
    default: wgrad_cat<OT, KS_X2, false>(act(11), dh, act(3), x2, sg0, sg1, slab, lds); break;
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
  int feat;
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
    } else if (a.feat && i >= L.fl_w && i < L.fl_b) {    // feature layer (task 6: tiles (o >> 5) * 10 + column block)
      const int e = (int)(i - L.fl_w), o = e / (HID + E2), c = e % (HID + E2);
      if (c < HID) v = tile(6, (o >> 5) * 10 + (c >> 5), o & 31, c & 31);
      else { const int pos = x2_col_pos(c - HID); v = tile(6, (o >> 5) * 10 + 8 + (pos >> 10), o & 31, pos & 1023); }
    } else if (a.feat && i >= L.fl_b && i < L.fl_b + HID) v = rowsum(6, (int)(i - L.fl_b));
    else has = false;                                    // (the 512-d head: featg_finish_kernel; without gt_feat: no gradient)
    if (has) a.grads[(long)k * a.p_stride + i] = v * a.inv_gs;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    float sl = 0.f;
    if (threadIdx.x < (a.feat ? 4 : 3))
      for (int g = 0; g < NWG_A; ++g) sl += a.part[((long)k * NWG_A + g) * PART_FLOATS + 64 + threadIdx.x];
    a.loss_terms[k * 4 + threadIdx.x] = sl;
    if (sl > 100000.0f) atomicOr(a.status, 1);           // render_rays.py:109-111
    if (!(fabsf(sl) <= 3.0e38f)) atomicOr(a.status, 2);  // NaN / Inf: reported, not fatal (same word as finalize_kernel)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
  WsLay wl;
  int parts[NTYPE]; int prefix[NTYPE + 1]; long slab_prefix[NTYPE + 1];
  size_t off_img, off_part, off_slabs, off_ws, off_dummy, off_feat, off_gimg, bytes;
};
#ifndef OBJ256_NW
#define OBJ256_NW 4          // waves per workgroup of kernel A where the ray length allows (S <= 32 NW)
#endif
static int waves_for(int) { return OBJ256_NW; }
static Plan make_plan(int K, long n, int S, bool feat = false, int R = 0, int C = 0) {
  Plan p;
  p.wl = WsLay::make(n, 32 * waves_for(S), feat);
  static const long target_mb = [] { const char* e = getenv("OBJ256_WG_TARGET_MB"); return e ? atol(e) : 48L; }();   // (diagnostic override)
  const long target = target_mb << 20;                // bytes one weight-gradient workgroup streams
  p.prefix[0] = 0; p.slab_prefix[0] = 0;
  for (int t = 0; t < NTYPE; ++t) {
    const long bytes = p.wl.nsg * type_pieces(t) * PIECE;
    long parts = (bytes + target - 1) / target;
    if (parts < 1) parts = 1;
    if (parts > p.wl.nsg / 2) parts = p.wl.nsg / 2 > 0 ? p.wl.nsg / 2 : 1;
    if (t == 6 && !feat) parts = 0;                     // the feature layer's task
    p.parts[t] = (int)parts;
    p.prefix[t + 1] = p.prefix[t] + (int)parts;
    p.slab_prefix[t + 1] = p.slab_prefix[t] + parts * type_slab(t);
  }
  size_t o = 0;
  p.off_img = o; o += al256((size_t)K * (feat ? SQ<true>::IMG_BYTES : SQ<false>::IMG_BYTES));
  p.off_part = o; o += al256((size_t)K * NWG_A * PART_FLOATS * 4);
  p.off_slabs = o; o += al256((size_t)K * p.slab_prefix[NTYPE] * 4);
  p.off_ws = o; o += al256((size_t)K * p.wl.obj_bytes);
  p.off_dummy = o; o += al256((size_t)NWG_A * 8 * 2048);      // kernel A: parking area of the stale stores of a layer's first stage
  p.off_feat = o;
  if (feat) o += al256(objgen::feat_head_workspace_bytes(K, R, HID, C));      // the hoisted 512-d head's buffers (objnerf_generic.hip)
  p.off_gimg = o;
  if (feat) o += al256((size_t)K * GIMG_PIECES * PIECE);                      // G as an operand image
  p.bytes = o;
  return p;
}

bool applicable(const objnerf_net* net, const objnerf_train_args* a) {
  if (net->hidden != HID || net->n_freqs != 6) return false;
  if (!(a->mode & (OBJNERF_TRAIN_BF16 | OBJNERF_TRAIN_FP16))) return false;
  if (a->relu_masks || a->emb_debug) return false;
  if (a->gt_feat && (OBJ256_NW != 4 || net->feat_dim % 4 != 0)) return false;   // (the feature step maps 256 threads to the 256 features)
  const int S = a->S;
  return S == 32 || S == 64 || S == 128;
}
size_t workspace_bytes(int K, int R, int S, int feat, int C) {
  return make_plan(K, (long)R * S, S, feat != 0, R, C).bytes + 256;
}

template <typename OT, int S, int NW, bool FEAT>
static void launch_fwd(const FwdArgs& fa, hipStream_t st) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)fwd256_kernel<OT, S, NW, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, l_total(NW));
  });
  hipLaunchKernelGGL((fwd256_kernel<OT, S, NW, FEAT>), dim3(NWG_A), dim3(NW * 64), l_total(NW), st, fa);
}
// (defined in objnerf_train256r.hip)
void launch_fwdr_bf16(const FwdArgs& fa, int S, hipStream_t st);
void launch_fwdr_fp16(const FwdArgs& fa, int S, hipStream_t st);
template <typename OT> static void launch_fwdr_any(const FwdArgs& fa, int S, hipStream_t st) {
  if (std::is_same<OT, __bf16>::value) launch_fwdr_bf16(fa, S, st);
  else launch_fwdr_fp16(fa, S, st);
}
// The row-split form of kernel A (objnerf_train256r_body.h) serves the steps without the feature loss (5 % faster than
// fwd256_kernel, same results on every specification shape); OBJ256_FIRST_FORM=1 (diagnostic) selects fwd256_kernel for
// them too.  Steps with the feature loss run fwd256_kernel<.., true>.
static bool use_row_split() {
  static const bool v = [] { const char* e = getenv("OBJ256_FIRST_FORM"); return !(e && e[0] == '1'); }();
  return v;
}
template <typename OT, bool FEAT>
static int run(const objnerf_net* net, const objnerf_train_args* a, hipStream_t st) {
  const int K = a->K;
  const long n = (long)a->R * a->S;
  const int C = net->feat_dim;
  const Plan p = make_plan(K, n, a->S, FEAT, a->R, C);
  if (a->workspace_bytes < p.bytes) return OBJNERF_EINVAL;
  char* base = (char*)a->workspace;
  int64_t off[OBJNERF_N_TENSORS + 1];
  objnerf_param_layout(net, off);
  Lay256 L;
  L.in_w = (int)off[0]; L.in_b = (int)off[1]; L.m1_w = (int)off[2]; L.m1_b = (int)off[3]; L.cat_w = (int)off[4];
  L.cat_b = (int)off[5]; L.m2_w = (int)off[6]; L.m2_b = (int)off[7]; L.a_w = (int)off[8]; L.a_b = (int)off[9];
  L.cl_w = (int)off[10]; L.cl_b = (int)off[11]; L.oc_w = (int)off[12]; L.oc_b = (int)off[13]; L.pe_b = (int)off[18];
  L.fl_w = (int)off[14]; L.fl_b = (int)off[15];
  OT* img = (OT*)(base + p.off_img);
  float* part = (float*)(base + p.off_part);
  float* slabs = (float*)(base + p.off_slabs);
  char* ws = base + p.off_ws;
  // diagnostic (tools/c5_cumask.py): OBJ256_ONLY=A issues kernel A's half of the step (pack, head preparation, kernel A), =B
  // the other half (kernel B, finalize, head gradients) on data an earlier full step left in the workspace.  A half step
  // returns wrong gradients by design, so the switch exists only in a -DOBJ256_DIAG build (tools/build_timing_lib.sh);
  // the production library never reads the variable.
#ifdef OBJ256_DIAG
  const char* only_e = getenv("OBJ256_ONLY");
#else
  const char* only_e = nullptr;
#endif
  const bool do_a = !(only_e && only_e[0] == 'B'), do_b = !(only_e && only_e[0] == 'A');
  if (do_a) {
  (void)hipMemsetAsync(a->status, 0, sizeof(int), st);
  (void)hipMemsetAsync(part, 0, (size_t)K * NWG_A * PART_FLOATS * 4, st);
  }
  if (do_a) {
    const long tot = (long)K * SQ<FEAT>::N_PIECES * 64;
    hipLaunchKernelGGL((pack256_kernel<OT, FEAT>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, K, a->params,
                       (long)a->p_stride, L, img);
  }
  objgen::FeatHead fh;
  if (FEAT && !do_a) fh = objgen::feat_head_carve(base + p.off_feat, K, a->R, HID, C);
  if (FEAT && do_a) {
    // the hoisted 512-d head: per object G = W_of^T W_of (+ wb, bb), per ray u = W_of^T g, beta, |g| -- ahead
    // of kernel A, with the operand type of the mode (16-bit GEMM operands, like the layer-wise path)
    fh = objgen::feat_head_carve(base + p.off_feat, K, a->R, HID, C);
    const int rc = objgen::feat_head_prep(st, K, a->R, HID, C, a->params, (long)a->p_stride, off[16], off[17], a->gt_feat, fh,
                                          (a->mode & OBJNERF_TRAIN_FP16) ? 2 : 1);
    if (rc) return rc;
    const long tot = (long)K * GIMG_PIECES * 64;
    hipLaunchKernelGGL((packg256_kernel<OT>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, K, fh.gram,
                       (long)HID * HID + HID + 1, (OT*)(base + p.off_gimg));
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
  fa.img = img; fa.ws = ws; fa.dummy = base + p.off_dummy; fa.part = part; fa.L = L; fa.wl = p.wl;
  fa.feat_scaling = a->feat_scaling;
  fa.gimg = base + p.off_gimg;
  fa.rayin = fh.rayin; fa.gram = fh.gram; fa.rayfeat = fh.rayfeat; fa.X1 = fh.X1; fa.X2 = fh.X2;
  if (do_a) {
#ifdef OBJ256_ONE          // diagnostic builds: one instantiation (compile time)
  if (!FEAT && use_row_split()) launch_fwdr_any<OT>(fa, 128, st);
  else launch_fwd<OT, 128, OBJ256_NW, FEAT>(fa, st);
#else
  if (!FEAT && use_row_split()) launch_fwdr_any<OT>(fa, a->S, st);
  else
  switch (a->S) {
    case 32: launch_fwd<OT, 32, OBJ256_NW, FEAT>(fa, st); break;
    case 64: launch_fwd<OT, 64, OBJ256_NW, FEAT>(fa, st); break;
    default: launch_fwd<OT, 128, OBJ256_NW, FEAT>(fa, st); break;
  }
#endif
  }
  if (!do_b) return hipGetLastError() != hipSuccess ? OBJNERF_ELAUNCH : OBJNERF_OK;
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
  fn.inv_gs = 1.0f / gs; fn.L = L; fn.feat = FEAT ? 1 : 0;
  hipLaunchKernelGGL(finalize256_kernel, dim3((unsigned)((fn.P + 255) / 256), (unsigned)K), dim3(256), 0, st, fn);
  if (FEAT) {
    // d W_of, d b_of from the rays' moments (two GEMMs over the rays + featg_finish_kernel)
    const int rc = objgen::feat_head_grads(st, K, a->R, HID, C, a->params, (long)a->p_stride, off[16], off[17], a->gt_feat, fh,
                                           a->grads, (a->mode & OBJNERF_TRAIN_FP16) ? 2 : 1);
    if (rc) return rc;
  }
  if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH;
  return OBJNERF_OK;
}

int train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream) {
  const bool feat = a->gt_feat != nullptr;
#ifndef OBJ256_ONE
  if (a->mode & OBJNERF_TRAIN_FP16) return feat ? run<_Float16, true>(net, a, (hipStream_t)stream) : run<_Float16, false>(net, a, (hipStream_t)stream);
#endif
  return feat ? run<__bf16, true>(net, a, (hipStream_t)stream) : run<__bf16, false>(net, a, (hipStream_t)stream);
}

#else   // OBJ256_ROWSPLIT_TU

#include "objnerf_train256r_body.h"

template <typename OT, int S>
static void launch_fwdr(const FwdArgs& fa, hipStream_t st) {
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)fwdr256_kernel<OT, S>, hipFuncAttributeMaxDynamicSharedMemorySize, R_TOTAL);
  });
  static const int nwg = [] { const char* e = getenv("OBJ256_NWG"); const int v = e ? atoi(e) : NWG_A; return (v >= 8 && v <= NWG_A && v % 8 == 0) ? v : NWG_A; }();
  hipLaunchKernelGGL((fwdr256_kernel<OT, S>), dim3(nwg), dim3(256), R_TOTAL, st, fa);       // (OBJ256_NWG: diagnostic, tools/c5_cumask.py)
}
template <typename OT> static void launch_fwdr_s(const FwdArgs& fa, int S, hipStream_t st) {
#ifdef OBJ256_ONE
  launch_fwdr<OT, 128>(fa, st);
#else
  switch (S) {
    case 32: launch_fwdr<OT, 32>(fa, st); break;
    case 64: launch_fwdr<OT, 64>(fa, st); break;
    default: launch_fwdr<OT, 128>(fa, st); break;
  }
#endif
}
void launch_fwdr_bf16(const FwdArgs& fa, int S, hipStream_t st) { launch_fwdr_s<__bf16>(fa, S, st); }
void launch_fwdr_fp16(const FwdArgs& fa, int S, hipStream_t st) {
#ifdef OBJ256_ONE
  launch_fwdr_s<__bf16>(fa, S, st);
#else
  launch_fwdr_s<_Float16>(fa, S, st);
#endif
}
#endif  // OBJ256_ROWSPLIT_TU

}  // namespace obj256
