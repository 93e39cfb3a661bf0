// Second generation of the bf16-operand fused training iteration (OBJNERF_TRAIN_BF16, hidden 32, 64 samples per ray, no
// feature loss: BASELINE configs[1]'s kernel).  Same arithmetic specification as objnerf_train_bf16.hip (operands of
// every hidden nn.Linear rounded to bf16, fp32 accumulation, fp32 compositing / losses; model.py:61-103,
// render_rays.py:6-63, loss.py:5-103), plus the head gradients rounded to bf16 as weight-gradient operands -- the
// rounding points of the fused hidden-256 kernels (oracle: round_head_grads).  What changed, and why: the first
// generation spent 60 % of a tile with neither pipe issuing (profiles/r03_pmc_bf16_v3.txt) -- eight workgroup barriers
// per 128-sample tile, the compositing on two of eight waves while six waited, three rounds of 2-byte staging stores.
//
//  * ONE staging image per tile, [sample][feature] in bf16 (992-byte rows): every activation block and every
//    pre-activation gradient block is written ONCE, as the packed MFMA operand the lane already holds (one
//    ds_write_b128 per 32-feature block instead of eight converts + eight 2-byte stores), and the weight gradients read
//    it back with ds_read_b64_tr_b16 -- gfx950's transposing LDS read turns [sample][feature] rows into operands whose
//    contraction index is the SAMPLE.  Sample <-> k-slot: lane group g, element e <-> sample 32 st + phi(g, e): a half's
//    8 rows are consecutive samples, whose 992-byte pitch (= 56 dwords mod 64) puts them on disjoint banks.
//  * No transposed weight images: the input-gradient operands W^T come from the FORWARD images through the same
//    transposing read (rows = outputs phi(g, e), the lane's 8 k-slots; 16-byte column chunks chosen so that the 16
//    result rows are the input features 16 T .. 16 T + 15 in natural order).  35 KB of LDS go to the staging image.
//  * TWO barriers per tile.  Tile t: forward (all waves) | barrier | waves 0-1 composite tile t's two rays WHILE waves
//    2-7 run ALL weight-gradient MFMAs of tile t-1 from the staging image | barrier | backward (all waves; writes
//    tile t's staging image).  The weight gradients are one round of 28 tile pairs -- the five layers, the two 32x32
//    layers' bias sums (against the embedding's constant-1 entry) and the head weights (a 4-row operand [d alpha,
//    d colour] against h4 / hc) -- so no per-lane head / bias partial sums exist any more.
//
// Weight-gradient accumulators: 4 tile pairs per wave (32 registers); waves 0-1 take 2 pairs after their compositing.  LDS: forward images 29.0 KB +
// small vectors 1.25 KB + per-sample heads 2 KB + staging 124 KB = 160 000 B.
//
// This file is the kernel's SOURCE, compiled twice: objnerf_train_bf16v2.hip (V2_FEAT 0: RGB + depth + opacity loss,
// BASELINE configs[1]) and objnerf_train_bf16v2f.hip (V2_FEAT 1: + the 512-d feature-distillation loss with the head
// hoisted past the compositing; configs[2] / [3]).  The two differ in the LDS layout (see below), so
// the layout constants and work lists are per-instantiation.
#ifndef V2_FEAT
#error "include through objnerf_train_bf16v2.hip / objnerf_train_bf16v2f.hip"
#endif
#if V2_FEAT && !defined(V2_RECOMPUTE_PE)
#define V2_RECOMPUTE_PE 1    // (the feature instantiation has no registers for the cached embedding / cosine factors)
#endif
#if !V2_FEAT && !defined(V2_EARLY_FETCH)
#define V2_LATE_FETCH 1      // next tile's point requested at the tile's end (measured: 2.293 ms against 2.311 a tile ahead;
#endif                       // the feature instantiation: 3.88 against 3.84 -- so each keeps its better one)
#if V2_FEAT
#define V2_KERNEL train_fused_bf16v2f_kernel
#else
#define V2_KERNEL train_fused_bf16v2_kernel
#endif
#define OBJ_HW_SINCOS 1      // embedding sin / cos on the transcendental unit (see objnerf_device.h)
#include <utility>
#include "objnerf_bf16_common.h"
#include "../../include/objnerf_hip.h"

namespace objtrain {
namespace {
using namespace bf16k;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- LDS layout (bytes)
#if !V2_FEAT
constexpr int B_SMALL = FWD_IMG_END;                 // fp32 small vectors
constexpr int B_SM = B_SMALL + SMALL_BYTES;          // s_alpha | s_col[3] (fp32), overwritten by d alpha | d colour
constexpr int B_STG = B_SM + 4 * 4 * TS;
constexpr int PITCH = 992;                           // one sample's row; 248 dwords = 56 (mod 64)
constexpr int LDS_BYTES = B_STG + TS * PITCH;
static_assert(B_SMALL % 16 == 0 && B_STG % 16 == 0 && LDS_BYTES <= 163840, "bf16 v2 lds layout");
// a sample's row (byte offsets): activations, then pre-activation gradients, then the head gradients
constexpr int F_X1 = 0;        // 3 blocks (96 embedding entries in K-order)
constexpr int F_X2 = 192;      // 2 blocks (48 entries + 16 zeros)
constexpr int F_H1 = 320, F_H2 = 384, F_H3 = 448, F_H4 = 512, F_HC = 576;
constexpr int F_DH1 = 640, F_DH2 = 704, F_DH3 = 768, F_DH4 = 832, F_DHC = 896;
constexpr int F_HEAD = 960;    // (d alpha, d c0, d c1, d c2) + 12 zeros
static_assert(F_HEAD + 32 == PITCH, "row");
constexpr int F_DHF = -1;      // (no feature layer)
#else
// With the feature layer: its forward image (7 KB) and the fp32 scratch of the feature compositing join, and the row
// loses what the 160 KB cannot hold -- the head-gradient operand and hc (the head weights' gradients are per-lane sums
// again, as in the first generation) and the 16 padding entries of x2 (stored as its three 16-entry K-tiles).
constexpr int B_FL = FWD_IMG_END;                    // feature layer, forward image (as B_CL)
constexpr int B_SMALL = B_FL + 32 * RS_CL;
constexpr int B_SM = B_SMALL + SMALL_BYTES;
constexpr int B_FS = B_SM + 4 * 4 * TS;              // fp32 scratch of the feature compositing (float offsets below)
constexpr int FS_W = 0;                              //   s_w   [128]       ray weight of every sample
constexpr int FS_PART = FS_W + TS;                   //   s_part [8][32]    per-wave partial composited hidden
constexpr int FS_GFH = FS_PART + NWAVE * 32;         //   s_gfh [2][32]     d loss / d fh of the tile's rays
constexpr int FS_GOF = FS_GFH + 64;                  //   s_gof [4]: d loss / d O (feature part) [2] | O [2]
constexpr int FS_DWF = FS_GOF + 16;                  //   s_dwf [128]       feature part of d loss / d weight
constexpr int FS_FHB = FS_DWF + TS;                  //   s_fhb [8][64]     per-wave exchange: fh | gfh
constexpr int FS_FLOATS = FS_FHB + NWAVE * 64;
constexpr int B_STG = (B_FS + FS_FLOATS * 4 + 15) / 16 * 16;
constexpr int PITCH = 928;                           // 232 dwords = 40 (mod 64) = 8 x odd
constexpr int LDS_BYTES = B_STG + TS * PITCH;
static_assert(B_SMALL % 16 == 0 && B_STG % 16 == 0 && LDS_BYTES <= 163840, "bf16 v2 feature lds layout");
constexpr int F_X1 = 0;        // 3 blocks
constexpr int F_X2 = 192;      // 3 K-tiles of 16 entries (natural order: position 4 g + r of tile T <-> kappa 16 T + 4 g + r)
constexpr int F_H1 = 288, F_H2 = 352, F_H3 = 416, F_H4 = 480;
constexpr int F_DH1 = 544, F_DH2 = 608, F_DH3 = 672, F_DH4 = 736, F_DHC = 800, F_DHF = 864;
static_assert(F_DHF + 64 == PITCH, "row");
constexpr int F_HC = -1, F_HEAD = -2;
#endif
constexpr int X1_ONE_TILE = F_X1 + 128;   // the 16 positions of the embedding that hold its constant-1 entry (position 13)

__device__ __forceinline__ s16x4 lds_tr(const char* p) {
#ifdef V2_PLAIN_READS   // diagnostic: same traffic through the plain 8-byte read (wrong operands)
  return *reinterpret_cast<const s16x4*>(p);
#else
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
#endif
}
__device__ __forceinline__ bf16x8 join(const s16x4 lo, const s16x4 hi) {
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// MFMA operand (rows = 16 consecutive positions at byte `off` of the row image, k = 8 rows of the lane's group) from a
// row-major image: base = image + (4 g + q) * pitch + 8 p (lane 4 q + p of its group), second half 16 rows further
template <int PITCH_>
__device__ __forceinline__ bf16x8 tr_operand(const char* base, const int off) {
  return join(lds_tr(base + off), lds_tr(base + 16 * PITCH_ + off));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(const float lo, const float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float unpack_lo(const unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float unpack_hi(const unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// the 8 values of a packed block as fp32 (element e = 4 tt + r <-> feature 16 tt + 4 g + r)
__device__ __forceinline__ void unpack8(const bf16x8 hp, float (&o)[8]) {
  const u32x4 u = __builtin_bit_cast(u32x4, hp);
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[2 * j] = unpack_lo(u[j]); o[2 * j + 1] = unpack_hi(u[j]); }
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// The packed 16-bit integer instructions below are inline assembly on plain dwords: written as <2 x i16> vector
// operations on elements of a bit-cast bf16x8, this compiler (HIP 7.2.26015) used the FIRST pair's result for all four
// pairs of a block (tools/bf16v2_diag.py caught it: every gradient off by 10 % - 170 %).  Their inputs are VALU results
// and their outputs feed VALU / LDS / MFMA-operand reads, none of which needs software wait states on gfx950.
__device__ __forceinline__ unsigned pk_max_i16_zero(const unsigned v) {
  unsigned o;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(o) : "v"(v));
  return o;
}
// d with the halves zeroed where the bf16 activation h (a ReLU output: >= 0 as an integer) is zero: d * min(h, 1) on
// the 16-bit halves (op_sel_hi 0 on the constant: its low half serves both)
__device__ __forceinline__ unsigned pk_keep_where_nonzero(const unsigned d, const unsigned h) {
  unsigned m, o;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(h));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(d), "v"(m));
  return o;
}
// ReLU + pack of an activation block nobody reads in fp32: round first, then max(bits, 0) on the 16-bit halves (a
// negative bf16 is a negative int16): 4 + 4 instructions instead of 8 + 4
__device__ __forceinline__ bf16x8 relu_pack(const T32& x) {
  u32x4 u;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int h = 0; h < 2; ++h) u[2 * tt + h] = pk_max_i16_zero(pack2(x.t[tt][2 * h], x.t[tt][2 * h + 1]));
  return __builtin_bit_cast(bf16x8, u);
}
// pack(d * (h > 0)) for a pre-activation gradient that is only ever used as an MFMA operand: round first, then the
// two packed instructions above per pair instead of two compares and two selects.  Rounding commutes with the mask
// (0 rounds to 0).
__device__ __forceinline__ bf16x8 mask_pack(const T32& gr, const bf16x8 hp) {
  const u32x4 h = __builtin_bit_cast(u32x4, hp);
  u32x4 u;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      u[2 * tt + hf] = pk_keep_where_nonzero(pack2(gr.t[tt][2 * hf], gr.t[tt][2 * hf + 1]), h[2 * tt + hf]);
  return __builtin_bit_cast(bf16x8, u);
}
// two chain-rule factors as an fp16 pair (|factor| <= 32 pi): v_fma_mix_f32 takes either half as an operand, so the
// backward pass needs no unpacking.  Opaque to the compiler on purpose: it saw through a bf16 pack / unpack pair and
// re-derived every factor (doubling formula, scale, rounding) in the backward pass -- five instructions per factor.
__device__ __forceinline__ unsigned pack_h2(const float lo, const float hi) {
  const f32x2 v = {lo, hi};
  unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
#ifdef V2_PACK_RELAXED
  asm("" : "+v"(u));
#else
  asm volatile("" : "+v"(u));      // (volatile: 2.185 ms; as a plain asm the scheduler moves the packs and the kernel is 1.3 % slower)
#endif
  return u;
}
__device__ __forceinline__ float fma_lo(const float x, const unsigned hpair, const float acc) {
  return fmaf(x, (float)__builtin_bit_cast(f16x2, hpair)[0], acc);
}
__device__ __forceinline__ float fma_hi(const float x, const unsigned hpair, const float acc) {
  return fmaf(x, (float)__builtin_bit_cast(f16x2, hpair)[1], acc);
}
// ---- weight-gradient work list: (d block, activation tile) pairs of a wave; kind 1 = the head operand against two tiles
struct Slot { int d, a, kind; };
template <int W> struct WaveSlots;
#if V2_FEAT
constexpr int NSLOT = 5;     // tile pairs per wave
#else
constexpr int NSLOT = 4;
#endif
#if !V2_FEAT
// 28 pairs = 6 waves x 4 + the two compositing waves x 2
template <> struct WaveSlots<0> { static constexpr int n = 2; static constexpr Slot s[NSLOT] = {
  {F_DHC, F_X2 + 64, 0}, {F_DHC, F_X2 + 96, 0}, {0, 0, 0}, {0, 0, 0}}; };
template <> struct WaveSlots<1> { static constexpr int n = 2; static constexpr Slot s[NSLOT] = {
  {F_HEAD, F_H4, 1}, {F_HEAD, F_HC, 1}, {0, 0, 0}, {0, 0, 0}}; };
template <> struct WaveSlots<4> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DHC, F_H4, 0}, {F_DHC, F_H4 + 32, 0}, {F_DHC, F_X2 + 0, 0}, {F_DHC, F_X2 + 32, 0}}; };
#else
// 30 pairs: the five layers as above (x2 now three tiles), the feature layer d hf x [h4 | x2] (5), the two bias pairs --
// five per wave on waves 2 - 7, in two halves of the contraction: samples 0 - 63 beside the compositing waves' part (i),
// samples 64 - 127 beside their part (iv) (they have no other work in either)
template <> struct WaveSlots<0> { static constexpr int n = 0; static constexpr Slot s[NSLOT] = {}; };
template <> struct WaveSlots<1> { static constexpr int n = 0; static constexpr Slot s[NSLOT] = {}; };
template <> struct WaveSlots<2> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DH1, F_X1 + 0, 0}, {F_DH1, F_X1 + 32, 0}, {F_DH1, F_X1 + 64, 0}, {F_DH1, F_X1 + 96, 0}, {F_DH1, F_X1 + 128, 0}}; };
template <> struct WaveSlots<3> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DH3, F_X1 + 0, 0}, {F_DH3, F_X1 + 32, 0}, {F_DH3, F_X1 + 64, 0}, {F_DH3, F_X1 + 96, 0}, {F_DH3, F_X1 + 128, 0}}; };
template <> struct WaveSlots<4> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DHC, F_H4, 0}, {F_DHC, F_H4 + 32, 0}, {F_DHC, F_X2 + 0, 0}, {F_DHC, F_X2 + 32, 0}, {F_DHC, F_X2 + 64, 0}}; };
template <> struct WaveSlots<5> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DHF, F_H4, 0}, {F_DHF, F_H4 + 32, 0}, {F_DHF, F_X2 + 0, 0}, {F_DHF, F_X2 + 32, 0}, {F_DHF, F_X2 + 64, 0}}; };
template <> struct WaveSlots<6> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DH1, F_X1 + 160, 0}, {F_DH2, F_H1, 0}, {F_DH2, F_H1 + 32, 0}, {F_DH2, X1_ONE_TILE, 0}, {F_DH4, X1_ONE_TILE, 0}}; };
template <> struct WaveSlots<7> { static constexpr int n = 5; static constexpr Slot s[NSLOT] = {
  {F_DH3, F_X1 + 160, 0}, {F_DH3, F_H2, 0}, {F_DH3, F_H2 + 32, 0}, {F_DH4, F_H3, 0}, {F_DH4, F_H3 + 32, 0}}; };
#endif
#if !V2_FEAT
template <> struct WaveSlots<2> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DH1, F_X1 + 0, 0}, {F_DH1, F_X1 + 32, 0}, {F_DH1, F_X1 + 64, 0}, {F_DH1, F_X1 + 96, 0}}; };
template <> struct WaveSlots<3> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DH3, F_X1 + 0, 0}, {F_DH3, F_X1 + 32, 0}, {F_DH3, F_X1 + 64, 0}, {F_DH3, F_X1 + 96, 0}}; };
template <> struct WaveSlots<5> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DH3, F_X1 + 128, 0}, {F_DH3, F_X1 + 160, 0}, {F_DH3, F_H2, 0}, {F_DH3, F_H2 + 32, 0}}; };
template <> struct WaveSlots<6> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DH1, F_X1 + 128, 0}, {F_DH1, F_X1 + 160, 0}, {F_DH2, F_H1, 0}, {F_DH2, F_H1 + 32, 0}}; };
template <> struct WaveSlots<7> { static constexpr int n = 4; static constexpr Slot s[NSLOT] = {
  {F_DH4, F_H3, 0}, {F_DH4, F_H3 + 32, 0}, {F_DH4, X1_ONE_TILE, 0}, {F_DH2, X1_ONE_TILE, 0}}; };
#endif

struct WAcc { f32x4 a[NSLOT][2]; };

// one 32-sample k-step of slot J: base = staging + (4 g + q) * PITCH + 8 p + 32 st * PITCH
template <int W, int J>
__device__ __forceinline__ void wgrad_slot(WAcc& acc, const char* base, bf16x8& a0, bf16x8& a1) {
  using WS = WaveSlots<W>;
  if constexpr (J < WS::n) {
    constexpr Slot sl = WS::s[J];
    if constexpr (sl.kind == 0) {
      if constexpr (J == 0 || WS::s[J > 0 ? J - 1 : 0].d != sl.d || WS::s[J > 0 ? J - 1 : 0].kind != 0) {
        a0 = tr_operand<PITCH>(base, sl.d);              // the pair's two 16-output halves of the gradient block
        a1 = tr_operand<PITCH>(base, sl.d + 32);
      }
      const bf16x8 b = tr_operand<PITCH>(base, sl.a);
      acc.a[J][0] = MFMA_BF16(a0, b, acc.a[J][0]);
      acc.a[J][1] = MFMA_BF16(a1, b, acc.a[J][1]);
    } else {
      if constexpr (J == 0 || WS::s[J > 0 ? J - 1 : 0].kind != 1) a0 = tr_operand<PITCH>(base, sl.d);
      const bf16x8 b0 = tr_operand<PITCH>(base, sl.a), b1 = tr_operand<PITCH>(base, sl.a + 32);
      acc.a[J][0] = MFMA_BF16(a0, b0, acc.a[J][0]);
      acc.a[J][1] = MFMA_BF16(a0, b1, acc.a[J][1]);
    }
  }
}
template <int W, int ST0 = 0, int ST1 = TS / 32>
__device__ __forceinline__ void wgrad_wave(WAcc& acc, const char* lane_base) {
#ifndef V2_WG_UNROLL
#define V2_WG_UNROLL 4
#endif
#pragma unroll V2_WG_UNROLL   // (fully unrolled with all reads hoisted, the first build spilled 200 registers)
  for (int st = ST0; st < ST1; ++st) {
    const char* base = lane_base + st * 32 * PITCH;
    bf16x8 a0, a1;
    wgrad_slot<W, 0>(acc, base, a0, a1);
    wgrad_slot<W, 1>(acc, base, a0, a1);
    wgrad_slot<W, 2>(acc, base, a0, a1);
    wgrad_slot<W, 3>(acc, base, a0, a1);
    if constexpr (NSLOT > 4) wgrad_slot<W, 4>(acc, base, a0, a1);
  }
}

// ---- end of the sweep: accumulators -> this workgroup's partial-gradient slab
// feature of position p_ (0..31) of a packed 32-block: position 8 g + 4 tt + r <-> feature 16 tt + 4 g + r
__device__ __forceinline__ int pos_feat(const int p_) { return 16 * ((p_ >> 2) & 1) + 4 * (p_ >> 3) + (p_ & 3); }

// normal pair: D[out position 16 h + 4 g + r][in position c of the tile at byte `aoff`]
__device__ __forceinline__ void emit_pair(float* slab, const Layout& L, const f32x4 (&acc)[2], const int c, const int g, const int doff,
                          const int aoff) {
  int w_off, ncols, b_off, hid;
  if (doff == F_DH1) { w_off = L.in_w; ncols = OBJ_E1; b_off = L.in_b; hid = 0; }
  else if (doff == F_DH2) { w_off = L.m1_w; ncols = H; b_off = L.m1_b; hid = 0; }
  else if (doff == F_DH3) { w_off = L.cat_w; ncols = H + OBJ_E1; b_off = L.cat_b; hid = H; }
  else if (doff == F_DH4) { w_off = L.m2_w; ncols = H; b_off = L.m2_b; hid = 0; }
  else if (doff == F_DHF) { w_off = L.fl_w; ncols = H + OBJ_E2; b_off = L.fl_b; hid = H; }
  else { w_off = L.cl_w; ncols = H + OBJ_E2; b_off = L.cl_b; hid = H; }
  int col;                                           // reference column, obj32n::BIAS_COL or obj32n::ZERO_COL
  const bool hidden_in = aoff >= F_H1;
  if (hidden_in) {
    col = pos_feat(((aoff - F_H1) & 63) / 2 + c);
  } else {
    const bool x2 = aoff >= F_X2;
    const int rel = aoff - (x2 ? F_X2 : F_X1);       // byte offset inside the embedding's blocks
#if V2_FEAT
    // x2 as three 16-entry K-tiles in natural order; x1 as packed blocks
    const int kappa = x2 ? 16 * (rel >> 5) + c : 32 * (rel >> 6) + pos_feat(((rel & 63) >> 1) + c);
#else
    const int kappa = 32 * (rel >> 6) + pos_feat(((rel & 63) >> 1) + c);
#endif
    int t_, g_;
    obj32n::kappa_tg(kappa, t_, g_);
    col = x2 ? obj32n::x2_col(t_, g_) : obj32n::x1_col(t_, g_);
    if (x2 && kappa >= 48) col = obj32n::ZERO_COL;   // (the 16 padding entries of the second x2 block)
    // the two 32x32 layers take only their bias from the embedding's constant-1 entry
    if ((doff == F_DH2 || doff == F_DH4) && col != obj32n::BIAS_COL) col = obj32n::ZERO_COL;
    if (col >= 0) col += hid;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = pos_feat(16 * h + 4 * g + r);
      if (col >= 0) slab[w_off + o * ncols + col] = acc[h][r];
      else if (col == obj32n::BIAS_COL) slab[b_off + o] = acc[h][r];
    }
}
// head pair: rows 4 g + r of the head operand (0: d alpha, 1..3: d colour) against the two tiles of h4 / hc
__device__ __forceinline__ void emit_head(float* slab, const Layout& L, const f32x4 (&acc)[2], const int c, const int g, const int aoff) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int f = pos_feat(16 * u + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * g + r;
      if (aoff == F_H4 && row == 0) slab[L.a_w + f] = acc[u][r];
      if (aoff == F_HC && row >= 1 && row < 4) slab[L.oc_w + (row - 1) * H + f] = acc[u][r];
    }
  }
}
template <int W, int J>
__device__ __forceinline__ void emit_slot(float* slab, const Layout& L, const WAcc& acc, const int c, const int g) {
  using WS = WaveSlots<W>;
  if constexpr (J < WS::n) {
    constexpr Slot sl = WS::s[J];
    if constexpr (sl.kind == 0) emit_pair(slab, L, acc.a[J], c, g, sl.d, sl.a);
    else emit_head(slab, L, acc.a[J], c, g, sl.a);
  }
}
template <int W>
__device__ __forceinline__ void emit_wave(float* slab, const Layout& L, const WAcc& acc, const int c, const int g) {
  emit_slot<W, 0>(slab, L, acc, c, g);
  emit_slot<W, 1>(slab, L, acc, c, g);
  emit_slot<W, 2>(slab, L, acc, c, g);
  emit_slot<W, 3>(slab, L, acc, c, g);
  if constexpr (NSLOT > 4) emit_slot<W, 4>(slab, L, acc, c, g);
}

// Tried on top of this and measured slower or equal (round 4, K = 50, R = 4096; product 2.185 ms): the embedding of tile
// t + 1 formed at the end of tile t's backward pass (2.30 - 2.35 ms, 256 registers); one weight-gradient pair on each
// compositing wave and five on waves 2 - 3 (2.190 ms); a single sin / cos anchor per direction (2.177 ms: kept at two); the embedding's
// doubling formulas and factor scaling as packed fp32 over direction pairs (74 fewer instructions per tile; 2.19 -> 2.22 ms
// -- packed fp32 VALU beside the partner wave's MFMAs is the anti-lever MI355X_MICROARCH.md says it is).
// Feature build: the ray term (part iii) on the two compositing waves only, beside the second half of the weight
// gradients on waves 2-7 (one more barrier): what the compositing waves carry from part (i) to part (iv) becomes live
// across the weight-gradient code -- 75 spilled registers with one shared path, 156 with role-specialised paths between
// the tile's barriers.  It needs ~30 registers first (the head weights' per-lane sums as butterfly sums would free them).
// Optional scheduling fences (-DV2_FENCES) at the layer boundaries of the forward / backward passes.  (The first build
// kept the five activation blocks in fp32 across the compositing: 256 registers + 70 spilled, the weight-gradient
// accumulators among them, and their scratch round trip cost 8 000 cycles per tile.  Keeping only the packed operands
// removed the spills; the fences then only cost instruction-level parallelism.)
#ifdef V2_FENCES
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() do {} while (0)      // (measured: 2.42 ms with the fences, 2.35 without)
#endif

constexpr int NRED2 = 8;        // per wave: a_b, oc_b[3], loss terms [3], spare

#ifdef PHASE_TIMING
__device__ unsigned long long g_phase_b2[8][24];
#define PT_INIT() unsigned long long pt_acc[16]; for (int i_ = 0; i_ < 16; ++i_) pt_acc[i_] = 0; \
  unsigned long long pt_t0 = __builtin_amdgcn_s_memtime()
#define PT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_acc[i] += t_ - pt_t0; pt_t0 = t_; } while (0)
#define PT_FLUSH() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 16; ++i_) g_phase_b2[w][i_] = pt_acc[i_]; } while (0)
#else
#define PT_INIT() do {} while (0)
#define PT(i) do {} while (0)
#define PT_FLUSH() do {} while (0)
#endif

__global__ __launch_bounds__(NTHR) void V2_KERNEL(const TrainDev a) {
  constexpr int S = 64, TR = TS / S;
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.x / a.G, gi = blockIdx.x % a.G;
  const float* sm = reinterpret_cast<const float*>(ldsb + B_SMALL);
  float* s_alpha = reinterpret_cast<float*>(ldsb + B_SM);
  float* s_col = s_alpha + TS;
  char* stg = ldsb + B_STG;

  for (int i = tid; i < LDS_BYTES / 4; i += NTHR) reinterpret_cast<float*>(ldsb)[i] = 0.0f;
  __syncthreads();
#if V2_FEAT
  stage_forward_bf16(ldsb, reinterpret_cast<float*>(ldsb + B_SMALL), a.params + (long)k * a.p_stride, a.L, tid, true, B_FL);
  float* fs = reinterpret_cast<float*>(ldsb + B_FS);
  float* s_w = fs + FS_W;
  float* s_part = fs + FS_PART;
  float* s_gfh = fs + FS_GFH;
  float* s_gof = fs + FS_GOF;
  float* s_dwf = fs + FS_DWF;
  const float* Gg = a.gram + (long)k * GRAM;       // G [32][32] | wb [32] | bb of this object (L1 / L2-resident)
#else
  stage_forward_bf16(ldsb, reinterpret_cast<float*>(ldsb + B_SMALL), a.params + (long)k * a.p_stride, a.L, tid, false, 0);
#endif
  __syncthreads();

  const float inv_scale = 1.0f / a.scale[k];
  const int R = a.R;
  const float n1 = (float)a.counts[2 * k], n2 = (float)a.counts[2 * k + 1];
  int bflag0, bflag1;
  batch_flags(a, bflag0, bflag1);
  const float inv1 = bflag0 ? 0.0f : 1.0f / (n1 + 1e-10f);
  const float inv2 = bflag1 ? 0.0f : 1.0f / (n2 + 1e-10f);

  WAcc acc;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) acc.a[j][0] = acc.a[j][1] = zero4();
  float g_ba = 0.f, g_boc0 = 0.f, g_boc1 = 0.f, g_boc2 = 0.f;
  float dB[6][3];                           // d B[4 i + g][x], summed over this lane's samples
#pragma unroll
  for (int i = 0; i < 6; ++i) dB[i][0] = dB[i][1] = dB[i][2] = 0.f;
  float l_d = 0.f, l_c = 0.f, l_o = 0.f;
#if V2_FEAT
  float l_f = 0.f;
  float hW[4][8];                           // d W_alpha, d W_oc[0..2] as per-lane sums; [.][4 tt + r] <-> feature 16 tt + 4 g + r
#pragma unroll
  for (int s_ = 0; s_ < 8; ++s_) hW[0][s_] = hW[1][s_] = hW[2][s_] = hW[3][s_] = 0.f;
#endif
  const SegRows sg = SegRows::make(64, lane);

  // per-lane bases, as macros over an opaque copy of the lane id that is re-defined at the phase boundaries: addresses
  // are recomputed next to their use instead of living in registers across the tile (lane 4 q + p of a 16-lane group
  // supplies row q, 8-byte chunk p of a transposing read)
  int lane_l = lane;
#define RELAUNDER() asm volatile("" : "+v"(lane_l))
#define V2_TILE_SYNC() do { if (!((V2_ABL) & 2)) __syncthreads(); } while (0)
#define c (lane_l & 15)
#define g (lane_l >> 4)
#define q4 ((lane_l >> 2) & 3)
#define p4 (lane_l & 3)
  // Staging rows are SWIZZLED (round 5): the 16-byte chunk at byte offset o of sample row R sits at o ^ 16 when bit 2
  // of R is set.  A ds_write_b128 is serviced in groups of 8 consecutive lanes = 8 consecutive samples at one column
  // offset, bank = (a / 4) mod 32 (MI355X_MICROARCH.md, LDS): with the 992- / 928-byte pitches (24 / 8 dwords mod 32,
  // which the transposing reads need: 8 x odd mod 64) rows R and R + 4 started on the same bank -- every staging store
  // was a 2-way conflict (SQ_LDS_BANK_CONFLICT 29 % of SQ_LDS_IDX_ACTIVE, profiles/r04_pmc_bf16_v4.txt).  Moving rows
  // 4..7 of a group by one chunk puts the eight stores on eight disjoint bank quads; a transposing read's 32-lane
  // group reads the same two chunks of rows 4..7 in swapped order, so its banks are unchanged.  (Reader: row 32 st +
  // 4 g + q (+ 16) -> bit 2 of the row = g & 1; writer: row 16 w + c -> (c >> 2) & 1.)
#define tr_stg (stg + (4 * g + q4) * PITCH + ((8 * p4) ^ ((g & 1) << 4)))   /* staging, weight gradients (k = samples) */
#define stg_swz (((lane_l >> 2) & 1) << 4)                          /* writer's swizzle: bit 2 of its sample row */
#define f_in (ldsb + B_IN + c * RS_IN + 16 * g)                    /* forward A operands (rows = outputs) */
#define f_m1 (ldsb + B_M1 + c * RS_M + 16 * g)
#define f_cat (ldsb + B_CAT + c * RS_CAT + 16 * g)
#define f_m2 (ldsb + B_M2 + c * RS_M + 16 * g)
#define f_cl (ldsb + B_CL + c * RS_CL + 16 * g)
  // transposed (input-gradient) A operands out of the same images: rows = outputs 4 g + q (+16), 16-byte column chunk p
#define t_in (ldsb + B_IN + (4 * g + q4) * RS_IN + 16 * p4)
#define t_m1 (ldsb + B_M1 + (4 * g + q4) * RS_M + 16 * p4)
#define t_cat (ldsb + B_CAT + (4 * g + q4) * RS_CAT + 16 * p4)
#define t_m2 (ldsb + B_M2 + (4 * g + q4) * RS_M + 16 * p4)
#define t_cl (ldsb + B_CL + (4 * g + q4) * RS_CL + 16 * p4)
#if V2_FEAT
#define f_fl (ldsb + B_FL + c * RS_CL + 16 * g)
#define t_fl (ldsb + B_FL + (4 * g + q4) * RS_CL + 16 * p4)
#endif
#define slot (16 * w + c)
#define row_st (stg + slot * PITCH + ((16 * g) ^ stg_swz))           /* this lane's 16 bytes of every block of its sample */
  // W^T d for the 16 input features (block blk, half tt) of an image: chunk p of feature 16 tt + 4 p + j sits at
  // position 8 p + 4 tt + j of the block
#define BWD_TILE(accv, timg, RS_, blk, tt, db) \
  accv = MFMA_BF16(tr_operand<RS_>(timg, 64 * (blk) + 8 * (tt)), db, accv)

  const int q = w >> 2;                                               // ray of the tile this wave's samples belong to

  // A sample's point in two halves: request_point only ISSUES the loads (the point itself, or z and the ray's origin /
  // direction), finish_point consumes them -- up to a whole tile later.  (Measured: the latency of these loads is NOT what
  // the "load + project" phase of tools/phase_timing.py shows; requesting a tile ahead moves the kernels by 1 % either
  // way.  That phase is the second wave of each SIMD running beside the first one's forward pass.)
  auto request_point = [&](const int tile_, float (&raw)[7]) {
    // branch-free (a join of branches made the compiler wait for the loads right there): a tile past the end reads the
    // object's last ray and finish_point discards it; with points given, the second / third pointers re-read the point
    const int ray_ = min(tile_ * TR + q, R - 1);
    const long rr = (long)k * R + ray_;
    const long sidx = rr * S + (slot & (S - 1));
    const bool pm = a.pts != nullptr;
    const float* p0 = pm ? a.pts + sidx * 3 : a.origins + rr * 3;
    const float* p1 = pm ? p0 : a.dirs + rr * 3;
    const float* p2 = pm ? p0 : a.z + sidx;
    raw[0] = p0[0]; raw[1] = p0[1]; raw[2] = p0[2];
    raw[3] = p1[0]; raw[4] = p1[1]; raw[5] = p1[2];
    raw[6] = p2[0];
  };
  auto finish_point = [&](const int tile_, const float (&raw)[7], float& x, float& y, float& z_) {
    const bool ok = tile_ < a.NT && tile_ * TR + q < R;
    const bool pm = a.pts != nullptr;
    const float px = pm ? raw[0] : (raw[0] + raw[3] * raw[6]) - a.obj_center;
    const float py = pm ? raw[1] : (raw[1] + raw[4] * raw[6]) - a.obj_center;
    const float pz = pm ? raw[2] : (raw[2] + raw[5] * raw[6]) - a.obj_center;
    x = ok ? px : 0.f; y = ok ? py : 0.f; z_ = ok ? pz : 0.f;
  };

#ifdef V2_PRIO      // diagnostic: static issue priority for the younger half of the workgroup (waves 4-7)
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(V2_PRIO);
#endif
  float nx, ny, nz;
  {
    float raw0[7];
    request_point(gi, raw0);
    finish_point(gi, raw0, nx, ny, nz);
  }
  bool have_prev = false;
  PT_INIT();
  for (int tile = gi; tile < a.NT; tile += a.G) {
    asm volatile("" ::: "memory");      // (the weight images are loop-invariant: keep their reads inside the tile body)
    RELAUNDER();
    const int ray0 = tile * TR;
    const bool valid = ray0 + q < R;
    // ---------------------------------------------------------------- 1. forward (model.py:61-103 on embedding.py:46-55)
    obj32n::Pe32 pe;
    pe_project_b(sm, g, nx, ny, nz, inv_scale, pe);
    float raw[7];                            // the next tile's point, requested a whole tile ahead
#ifndef V2_LATE_FETCH
    request_point(tile + a.G, raw);
#endif
    PT(0);
    bf16x8 h1p, h2p, h3p, h4p, hcp;          // the activations as the packed operands every consumer sees
#if V2_FEAT
    bf16x8 hfp;                              // hidden of the feature branch (model.py:98-100)
#endif
#ifndef V2_RECOMPUTE_PE
    // the embedding blocks and the chain-rule factors d sin(2^f a) / d proj = cos(2^f a) pi 2^f of the lane's six
    // directions (fp16 pairs; zero where the slot holds no direction) stay in registers for the backward pass: it then
    // needs no transcendental and no range reduction at all (they were ~20 % of the kernel's VALU instructions)
    bf16x8 xb1[3], xb2[2];
    unsigned cpk[18];
#endif
    {
#ifdef V2_RECOMPUTE_PE
      bf16x8 xb1[3], xb2[2];
#endif
      {
        // forward-only embedding (the backward re-creates it tile by tile), packed block by block
        f32x4 x2v[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          f32x4 xv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = 2 * b + u;
            float sn[6], cs[6];
#ifdef V2_RECOMPUTE_PE
            obj32n::pe32_octaves<0, 5, false>(pe.vh[i], pe.vl[i], sn, cs);
#else
            obj32n::pe32_octaves<0, 5, true>(pe.vh[i], pe.vl[i], sn, cs);
            const float live = (i == 5 && g != 0) ? 0.0f : OBJ_PI_F;      // only group 0 has a sixth direction
#pragma unroll
            for (int f = 0; f < 6; f += 2)
              cpk[3 * i + (f >> 1)] = pack_h2(cs[f] * (live * (float)(1 << f)), cs[f + 1] * (live * (float)(2 << f)));
#endif
            xv[u] = obj32n::pe32_x1_tile(pe, i, g, sn);
            float v4, v5;
            obj32n::pe32_x2_pair(i, g, sn, v4, v5);
            x2v[b][2 * u] = v4;
            x2v[b][2 * u + 1] = v5;
          }
          xb1[b] = pack8(xv[0], xv[1]);
          SCHED_FENCE();
        }
        xb2[0] = pack8(x2v[0], x2v[1]);
        xb2[1] = pack8(x2v[2], zero4());
      }
      T32 av = zero32();
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_IN>(av, f_in, b, xb1[b]);
      h1p = relu_pack(av);
      SCHED_FENCE();
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) av.t[tt][r] = sm[S_BM1 + 16 * tt + 4 * g + r];
      fwd_blk<RS_M>(av, f_m1, 0, h1p);
      h2p = relu_pack(av);
      av = zero32();
      fwd_blk<RS_CAT>(av, f_cat, 0, h2p);
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_CAT>(av, f_cat, 1 + b, xb1[b]);
      h3p = relu_pack(av);
      SCHED_FENCE();
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) av.t[tt][r] = sm[S_BM2 + 16 * tt + 4 * g + r];
      fwd_blk<RS_M>(av, f_m2, 0, h3p);
      const T32 h4 = relu32(av);
      h4p = pack32(h4);
      av = zero32();
      fwd_blk<RS_CL>(av, f_cl, 0, h4p);
      fwd_blk<RS_CL>(av, f_cl, 1, xb2[0]);
      fwd_blk<RS_CL>(av, f_cl, 2, xb2[1]);
      const T32 hc = relu32(av);
      hcp = pack32(hc);
#if V2_FEAT
      av = zero32();
      fwd_blk<RS_CL>(av, f_fl, 0, h4p);
      fwd_blk<RS_CL>(av, f_fl, 1, xb2[0]);
      fwd_blk<RS_CL>(av, f_fl, 2, xb2[1]);
      hfp = relu_pack(av);
#endif
      SCHED_FENCE();
      // the heads read the fp32 activations (model.py:88,96)
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
      // every lane group ends with ONE of the sample's four head outputs (one sigmoid per lane)
      const float sa = xgroup_sum(pa), s0 = xgroup_sum(pc0), s1 = xgroup_sum(pc1), s2 = xgroup_sum(pc2);
      const float mine = (g == 0) ? sa : ((g == 1) ? s0 : ((g == 2) ? s1 : s2));
      const float zv = mine + sm[S_HB + g];
      s_alpha[g * TS + slot] = (g == 0) ? zv * 10.0f : sigmoid_acc(zv);       // s_alpha | s_col[0..2] are contiguous
    }
    // ray inputs of the compositing waves, requested BEFORE the barrier so their latency hides behind it
    float pf_zz = 0.f, pf_gtd = 0.f, pf_gr = 0.f, pf_gg = 0.f, pf_gb = 0.f;
    int pf_lab = 2;
    if (w < TR && ray0 + w < R) {
      const long rr = (long)k * R + ray0 + w;
      pf_zz = a.z[rr * S + lane];
      pf_gtd = a.gt_depth[rr];
      pf_gr = a.gt_rgb[rr * 3]; pf_gg = a.gt_rgb[rr * 3 + 1]; pf_gb = a.gt_rgb[rr * 3 + 2];
      pf_lab = a.labels[rr];
    }
#if V2_FEAT
    // this wave's ray: u = W_of^T g (32), beta = b_of . g, |g| (feat_pre_kernel) and its label
    float pf_uh = 0.f, pf_beta = 0.f, pf_ngv = 1.f;
    int pf_lab2 = 2;
    if (valid) {
      const long rr2_ = (long)k * R + ray0 + q;
      pf_uh = a.rayin[rr2_ * RAYIN + (lane_l & 31)];
      pf_beta = a.rayin[rr2_ * RAYIN + 32];
      pf_ngv = a.rayin[rr2_ * RAYIN + 33];
      pf_lab2 = (int)a.labels[rr2_];
    }
#endif
    PT(1);
    V2_TILE_SYNC();
    RELAUNDER();
    PT(2);
    // ---------------------------------------------------------------- 2. waves 0-1: composite + loss of tile t's rays
    //                                                                     (render_rays.py:6-63, loss.py:5-103; fp32);
    //                                                                     waves 2-7: weight gradients of tile t - 1
#if !V2_FEAT
    if (w < TR && !((V2_ABL) & 1)) {
      const int pos = lane;
      const int sl = w * S + pos;
      const bool on = ray0 + w < R;
      float al = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
      const float zz = pf_zz;
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
      const float m1 = (pf_lab == 1) ? 1.0f : 0.0f;
      const float m2 = (pf_lab != 2) ? 1.0f : 0.0f;
      const float tgt = (pf_lab != 0) ? 1.0f : 0.0f;
      const float info = 1.0f / (sqrtf(V) + 1e-4f);
      const float rd = D - pf_gtd, r0 = C0 - pf_gr, r1 = C1 - pf_gg, r2 = C2 - pf_gb, ro = O - tgt;
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
      const float suf = sg.rscan_add(qv, pos) - qv;
      const float docc = dw * T - suf / fr;
      if (on) {
        s_alpha[sl] = 10.0f * (docc * occ * (1.0f - occ));
        s_col[sl] = gC0 * wgt * c0 * (1.0f - c0);
        s_col[TS + sl] = gC1 * wgt * c1 * (1.0f - c1);
        s_col[2 * TS + sl] = gC2 * wgt * c2 * (1.0f - c2);
      }
    }
#if defined(V2_NO_WGRAD) || ((V2_ABL) & 8)
    if (have_prev && a.K < 0) {
#else
    if (have_prev) {         // (the compositing waves take two of the 28 tile pairs each, the other six four)
#endif
      switch (w) {
        case 0: wgrad_wave<0>(acc, tr_stg); break;
        case 1: wgrad_wave<1>(acc, tr_stg); break;
        case 2: wgrad_wave<2>(acc, tr_stg); break;
        case 3: wgrad_wave<3>(acc, tr_stg); break;
        case 4: wgrad_wave<4>(acc, tr_stg); break;
        case 5: wgrad_wave<5>(acc, tr_stg); break;
        case 6: wgrad_wave<6>(acc, tr_stg); break;
        default: wgrad_wave<7>(acc, tr_stg); break;
      }
    }
    PT(3);
#else
    // With the feature loss the compositing has four parts: (i) waves 0-1 composite their ray (waves 2-7: weight
    // gradients of tile t - 1); (ii) every wave forms its 16 samples' share of fh = sum_s w_s hf_s; (iii) every wave
    // evaluates its ray's feature term -- F = W_of fh + b_of O enters only through fh . u, fh^T G fh, fh . wb (DESIGN.md
    // 4.3) -- and the samples' d loss / d weight of it; (iv) waves 0-1 finish the backward of the compositing.
    float cw_occ = 0.f, cw_fr = 1.f, cw_T = 1.f, cw_wgt = 0.f, cw_dw = 0.f;
    float cw_c0 = 0.f, cw_c1 = 0.f, cw_c2 = 0.f, cw_g0 = 0.f, cw_g1 = 0.f, cw_g2 = 0.f;
    if (w < TR) {
      const int pos = lane;
      const int sl = w * S + pos;
      const bool on = ray0 + w < R;
      float al = 0.f;
      const float zz = pf_zz;
      if (on) { al = s_alpha[sl]; cw_c0 = s_col[sl]; cw_c1 = s_col[TS + sl]; cw_c2 = s_col[2 * TS + sl]; }
      cw_occ = on ? sigmoid_acc(al) : 0.0f;
      cw_fr = on ? (1.0f - cw_occ) + 1e-10f : 1.0f;
      const float Pinc = sg.scan_mul(cw_fr, pos);
      cw_T = __shfl_up(Pinc, 1, 64);
      if (pos == 0) cw_T = 1.0f;
      cw_wgt = cw_occ * cw_T;
      const float D = sg.total_add(cw_wgt * zz, pos);
      const float O = sg.total_add(cw_wgt, pos);
      const float C0 = sg.total_add(cw_wgt * cw_c0, pos);
      const float C1 = sg.total_add(cw_wgt * cw_c1, pos);
      const float C2 = sg.total_add(cw_wgt * cw_c2, pos);
      const float dz = zz - D;
      const float V = sg.total_add(cw_wgt * (dz * dz), pos);
      const float m1 = (pf_lab == 1) ? 1.0f : 0.0f;
      const float m2 = (pf_lab != 2) ? 1.0f : 0.0f;
      const float tgt = (pf_lab != 0) ? 1.0f : 0.0f;
      const float info = 1.0f / (sqrtf(V) + 1e-4f);
      const float rd = D - pf_gtd, r0 = C0 - pf_gr, r1 = C1 - pf_gg, r2 = C2 - pf_gb, ro = O - tgt;
      auto sgn = [](float x) { return x > 0.f ? 1.0f : (x < 0.f ? -1.0f : 0.0f); };
      const float gD = m1 * sgn(rd) * info * inv1;
      cw_g0 = a.color_scaling * m1 * sgn(r0) * inv1;
      cw_g1 = a.color_scaling * m1 * sgn(r1) * inv1;
      cw_g2 = a.color_scaling * m1 * sgn(r2) * inv1;
      const float gO = a.opacity_scaling * m2 * sgn(ro) * inv2;
      if (on && pos == 0) {
        l_d += m1 * fabsf(rd) * info * inv1;
        l_c += m1 * (fabsf(r0) + fabsf(r1) + fabsf(r2)) * inv1;
        l_o += m2 * fabsf(ro) * inv2;
        s_gof[2 + w] = O;
      }
      cw_dw = gD * zz + gO + cw_g0 * cw_c0 + cw_g1 * cw_c1 + cw_g2 * cw_c2;
      if (on) s_w[sl] = cw_wgt;
    } else if (have_prev) {
      switch (w) {
        case 2: wgrad_wave<2, 0, 2>(acc, tr_stg); break;
        case 3: wgrad_wave<3, 0, 2>(acc, tr_stg); break;
        case 4: wgrad_wave<4, 0, 2>(acc, tr_stg); break;
        case 5: wgrad_wave<5, 0, 2>(acc, tr_stg); break;
        case 6: wgrad_wave<6, 0, 2>(acc, tr_stg); break;
        default: wgrad_wave<7, 0, 2>(acc, tr_stg); break;
      }
    }
    // (ii) -- the Gram row of part (iii) is requested here: its L2 latency hides behind this part and the barrier
    float hf8[8];
    unpack8(hfp, hf8);
    f32x4 gq[4];
    float wbh, bbv;
    {
      const int half = lane_l >> 5, hh = lane_l & 31;
#pragma unroll
      for (int j = 0; j < 4; ++j) gq[j] = *reinterpret_cast<const f32x4*>(Gg + hh * 32 + 16 * half + 4 * j);
      wbh = Gg[32 * 32 + hh];
      bbv = Gg[32 * 32 + 32];
    }
    PT(3);
    V2_TILE_SYNC();
    RELAUNDER();
    PT(9);
    {
      const float wv = valid ? s_w[slot] : 0.0f;
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = wv * hf8[e];
      const float psum = slot_sums8(v8, c);            // lane c & 7 <-> element e: the sum over the wave's 16 samples
      if (c < 8) s_part[w * 32 + 16 * ((c & 7) >> 2) + 4 * g + (c & 3)] = psum;
    }
    V2_TILE_SYNC();
    RELAUNDER();
    PT(10);
    {
      float* s_fhb = fs + FS_FHB + 64 * w;
      const int half = lane_l >> 5, hh = lane_l & 31;
      const int w0 = w & ~3;
      const float fh = valid ? (s_part[w0 * 32 + hh] + s_part[(w0 + 1) * 32 + hh]) +
                               (s_part[(w0 + 2) * 32 + hh] + s_part[(w0 + 3) * 32 + hh]) : 0.0f;
      if (half == 0) s_fhb[hh] = fh;
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      float Gp = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 16; ++h2) Gp = fmaf(gq[h2 >> 2][h2 & 3], s_fhb[16 * half + h2], Gp);
      const float Gfh = Gp + __shfl_xor(Gp, 32, 64);
      const float uh = pf_uh, beta = pf_beta, ngv = pf_ngv;
      const float O2 = valid ? s_gof[2 + q] : 0.f;
      const float fu = wave_sum32(fh * uh), fGf = wave_sum32(fh * Gfh), fwb = wave_sum32(fh * wbh);
      const float dotFg = fu + O2 * beta;
      const float nF2 = fmaxf(fGf + 2.0f * O2 * fwb + O2 * O2 * bbv, 0.0f);
      const float nF = fmaxf(__builtin_amdgcn_sqrtf(nF2), 1e-8f), ngc = fmaxf(ngv, 1e-8f);
      const float rnn = __builtin_amdgcn_rcpf(nF * ngc);       // (bf16 mode: 1-ulp hardware reciprocals)
      const float cosv = dotFg * rnn;
      const float mm1 = (pf_lab2 == 1) ? 1.0f : 0.0f;
      const float gam = -a.feat_scaling * mm1 * inv1;
      const float ar = gam * rnn, cr = -gam * cosv * __builtin_amdgcn_rcpf(nF * nF);
      const float gfh = ar * uh + cr * (Gfh + O2 * wbh);
      const float gof = ar * beta + cr * (fwb + O2 * bbv);
      if (valid && (w & 3) == 0 && half == 0) {
        const long rr2 = (long)k * R + ray0 + q;
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
        for (int r = 0; r < 4; ++r) dp = fmaf(s_fhb[32 + 16 * tt + 4 * g + r], hf8[4 * tt + r], dp);
      const float dwf = xgroup_sum(dp) + gof;
      if (g == 0 && valid) s_dwf[slot] = dwf;
    }
    V2_TILE_SYNC();
    RELAUNDER();
    PT(11);
    if (w < TR) {
      const int pos = lane;
      const int sl = w * S + pos;
      const bool on = ray0 + w < R;
      if (on) cw_dw += s_dwf[sl];
      const float qv = cw_dw * cw_wgt;
      const float suf = sg.rscan_add(qv, pos) - qv;
      const float docc = cw_dw * cw_T - suf / cw_fr;
      if (on) {
        s_alpha[sl] = 10.0f * (docc * cw_occ * (1.0f - cw_occ));
        s_col[sl] = cw_g0 * cw_wgt * cw_c0 * (1.0f - cw_c0);
        s_col[TS + sl] = cw_g1 * cw_wgt * cw_c1 * (1.0f - cw_c1);
        s_col[2 * TS + sl] = cw_g2 * cw_wgt * cw_c2 * (1.0f - cw_c2);
      }
    } else if (have_prev) {
      switch (w) {
        case 2: wgrad_wave<2, 2, 4>(acc, tr_stg); break;
        case 3: wgrad_wave<3, 2, 4>(acc, tr_stg); break;
        case 4: wgrad_wave<4, 2, 4>(acc, tr_stg); break;
        case 5: wgrad_wave<5, 2, 4>(acc, tr_stg); break;
        case 6: wgrad_wave<6, 2, 4>(acc, tr_stg); break;
        default: wgrad_wave<7, 2, 4>(acc, tr_stg); break;
      }
    }
    PT(12);
#endif
    V2_TILE_SYNC();
    RELAUNDER();
    PT(4);
    // ---------------------------------------------------------------- 3. backward; writes tile t's staging image
    const float da = valid ? s_alpha[slot] : 0.0f;
    const float dc0 = valid ? s_col[slot] : 0.0f;
    const float dc1 = valid ? s_col[TS + slot] : 0.0f;
    const float dc2 = valid ? s_col[2 * TS + slot] : 0.0f;
    if (g == 0) { g_ba += da; g_boc0 += dc0; g_boc1 += dc1; g_boc2 += dc2; }
#if !V2_FEAT
    {                                             // head gradients as a weight-gradient operand: 4 values + 12 zeros
      bf16x4 hv;
      hv[0] = (__bf16)(g == 0 ? da : 0.0f); hv[1] = (__bf16)(g == 0 ? dc0 : 0.0f);
      hv[2] = (__bf16)(g == 0 ? dc1 : 0.0f); hv[3] = (__bf16)(g == 0 ? dc2 : 0.0f);
      *reinterpret_cast<bf16x4*>(stg + slot * PITCH + F_HEAD + ((8 * g) ^ stg_swz)) = hv;
    }
#else
    {                                             // head-weight gradients as per-lane sums (no room for their operands)
      float h4f[8], hcf[8];
      unpack8(h4p, h4f);
      unpack8(hcp, hcf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        hW[0][e] = fmaf(da, h4f[e], hW[0][e]);
        hW[1][e] = fmaf(dc0, hcf[e], hW[1][e]);
        hW[2][e] = fmaf(dc1, hcf[e], hW[2][e]);
        hW[3][e] = fmaf(dc2, hcf[e], hW[3][e]);
      }
    }
    // the embedding is re-created below: hide the angles from CSE, or every sin / cos of the forward stays live
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(pe.vh[i]));
#endif
    float dps[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dps[i] = 0.f;

    T32 d_hc, d_h4;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tt + 4 * g + r;
        const float dv = fmaf(sm[S_WOC + 2 * H + row], dc2, fmaf(sm[S_WOC + H + row], dc1, sm[S_WOC + row] * dc0));
        d_hc.t[tt][r] = dv;
        d_h4.t[tt][r] = sm[S_WA + row] * da;
      }
    *reinterpret_cast<bf16x8*>(row_st + F_H4) = h4p;
#if !V2_FEAT
    *reinterpret_cast<bf16x8*>(row_st + F_HC) = hcp;
#endif
    const bf16x8 d_hc_b = mask_pack(d_hc, hcp);
    *reinterpret_cast<bf16x8*>(row_st + F_DHC) = d_hc_b;
    BWD_TILE(d_h4.t[0], t_cl, RS_CL, 0, 0, d_hc_b);
    BWD_TILE(d_h4.t[1], t_cl, RS_CL, 0, 1, d_hc_b);
#if V2_FEAT
    // d loss / d hf_s = w_s * (d loss / d fh of the sample's ray), through the ReLU; then the feature layer's share of d h4
    bf16x8 d_hf_b;
    {
      const float wv = valid ? s_w[slot] : 0.0f;
      T32 d_hf;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) d_hf.t[tt][r] = valid ? wv * s_gfh[q * 32 + 16 * tt + 4 * g + r] : 0.0f;
      d_hf_b = mask_pack(d_hf, hfp);
    }
    *reinterpret_cast<bf16x8*>(row_st + F_DHF) = d_hf_b;
    BWD_TILE(d_h4.t[0], t_fl, RS_CL, 0, 0, d_hf_b);
    BWD_TILE(d_h4.t[1], t_fl, RS_CL, 0, 1, d_hf_b);
#endif
    PT(5);
    SCHED_FENCE();
    const bf16x8 d_h4_b = mask_pack(d_h4, h4p);
    *reinterpret_cast<bf16x8*>(row_st + F_DH4) = d_h4_b;
    {
      f32x4 x2v[3];
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        f32x4 d_x = zero4();
        BWD_TILE(d_x, t_cl, RS_CL, 1 + (T >> 1), T & 1, d_hc_b);
#if V2_FEAT
        BWD_TILE(d_x, t_fl, RS_CL, 1 + (T >> 1), T & 1, d_hf_b);
#endif
#ifdef V2_RECOMPUTE_PE
        float o0, o1, o2, o3;
        obj32n::pe32_x2_pair_fb(pe, 2 * T, g, d_x[0], d_x[1], dps[2 * T], o0, o1);
        obj32n::pe32_x2_pair_fb(pe, 2 * T + 1, g, d_x[2], d_x[3], dps[2 * T + 1], o2, o3);
        x2v[T] = f32x4{o0, o1, o2, o3};
#else
        dps[2 * T] = fma_hi(d_x[1], cpk[6 * T + 2], fma_lo(d_x[0], cpk[6 * T + 2], dps[2 * T]));
        dps[2 * T + 1] = fma_hi(d_x[3], cpk[6 * T + 5], fma_lo(d_x[2], cpk[6 * T + 5], dps[2 * T + 1]));
#endif
        SCHED_FENCE();
      }
#if V2_FEAT
      // x2 as its three 16-entry K-tiles: this lane's four entries of tile T at position 4 g of the tile
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        bf16x4 xv4;
        xv4[0] = (__bf16)x2v[T][0]; xv4[1] = (__bf16)x2v[T][1]; xv4[2] = (__bf16)x2v[T][2]; xv4[3] = (__bf16)x2v[T][3];
        *reinterpret_cast<bf16x4*>(stg + slot * PITCH + F_X2 + ((32 * T + 8 * g) ^ stg_swz)) = xv4;
      }
#elif defined(V2_RECOMPUTE_PE)
      *reinterpret_cast<bf16x8*>(row_st + F_X2) = pack8(x2v[0], x2v[1]);
      *reinterpret_cast<bf16x8*>(row_st + F_X2 + 64) = pack8(x2v[2], zero4());
#else
      (void)x2v;
      *reinterpret_cast<bf16x8*>(row_st + F_X2) = xb2[0];
      *reinterpret_cast<bf16x8*>(row_st + F_X2 + 64) = xb2[1];
#endif
    }
    PT(6);
    SCHED_FENCE();
    T32 d_h3 = zero32();
    BWD_TILE(d_h3.t[0], t_m2, RS_M, 0, 0, d_h4_b);
    BWD_TILE(d_h3.t[1], t_m2, RS_M, 0, 1, d_h4_b);
    *reinterpret_cast<bf16x8*>(row_st + F_H3) = h3p;
    const bf16x8 d_h3_b = mask_pack(d_h3, h3p);
    *reinterpret_cast<bf16x8*>(row_st + F_DH3) = d_h3_b;
    SCHED_FENCE();
    T32 d_h2 = zero32();
    BWD_TILE(d_h2.t[0], t_cat, RS_CAT, 0, 0, d_h3_b);
    BWD_TILE(d_h2.t[1], t_cat, RS_CAT, 0, 1, d_h3_b);
    *reinterpret_cast<bf16x8*>(row_st + F_H2) = h2p;
    const bf16x8 d_h2_b = mask_pack(d_h2, h2p);
    *reinterpret_cast<bf16x8*>(row_st + F_DH2) = d_h2_b;
    SCHED_FENCE();
    T32 d_h1 = zero32();
    BWD_TILE(d_h1.t[0], t_m1, RS_M, 0, 0, d_h2_b);
    BWD_TILE(d_h1.t[1], t_m1, RS_M, 0, 1, d_h2_b);
    *reinterpret_cast<bf16x8*>(row_st + F_H1) = h1p;
    const bf16x8 d_h1_b = mask_pack(d_h1, h1p);
    *reinterpret_cast<bf16x8*>(row_st + F_DH1) = d_h1_b;
    PT(7);
    SCHED_FENCE();
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 xv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int T = 2 * b + u;
        f32x4 d_x = zero4();
        BWD_TILE(d_x, t_cat, RS_CAT, 1 + b, u, d_h3_b);
        BWD_TILE(d_x, t_in, RS_IN, b, u, d_h1_b);
#ifdef V2_RECOMPUTE_PE
        xv[u] = obj32n::pe32_x1_tile_fb(pe, T, g, d_x, dps[T]);
#else
        dps[T] = fma_hi(d_x[3], cpk[3 * T + 1], fma_lo(d_x[2], cpk[3 * T + 1],
                 fma_hi(d_x[1], cpk[3 * T], fma_lo(d_x[0], cpk[3 * T], dps[T]))));
#endif
        SCHED_FENCE();
      }
#ifdef V2_RECOMPUTE_PE
      *reinterpret_cast<bf16x8*>(row_st + F_X1 + 64 * b) = pack8(xv[0], xv[1]);
#else
      (void)xv;
      *reinterpret_cast<bf16x8*>(row_st + F_X1 + 64 * b) = xb1[b];
#endif
    }
    // d B[j][x] += d proj_j * t_x (embedding.py:48); j = 4 i + g lives in this lane only
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dB[i][0] = fmaf(dps[i], pe.t[0], dB[i][0]);
      dB[i][1] = fmaf(dps[i], pe.t[1], dB[i][1]);
      dB[i][2] = fmaf(dps[i], pe.t[2], dB[i][2]);
    }
#ifdef V2_LATE_FETCH
    request_point(tile + a.G, raw);
#endif
    finish_point(tile + a.G, raw, nx, ny, nz);
    have_prev = true;
    PT(8);
  }
  PT_FLUSH();
  // the last tile's weight gradients
  __syncthreads();
  if (have_prev) {
    switch (w) {
      case 0: wgrad_wave<0>(acc, tr_stg); break;
      case 1: wgrad_wave<1>(acc, tr_stg); break;
      case 2: wgrad_wave<2>(acc, tr_stg); break;
      case 3: wgrad_wave<3>(acc, tr_stg); break;
      case 4: wgrad_wave<4>(acc, tr_stg); break;
      case 5: wgrad_wave<5>(acc, tr_stg); break;
      case 6: wgrad_wave<6>(acc, tr_stg); break;
      default: wgrad_wave<7>(acc, tr_stg); break;
    }
  }
#undef BWD_TILE
#undef c
#undef g
#undef q4
#undef p4
#undef tr_stg
#undef stg_swz
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
#if V2_FEAT
#undef f_fl
#undef t_fl
#endif
#undef slot
#undef row_st
  const int c = lane & 15, g = lane >> 4;
  __syncthreads();                                  // (the reduction scratch below aliases the staging image)

  float* slab = a.slab + ((long)k * a.G + gi) * a.slab_stride;
  const Layout& L = a.L;
  switch (w) {
    case 0: emit_wave<0>(slab, L, acc, c, g); break;
    case 1: emit_wave<1>(slab, L, acc, c, g); break;
    case 2: emit_wave<2>(slab, L, acc, c, g); break;
    case 3: emit_wave<3>(slab, L, acc, c, g); break;
    case 4: emit_wave<4>(slab, L, acc, c, g); break;
    case 5: emit_wave<5>(slab, L, acc, c, g); break;
    case 6: emit_wave<6>(slab, L, acc, c, g); break;
    case 7: emit_wave<7>(slab, L, acc, c, g); break;
    default: break;
  }
  float* red = reinterpret_cast<float*>(stg);       // [NWAVE][NRED2] | d B [NWAVE][72]
  {
    float* mine = red + w * NRED2;
    const float s0 = wave_sum64(g_ba), s1 = wave_sum64(g_boc0), s2 = wave_sum64(g_boc1), s3 = wave_sum64(g_boc2);
    const float e0 = wave_sum64(l_d), e1 = wave_sum64(l_c), e2 = wave_sum64(l_o);
#if V2_FEAT
    const float e3 = wave_sum64(l_f);
#else
    const float e3 = 0.f;
#endif
    if (lane == 0) { mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3; mine[4] = e0; mine[5] = e1; mine[6] = e2; mine[7] = e3; }
#if V2_FEAT
    float* hwp = red + NWAVE * NRED2 + NWAVE * 72 + w * 128;     // [4 heads][32 features] of this wave
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      const int rw = 16 * (s_ >> 2) + 4 * g + (s_ & 3);
      const float v0 = dpp_rowsum16(hW[0][s_]), v1 = dpp_rowsum16(hW[1][s_]);
      const float v2 = dpp_rowsum16(hW[2][s_]), v3 = dpp_rowsum16(hW[3][s_]);
      if (c == 0) { hwp[rw] = v0; hwp[32 + rw] = v1; hwp[64 + rw] = v2; hwp[96 + rw] = v3; }
    }
#endif
    float* dbw = red + NWAVE * NRED2 + w * 72;      // [slot i][g][3] = B's own row-major order
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
    for (int ww = 0; ww < NWAVE; ++ww) v += red[NWAVE * NRED2 + ww * 72 + tid];
    slab[L.pe_b + tid] = v;
  }
  if (tid < NRED2) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[ww * NRED2 + tid];
    if (tid == 0) slab[L.a_b] = v;
    else if (tid < 4) slab[L.oc_b + tid - 1] = v;
    else a.loss_part[((long)k * a.G + gi) * 4 + (tid - 4)] = v;     // (tid == 7: the feature term; zero without it)
  }
#if V2_FEAT
  if (tid < 4 * H) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWAVE; ++ww) v += red[NWAVE * NRED2 + NWAVE * 72 + ww * 128 + tid];
    const int head = tid >> 5, f = tid & 31;
    if (head == 0) slab[L.a_w + f] = v;
    else slab[L.oc_w + (head - 1) * H + f] = v;
  }
#endif
}

}  // namespace

#if !V2_FEAT
size_t bf16v2_lds_bytes() { return LDS_BYTES; }
#endif

#ifdef PHASE_TIMING
#if V2_FEAT
extern "C" int objnerf_debug_phase_bf16v2f(unsigned long long* out_host) {
#else
extern "C" int objnerf_debug_phase_bf16v2(unsigned long long* out_host) {
#endif
  if (hipDeviceSynchronize() != hipSuccess) return OBJNERF_ELAUNCH;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase_b2), sizeof(unsigned long long) * 8 * 24) == hipSuccess
             ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
#endif

#if V2_FEAT
void launch_train_bf16_v2f(const TrainDev& d, void* stream) {
#else
void launch_train_bf16_v2(const TrainDev& d, void* stream) {
#endif
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)V2_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  hipLaunchKernelGGL(V2_KERNEL, dim3(d.K * d.G), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, d);
}

}  // namespace objtrain
