// objnerf_render_fwd, OBJNERF_TRAIN_BF16 mode: the lane-per-ray renderer of objnerf_render.hip with the network on
// v_mfma_f32_16x16x32_bf16 -- the forward half of the second-generation bf16 training kernel (objnerf_train_bf16v2.hip:
// packed operands, forward images of objnerf_bf16_common.h, hardware sin / cos / exp) under the same fp32 compositing
// (render_rays.py:6-63).  The fp32 renderer is bound by the fp32 contraction (0.61 of that MFMA peak); here the
// contraction is 16x cheaper and the kernel is bound by its VALU work.  Opt-in, like the training mode: not the
// reference's arithmetic (colour / depth differ at the 1e-2 / 1e-3 level), so it is held to the fp32 renderer with a
// tolerance and never selected by default.
#define OBJ_HW_SINCOS 1
#include "objnerf_bf16_common.h"
#include "objnerf_philox.h"
#include "../../include/objnerf_hip.h"

namespace {
using namespace objtrain;
using namespace objtrain::bf16k;

constexpr int B_FL = FWD_IMG_END;                         // feature layer image (as B_CL)
constexpr int B_SMALL = B_FL + 32 * RS_CL;
constexpr int LDS_BYTES = B_SMALL + SMALL_BYTES;

struct RenderDevB {
  long n; int n_bins, G;
  const float* params; const float* scale; const float* origin; const float* dirs_W; const float* near_; const float* far_;
  const float* u; uint64_t seed; uint32_t draw;
  float* depth; float* opacity; float* rgb; float* hfeat; float* z_out;
  Layout L;
};

template <bool FEAT>
__global__ __launch_bounds__(256) void render_fwd_bf16_kernel(const RenderDevB a) {
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < LDS_BYTES / 4; i += 256) reinterpret_cast<float*>(ldsb)[i] = 0.0f;
  __syncthreads();
  stage_forward_bf16(ldsb, reinterpret_cast<float*>(ldsb + B_SMALL), a.params, a.L, tid, FEAT, B_FL, 256);
  __syncthreads();
  const float* sm = reinterpret_cast<const float*>(ldsb + B_SMALL);
  const char* f_in = ldsb + B_IN + c * RS_IN + 16 * g;
  const char* f_m1 = ldsb + B_M1 + c * RS_M + 16 * g;
  const char* f_cat = ldsb + B_CAT + c * RS_CAT + 16 * g;
  const char* f_m2 = ldsb + B_M2 + c * RS_M + 16 * g;
  const char* f_cl = ldsb + B_CL + c * RS_CL + 16 * g;
  const char* f_fl = ldsb + B_FL + c * RS_CL + 16 * g;
  const float inv_scale = 1.0f / a.scale[0];
  const float ox = a.origin[0], oy = a.origin[1], oz = a.origin[2];
  const int S = a.n_bins - 1;
  const long ngroups = (a.n + 15) / 16;                      // 16 rays per wave step
  const uint32_t st = objrng::S_BOX_U | (a.draw << 3);
  for (long grp = (long)blockIdx.x * 4 + w; grp < ngroups; grp += (long)a.G * 4) {
    const long ray = grp * 16 + c;
    const bool valid = ray < a.n;
    const long r = valid ? ray : a.n - 1;
    const float lo = a.near_[r], hi = a.far_[r];
    const float dx = a.dirs_W[r * 3], dy = a.dirs_W[r * 3 + 1], dz = a.dirs_W[r * 3 + 2];
    const float* ur = a.u ? a.u + r * a.n_bins : nullptr;
    float ublk[4] = {0.f, 0.f, 0.f, 0.f};
    auto draw_u = [&](const int s) {
      if (ur) return ur[s];
      if ((s & 3) == 0) objrng::uniform4(a.seed, st, (uint32_t)(r >> 32), (uint32_t)r, (uint32_t)(s >> 2), ublk);
      return ublk[s & 3];
    };
    float z0 = strat(lo, hi, 0, a.n_bins, draw_u(0));
    float T = 1.0f;
    float aD = 0.f, aO = 0.f, aC = 0.f;
    T32 aF = zero32();
    for (int s = 0; s < S; ++s) {
      asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop
      const float z1 = strat(lo, hi, s + 1, a.n_bins, draw_u(s + 1));
      const float z = 0.5f * (z1 + z0);                       // trainer.py:175
      z0 = z1;
      if (a.z_out && valid && g == 0) a.z_out[r * S + s] = z;
      obj32n::Pe32 pe;
      pe_project_b(sm, g, ox + dx * z, oy + dy * z, oz + dz * z, inv_scale, pe);
      bf16x8 xb1[3], xb2[2];
      {
        f32x4 x2v[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          f32x4 xv[2];
#pragma unroll
          for (int u2 = 0; u2 < 2; ++u2) {
            const int i = 2 * b + u2;
            float sn[6], cs[6];
            obj32n::pe32_octaves<0, 5, false>(pe.vh[i], pe.vl[i], sn, cs);
            xv[u2] = obj32n::pe32_x1_tile(pe, i, g, sn);
            float v4, v5;
            obj32n::pe32_x2_pair(i, g, sn, v4, v5);
            x2v[b][2 * u2] = v4;
            x2v[b][2 * u2 + 1] = v5;
          }
          xb1[b] = pack8(xv[0], xv[1]);
        }
        xb2[0] = pack8(x2v[0], x2v[1]);
        xb2[1] = pack8(x2v[2], zero4());
      }
      T32 av = zero32();
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_IN>(av, f_in, b, xb1[b]);
      const bf16x8 h1p = pack32(relu32(av));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) av.t[tt][rr] = sm[S_BM1 + 16 * tt + 4 * g + rr];
      fwd_blk<RS_M>(av, f_m1, 0, h1p);
      const bf16x8 h2p = pack32(relu32(av));
      av = zero32();
      fwd_blk<RS_CAT>(av, f_cat, 0, h2p);
#pragma unroll
      for (int b = 0; b < 3; ++b) fwd_blk<RS_CAT>(av, f_cat, 1 + b, xb1[b]);
      const bf16x8 h3p = pack32(relu32(av));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) av.t[tt][rr] = sm[S_BM2 + 16 * tt + 4 * g + rr];
      fwd_blk<RS_M>(av, f_m2, 0, h3p);
      const T32 h4 = relu32(av);
      const bf16x8 h4p = pack32(h4);
      av = zero32();
      fwd_blk<RS_CL>(av, f_cl, 0, h4p);
      fwd_blk<RS_CL>(av, f_cl, 1, xb2[0]);
      fwd_blk<RS_CL>(av, f_cl, 2, xb2[1]);
      const T32 hc = relu32(av);
      T32 hf = zero32();
      if (FEAT) {
        av = zero32();
        fwd_blk<RS_CL>(av, f_fl, 0, h4p);
        fwd_blk<RS_CL>(av, f_fl, 1, xb2[0]);
        fwd_blk<RS_CL>(av, f_fl, 2, xb2[1]);
        hf = relu32(av);
      }
      float pa = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = 16 * tt + 4 * g + rr;
          pa = fmaf(sm[S_WA + row], h4.t[tt][rr], pa);
          pc0 = fmaf(sm[S_WOC + row], hc.t[tt][rr], pc0);
          pc1 = fmaf(sm[S_WOC + H + row], hc.t[tt][rr], pc1);
          pc2 = fmaf(sm[S_WOC + 2 * H + row], hc.t[tt][rr], pc2);
        }
      const float sa = xgroup_sum(pa), s0 = xgroup_sum(pc0), s1 = xgroup_sum(pc1), s2 = xgroup_sum(pc2);
      const float alpha10 = (sa + sm[S_HB]) * 10.0f;          // model.py:88, on every lane group
      const float mine = (g == 1) ? s0 : ((g == 2) ? s1 : s2);
      const float col = (g == 0) ? 0.0f : sigmoid_acc(mine + sm[S_HB + g]);
      const float occ = sigmoid_acc(alpha10);                 // render_rays.py:6-14
      const float wgt = occ * T;                              // render_rays.py:32-54
      T *= (1.0f - occ) + 1e-10f;
      aD = fmaf(wgt, z, aD);                                  // render_rays.py:56-63
      aO += wgt;
      aC = fmaf(wgt, col, aC);
      if (FEAT) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) aF.t[tt][rr] = fmaf(wgt, hf.t[tt][rr], aF.t[tt][rr]);
      }
    }
    if (valid) {
      if (g == 0) { a.depth[ray] = aD; a.opacity[ray] = aO; }
      else a.rgb[ray * 3 + g - 1] = aC;
      if (FEAT) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          *reinterpret_cast<float4*>(a.hfeat + ray * H + 16 * tt + 4 * g) =
              make_float4(aF.t[tt][0], aF.t[tt][1], aF.t[tt][2], aF.t[tt][3]);
      }
    }
  }
}

}  // namespace

int objnerf_render_fwd_bf16(const objnerf_net* net, int64_t n, int32_t n_bins, const float* params, const float* scale,
                            const float* origin, const float* dirs_W, const float* near, const float* far, const float* u,
                            uint64_t seed, uint32_t draw, float* out_depth, float* out_opacity, float* out_rgb,
                            float* out_hfeat, float* out_z, void* stream) {
  RenderDevB d;
  d.n = n; d.n_bins = n_bins;
  d.params = params; d.scale = scale; d.origin = origin; d.dirs_W = dirs_W; d.near_ = near; d.far_ = far;
  d.u = u; d.seed = seed; d.draw = draw;
  d.depth = out_depth; d.opacity = out_opacity; d.rgb = out_rgb; d.hfeat = out_hfeat; d.z_out = out_z;
  d.L = make_layout(net->feat_dim);
  int dev = 0, cu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
  const long ngroups = (n + 15) / 16;
  long G = 2L * cu;                                           // two workgroups of 4 waves per CU
  if (G * 4 > ngroups) G = (ngroups + 3) / 4;
  d.G = (int)G;
  const bool feat = out_hfeat != nullptr;
  if (feat) hipLaunchKernelGGL(render_fwd_bf16_kernel<true>, dim3((unsigned)G), dim3(256), LDS_BYTES, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(render_fwd_bf16_kernel<false>, dim3((unsigned)G), dim3(256), LDS_BYTES, (hipStream_t)stream, d);
  return hipGetLastError() == hipSuccess ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
