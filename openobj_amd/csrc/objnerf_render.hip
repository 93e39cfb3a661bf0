// objnerf_render_fwd: novel-view rendering of ONE object inside its box in ONE launch (row f-1 of SURVEY.md 8;
// reference sceneObject.render_2D_syn vmap.py:604-685 on Trainer.sample_points_bbox trainer.py:130-198, the stacked
// modules embedding.py:46-55 + model.py:61-103 and render_rays.py:6-63).
//
// The reference (and rounds 1-3 here) materialise the sample points, then alpha / colour / the 512-d feature of every
// sample, then composite.  Per ray of 149 samples that is 21 KB of per-sample tensors written and read back.  Here a
// lane owns a RAY: a wave steps through the 149 mid-points of 16 rays at once (lane c <-> ray, lane group g <-> the
// feature rows 4 g + r of the register-resident MFMA chain, objnerf_mlp32.h), so the transmittance is a running product
// in a register -- no scan, no shuffle -- and depth / opacity / colour / the 32-wide feature hidden accumulate per lane.
// Nothing per sample ever leaves the registers; a ray costs 20 B of input and 16 + 128 B of output.  The 512-d head is
// applied afterwards to the composited hidden of the rays that pass the masks (exact: the head is linear).
#include "objnerf_mlp32.h"
#include "objnerf_philox.h"
#include "../../include/objnerf_hip.h"

namespace {
using namespace obj32n;

struct RenderDev {
  long n; int n_bins, G;
  const float* params; const float* scale; const float* origin; const float* dirs_W; const float* near_; const float* far_;
  const float* u; uint64_t seed; uint32_t draw;
  float* depth; float* opacity; float* rgb; float* hfeat; float* z_out;
  Layout L;
};

template <bool FEAT>
__global__ __launch_bounds__(256) void render_fwd_kernel(const RenderDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  stage_weights32(lds, a.params, a.L, FEAT, tid, 256);
  const float* sv = lds + sv_base(FEAT);
  const float* wf = (const float*)__builtin_assume_aligned(lds + 4 * g * WROW + out_pos(c), 8);
  const float scale = a.scale[0];
  const float ox = a.origin[0], oy = a.origin[1], oz = a.origin[2];
  const int S = a.n_bins - 1;
  const long ngroups = (a.n + 15) / 16;                      // 16 rays per wave step
  const uint32_t st = objrng::S_BOX_U | (a.draw << 3);
  for (long grp = (long)blockIdx.x * 4 + w; grp < ngroups; grp += (long)a.G * 4) {
    asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (no LICM into registers)
    const long ray = grp * 16 + c;
    const bool valid = ray < a.n;
    const long r = valid ? ray : a.n - 1;
    const float lo = a.near_[r], hi = a.far_[r];
    const float dx = a.dirs_W[r * 3], dy = a.dirs_W[r * 3 + 1], dz = a.dirs_W[r * 3 + 2];
    const float* ur = a.u ? a.u + r * a.n_bins : nullptr;
    float ublk[4] = {0.f, 0.f, 0.f, 0.f};
    auto draw_u = [&](const int s) {                          // the draw of bin s (injected, or Philox block s >> 2)
      if (ur) return ur[s];
      if ((s & 3) == 0 || s == 0) objrng::uniform4(a.seed, st, (uint32_t)(r >> 32), (uint32_t)r, (uint32_t)(s >> 2), ublk);
      return ublk[s & 3];
    };
    float z0 = strat(lo, hi, 0, a.n_bins, draw_u(0));
    float T = 1.0f;                                           // transmittance in front of the current sample
    float aD = 0.f, aO = 0.f, aC = 0.f;
    T32 aF = zero32();
    for (int s = 0; s < S; ++s) {
      const float z1 = strat(lo, hi, s + 1, a.n_bins, draw_u(s + 1));
      const float z = 0.5f * (z1 + z0);                       // trainer.py:175
      z0 = z1;
      if (a.z_out && valid && g == 0) a.z_out[r * S + s] = z;
      Pe32 pe;
      pe32_project(sv, g, ox + dx * z, oy + dy * z, oz + dz * z, scale, pe);     // trainer.py:176, embedding.py:47-48
      Emb32 e;
      embed32(e, pe, g);
      Acts act;
      float alpha10, col;
      mlp32_forward_all<FEAT>(wf, sv, g, e, act, alpha10, col);
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
          for (int rr = 0; rr < 4; ++rr) aF.t[tt][rr] = fmaf(wgt, act.hf.t[tt][rr], aF.t[tt][rr]);
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

// objnerf_render_bf16.hip
int objnerf_render_fwd_bf16(const objnerf_net* net, int64_t n, int32_t n_bins, const float* params, const float* scale,
                            const float* origin, const float* dirs_W, const float* near, const float* far, const float* u,
                            uint64_t seed, uint32_t draw, float* out_depth, float* out_opacity, float* out_rgb,
                            float* out_hfeat, float* out_z, void* stream);

extern "C" int objnerf_render_fwd(const objnerf_net* net, int64_t n, int32_t n_bins, const float* params, const float* scale,
                                  const float* origin, const float* dirs_W, const float* near, const float* far,
                                  const float* u, uint64_t seed, uint32_t draw, float* out_depth, float* out_opacity,
                                  float* out_rgb, float* out_hfeat, float* out_z, int32_t mode, void* stream) {
  (void)hipGetLastError();
  if (!net || n <= 0 || n_bins < 2 || !params || !scale || !origin || !dirs_W || !near || !far || !out_depth ||
      !out_opacity || !out_rgb)
    return OBJNERF_EINVAL;
  if (net->hidden != 32 || net->n_freqs != 6) return OBJNERF_ENOTSUP;      // (wider networks: the layer-wise chain)
  if (mode & ~OBJNERF_TRAIN_BF16) return OBJNERF_ENOTSUP;
  if (mode & OBJNERF_TRAIN_BF16)
    return objnerf_render_fwd_bf16(net, n, n_bins, params, scale, origin, dirs_W, near, far, u, seed, draw, out_depth,
                                   out_opacity, out_rgb, out_hfeat, out_z, stream);
  RenderDev d;
  d.n = n; d.n_bins = n_bins;
  d.params = params; d.scale = scale; d.origin = origin; d.dirs_W = dirs_W; d.near_ = near; d.far_ = far;
  d.u = u; d.seed = seed; d.draw = draw;
  d.depth = out_depth; d.opacity = out_opacity; d.rgb = out_rgb; d.hfeat = out_hfeat; d.z_out = out_z;
  d.L = make_layout(net->feat_dim);
  int dev = 0, cu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
  const long ngroups = (n + 15) / 16;
  const bool feat = out_hfeat != nullptr;
  const size_t lds_bytes = (size_t)img_floats(feat) * 4;
  long G = 2L * cu;                                           // two workgroups of 4 waves per CU: 2 waves per SIMD
  if (G * 4 > ngroups) G = (ngroups + 3) / 4;
  d.G = (int)G;
  objnerf_once_per_device([] {
    (void)hipFuncSetAttribute((const void*)render_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, img_floats(true) * 4);
    (void)hipFuncSetAttribute((const void*)render_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, img_floats(false) * 4);
  });
  if (feat) hipLaunchKernelGGL(render_fwd_kernel<true>, dim3((unsigned)G), dim3(256), lds_bytes, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(render_fwd_kernel<false>, dim3((unsigned)G), dim3(256), lds_bytes, (hipStream_t)stream, d);
  return hipGetLastError() == hipSuccess ? OBJNERF_OK : OBJNERF_ELAUNCH;
}
