// Small kernels behind the reference's helper functions that a caller of the mirrored Python API may use outside the
// fused iteration: render_rays.render_loss / reduce_batch_loss / make_3D_grid (render_rays.py:65-146) and
// utils.ray_box_intersection / origin_dirs_W / stratified_bins / normal_bins_sampling (utils.py:309-397).
// All HBM-bound elementwise / per-row work: one thread per element or per row, coalesced; nothing here is on the
// training path (the fused kernels carry their own loss and sampler).
#include <hip/hip_runtime.h>
#include <math.h>
#include "objnerf_device.h"
#include "objnerf_philox.h"
#include "../../include/objnerf_hip.h"

namespace {

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return OBJNERF_ELAUNCH; } while (0)

// residuals of render_rays.py:65-83.  mode 0: |a - b| (L1), 1: (a - b)^2 (L2): n elements; 2: 1 - cos(a, b) over rows
// of C entries (F.cosine_similarity: each norm clamped at 1e-8): n rows.
__global__ void render_loss_kernel(long n, int C, int mode, int normalise, const float* a, const float* b, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 2) {
    const float* pa = a + i * C;
    const float* pb = b + i * C;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = 0; c < C; ++c) { ab = fmaf(pa[c], pb[c], ab); aa = fmaf(pa[c], pa[c], aa); bb = fmaf(pb[c], pb[c], bb); }
    out[i] = 1.0f - ab / (fmaxf(sqrtf(aa), 1e-8f) * fmaxf(sqrtf(bb), 1e-8f));
    return;
  }
  const float r = a[i] - b[i];
  float v = mode == 1 ? r * r : fabsf(r);
  if (normalise) v = v / b[i];
  out[i] = v;
}

// reduce_batch_loss, render_rays.py:85-117.  Pass 1: per-object mask counts and the cross-object "some mask is empty"
// flag (one workgroup per object).  Pass 2: information-weighted masked mean per object, zero everywhere when the flag
// is set (the early return of :89-94), status bit when a mean exceeds 1e5 (:109-111).
__global__ __launch_bounds__(256) void mask_count_kernel(int R, const uint8_t* mask, int* counts, int* any_empty) {
  __shared__ int red[256];
  const int k = blockIdx.x;
  int c = 0;
  for (int r = threadIdx.x; r < R; r += 256) c += mask[(long)k * R + r] ? 1 : 0;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) { counts[k] = red[0]; if (red[0] == 0) atomicOr(any_empty, 1); }
}
__global__ __launch_bounds__(256) void reduce_loss_kernel(int R, const float* loss_mat, const float* var, int l2, int avg,
                                                          const int* counts, const int* any_empty, float* out, int* status) {
  __shared__ float red[256];
  const int k = blockIdx.x;
  const bool zero = *any_empty != 0;
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const long i = (long)k * R + r;
    float v = loss_mat[i];
    if (var) v *= l2 ? 1.0f / (var[i] + 1e-4f) : 1.0f / (sqrtf(var[i]) + 1e-4f);
    if (zero) v = 0.0f;
    if (!avg) out[i] = v;
    s += v;
  }
  if (!avg) return;
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float m = zero ? 0.0f : red[0] / ((float)counts[k] + 1e-10f);
    out[k] = m;
    if (m > 100000.0f) atomicOr(status, 1);
    if (!(fabsf(m) <= 3.0e38f)) atomicOr(status, 2);   // NaN / Inf: reported, not fatal
  }
}

__device__ __forceinline__ float linspace_at(float lo, float hi, int i, int n) {   // torch.linspace(lo, hi, n)[i] (GPU formula)
  const float step = (hi - lo) / (float)(n - 1);
  return (i < n / 2) ? lo + step * (float)i : hi - step * (float)(n - 1 - i);
}
// make_3D_grid, render_rays.py:119-146: the dim^3 lattice on [lo, hi]^3, optionally scaled per axis and moved by T [4,4]
__global__ void grid_kernel(int dim, float lo, float hi, const float* scale, const float* T, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)dim * dim * dim;
  if (i >= n) return;
  const int iz = (int)(i % dim), iy = (int)((i / dim) % dim), ix = (int)(i / ((long)dim * dim));
  float g[3] = {linspace_at(lo, hi, ix, dim), linspace_at(lo, hi, iy, dim), linspace_at(lo, hi, iz, dim)};
  if (scale) { g[0] *= scale[0]; g[1] *= scale[1]; g[2] *= scale[2]; }
  if (T) {
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = ((T[4 * r] * g[0] + T[4 * r + 1] * g[1]) + T[4 * r + 2] * g[2]) + T[4 * r + 3];
    g[0] = o[0]; g[1] = o[1]; g[2] = o[2];
  }
  out[i * 3] = g[0]; out[i * 3 + 1] = g[1]; out[i * 3 + 2] = g[2];
}

// ray_box_intersection, utils.py:309-319 (slab test, axis-aligned bounds)
__global__ void ray_box_kernel(long n, const float* o, const float* d, const float* bmin, const float* bmax, float* near_o,
                               float* far_o, uint8_t* hit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float nr = -INFINITY, fr = INFINITY;
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    const float t0 = (bmin[x] - o[i * 3 + x]) / d[i * 3 + x];
    const float t1 = (bmax[x] - o[i * 3 + x]) / d[i * 3 + x];
    nr = fmaxf(nr, fminf(t0, t1));
    fr = fminf(fr, fmaxf(t0, t1));
  }
  near_o[i] = nr; far_o[i] = fr;
  hit[i] = (nr <= fr) && (fr > 0.0f) ? 1 : 0;
}

// origin_dirs_W, utils.py:324-336: dirs_W[f][p] = R_wc[f] dirs_C[f][p]
__global__ void dirs_w_kernel(long F, long P, const float* T_WC, const float* dirs_C, float* dirs_W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * P) return;
  const float* T = T_WC + (i / P) * 16;
  const float d0 = dirs_C[i * 3], d1 = dirs_C[i * 3 + 1], d2 = dirs_C[i * 3 + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) dirs_W[i * 3 + r] = (T[4 * r] * d0 + T[4 * r + 1] * d1) + T[4 * r + 2] * d2;
}

// stratified_bins, utils.py:342-379: lo + range * i / n + U(0, 1) * range / n.  lo / hi: per ray, or NULL = the scalar
__global__ void strat_bins_kernel(long n_rays, int n_bins, const float* lo_p, float lo_s, const float* hi_p, float hi_s,
                                  const float* u, unsigned long long seed, unsigned long long offset, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * n_bins) return;
  const long r = i / n_bins;
  const int s = (int)(i - r * n_bins);
  const float lo = lo_p ? lo_p[r] : lo_s, hi = hi_p ? hi_p[r] : hi_s;
  const float uu = u ? u[i] : objrng::uniform1(seed, objrng::S_HELPER_U, (uint32_t)offset, (uint32_t)r, (uint32_t)s);
  const float rng = hi - lo;
  out[i] = (rng * lin01(s, n_bins) + lo) + uu * (rng / (float)n_bins);
}
// normal_bins_sampling, utils.py:382-397: depth + clip(sort(N(0, (delta / 3)^2)), -delta, delta); one thread per ray
__global__ void normal_bins_kernel(long n_rays, int n_bins, const float* depth, float delta, const float* g,
                                   unsigned long long seed, unsigned long long offset, float* out) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float sd = delta / 3.0f;
  float* z = out + r * n_bins;
  for (int s = 0; s < n_bins; ++s)
    z[s] = g ? g[r * n_bins + s] : sd * objrng::normal1(seed, objrng::S_HELPER_G, (uint32_t)offset, (uint32_t)r, (uint32_t)s);
  for (int s = 1; s < n_bins; ++s) {              // insertion sort in place (n_bins <= ~100)
    const float v = z[s];
    int j = s - 1;
    while (j >= 0 && z[j] > v) { z[j + 1] = z[j]; --j; }
    z[j + 1] = v;
  }
  const float d = depth[r];
  for (int s = 0; s < n_bins; ++s) z[s] = d + fminf(fmaxf(z[s], -delta), delta);
}

}  // namespace

// render_rays.render (render_rays.py:56-63): out[ray][c] = sum_s termination[ray][s] * vals[ray][s][c]; one thread per
// output, the samples of a ray are read in order (the reference's torch.sum order is unspecified)
__global__ void render_kernel(long n_out, int S, int C, const float* term, const float* vals, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const long ray = i / C;
  const int ch = (int)(i - ray * C);
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc = fmaf(term[ray * S + s], vals[(ray * S + s) * C + ch], acc);
  out[i] = acc;
}

extern "C" {

int objnerf_render(int64_t n_rays, int32_t S, int32_t C, const float* termination, const float* vals, float* out,
                   void* stream) {
  (void)hipGetLastError();
  if (n_rays <= 0 || S <= 0 || C <= 0 || !termination || !vals || !out) return OBJNERF_EINVAL;
  const long n_out = (long)n_rays * C;
  hipLaunchKernelGGL(render_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_out, S, C,
                     termination, vals, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_render_loss(int64_t n, int32_t C, int32_t mode, int32_t normalise, const float* render, const float* gt,
                        float* out, void* stream) {
  (void)hipGetLastError();
  if (n <= 0 || !render || !gt || !out || mode < 0 || mode > 2 || (mode == 2 && C <= 0)) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(render_loss_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, C,
                     mode, normalise, render, gt, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_reduce_batch_loss(int32_t K, int32_t R, const float* loss_mat, const float* var, const uint8_t* mask,
                              int32_t l2, int32_t avg, int32_t* counts_ws, float* out, int32_t* status, void* stream) {
  (void)hipGetLastError();
  if (K <= 0 || R <= 0 || !loss_mat || !mask || !counts_ws || !out || !status) return OBJNERF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(counts_ws + K, 0, sizeof(int), st);       // counts_ws[K] = the any-empty flag
  (void)hipMemsetAsync(status, 0, sizeof(int), st);
  hipLaunchKernelGGL(mask_count_kernel, dim3(K), dim3(256), 0, st, R, mask, counts_ws, counts_ws + K);
  hipLaunchKernelGGL(reduce_loss_kernel, dim3(K), dim3(256), 0, st, R, loss_mat, var, l2, avg, counts_ws, counts_ws + K, out,
                     status);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_make_grid(int32_t dim, float lo, float hi, const float* scale, const float* transform, float* out,
                      void* stream) {
  (void)hipGetLastError();
  if (dim < 2 || !out) return OBJNERF_EINVAL;
  const long n = (long)dim * dim * dim;
  hipLaunchKernelGGL(grid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dim, lo, hi, scale,
                     transform, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_ray_box(int64_t n, const float* origins, const float* dirs, const float* bmin, const float* bmax,
                    float* out_near, float* out_far, uint8_t* out_hit, void* stream) {
  (void)hipGetLastError();
  if (n <= 0 || !origins || !dirs || !bmin || !bmax || !out_near || !out_far || !out_hit) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(ray_box_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, origins,
                     dirs, bmin, bmax, out_near, out_far, out_hit);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_dirs_w(int64_t F, int64_t P, const float* T_WC, const float* dirs_C, float* out_dirs_W, void* stream) {
  (void)hipGetLastError();
  if (F <= 0 || P <= 0 || !T_WC || !dirs_C || !out_dirs_W) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(dirs_w_kernel, dim3((unsigned)((F * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)F, (long)P,
                     T_WC, dirs_C, out_dirs_W);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_stratified_bins(int64_t n_rays, int32_t n_bins, const float* lo, float lo_scalar, const float* hi,
                            float hi_scalar, const float* u, uint64_t seed, uint64_t offset, float* out, void* stream) {
  (void)hipGetLastError();
  if (n_rays <= 0 || n_bins <= 0 || !out) return OBJNERF_EINVAL;
  const long n = (long)n_rays * n_bins;
  hipLaunchKernelGGL(strat_bins_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n_rays,
                     n_bins, lo, lo_scalar, hi, hi_scalar, u, (unsigned long long)seed, (unsigned long long)offset, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

int objnerf_normal_bins(int64_t n_rays, int32_t n_bins, const float* depth, float delta, const float* g, uint64_t seed,
                        uint64_t offset, float* out, void* stream) {
  (void)hipGetLastError();
  if (n_rays <= 0 || n_bins <= 0 || !depth || !out) return OBJNERF_EINVAL;
  hipLaunchKernelGGL(normal_bins_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (long)n_rays, n_bins, depth, delta, g, (unsigned long long)seed, (unsigned long long)offset, out);
  CHECK_LAUNCH();
  return OBJNERF_OK;
}

}  // extern "C"
