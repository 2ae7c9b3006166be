// hg_augment.hip -- DiffAugment (include/hg_augment.h): one HBM pass for a run of spatial augmentations
// (flip -> roll -> zero-filled shift -> cutout), one for the colour augmentations, parameterised per sample.
// All maps are linear; the spatial ones are injective gathers, so the adjoint is a gather through the inverse map.
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_augment.h"

namespace {

__device__ __forceinline__ int wrap(int v, int n) {   // v mod n for v in (-n, 2n)
  v = v < 0 ? v + n : v;
  return v >= n ? v - n : v;
}

// grid (ceil(W/64), ceil(H/4), B*C)
template <bool ADJ>
__global__ __launch_bounds__(256) void k_aug_spatial(const float *__restrict__ x, const int32_t *__restrict__ params,
                                                     float *__restrict__ out, int C, int H, int W) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= H || j >= W) return;
  const int bc = blockIdx.z, b = bc / C;
  const int32_t *p = params + (size_t)b * HG_AUG_NP;
  const int flip = p[0], rh = p[1], rw = p[2], sh = p[3], sw = p[4], r0 = p[5], r1 = p[6], c0 = p[7], c1 = p[8];
  const float *xp = x + (size_t)bc * H * W;
  float v = 0.f;
  if constexpr (!ADJ) {
    // out[i,j] = mask[i,j] * flip(roll(.))[i + sh, j + sw]
    const bool cut = i >= r0 && i <= r1 && j >= c0 && j <= c1;
    const int ti = i + sh, tj = j + sw;
    if (!cut && ti >= 0 && ti < H && tj >= 0 && tj < W) {
      const int si = wrap(ti - rh, H);
      int sj = wrap(tj - rw, W);
      if (flip) sj = W - 1 - sj;
      v = xp[(size_t)si * W + sj];
    }
  } else {
    // source pixel (i, j): the unique output pixel reading it, if that pixel exists and is not cut out
    const int fj = flip ? W - 1 - j : j;
    const int oi = wrap(i + rh, H) - sh, oj = wrap(fj + rw, W) - sw;
    if (oi >= 0 && oi < H && oj >= 0 && oj < W && !(oi >= r0 && oi <= r1 && oj >= c0 && oj <= c1))
      v = xp[(size_t)oi * W + oj];
  }
  out[(size_t)bc * H * W + (size_t)i * W + j] = v;
}

// per-sample sum over CHW: grid (chunks, B) partials, then a finishing launch
__global__ __launch_bounds__(256) void k_sample_sum(const float *__restrict__ x, float *__restrict__ part, long long n) {
  __shared__ float sm[4];
  const float *xp = x + (size_t)blockIdx.y * n;
  float s = 0.f;
  if ((n & 3) == 0) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n / 4; e += (long long)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4 *>(xp)[e];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) s += xp[e];
  }
  s = hg_block_sum_256(s, sm);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_sample_mean_finish(const float *__restrict__ part, float *__restrict__ mean,
                                                           int B, int chunks, float inv_n) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)b * chunks + c];
  mean[b] = s * inv_n;
}

// grid (ceil(HW/256), B); one thread per pixel, all channels
template <bool ADJ>
__global__ __launch_bounds__(256) void k_aug_color(const float *__restrict__ x, const float *__restrict__ mean,
                                                   const float *__restrict__ color, float *__restrict__ out, int C,
                                                   int HW) {
  const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (p >= HW) return;
  const float br = color[3 * b], sat = color[3 * b + 1], con = color[3 * b + 2];
  const float *xp = x + (size_t)b * C * HW + p;
  float *op = out + (size_t)b * C * HW + p;
  const float invc = 1.f / (float)C;
  if constexpr (!ADJ) {
    const float m = mean[b] + br;
    float mc = 0.f;
    for (int c = 0; c < C; ++c) mc += xp[(size_t)c * HW] + br;
    mc *= invc;
    for (int c = 0; c < C; ++c) {
      const float x2 = (xp[(size_t)c * HW] + br - mc) * sat + mc;
      op[(size_t)c * HW] = (x2 - m) * con + m;
    }
  } else {
    const float gm = (1.f - con) * mean[b];
    float mc = 0.f;
    for (int c = 0; c < C; ++c) mc += con * xp[(size_t)c * HW] + gm;
    mc *= invc;
    for (int c = 0; c < C; ++c) op[(size_t)c * HW] = sat * (con * xp[(size_t)c * HW] + gm) + (1.f - sat) * mc;
  }
}

constexpr int MEAN_CHUNKS = 32;

}  // namespace

extern "C" {

int hg_augment_spatial(const float *x, const int32_t *params, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                       int32_t adjoint, void *stream) {
  if (!x || !params || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)B * C > 65535) return HG_EINVAL;
  const dim3 grid((W + 63) / 64, (H + 3) / 4, B * C);
  if (adjoint) hipLaunchKernelGGL(k_aug_spatial<true>, grid, dim3(256), 0, (hipStream_t)stream, x, params, out, C, H, W);
  else hipLaunchKernelGGL(k_aug_spatial<false>, grid, dim3(256), 0, (hipStream_t)stream, x, params, out, C, H, W);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

size_t hg_augment_workspace_bytes(int32_t B) { return B > 0 ? (size_t)B * MEAN_CHUNKS * sizeof(float) : 0; }

int hg_sample_mean(const float *x, float *mean, int32_t B, int64_t CHW, void *workspace, size_t workspace_bytes,
                   void *stream) {
  if (!x || !mean || B <= 0 || B > 65535 || CHW <= 0) return HG_EINVAL;
  if (!workspace || workspace_bytes < hg_augment_workspace_bytes(B)) return HG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int chunks = (int)((CHW / 4 + 1023) / 1024);
  chunks = chunks < 1 ? 1 : (chunks > MEAN_CHUNKS ? MEAN_CHUNKS : chunks);
  hipLaunchKernelGGL(k_sample_sum, dim3(chunks, B), dim3(256), 0, st, x, (float *)workspace, (long long)CHW);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sample_mean_finish, dim3((B + 63) / 64), dim3(64), 0, st, (const float *)workspace, mean, B,
                     chunks, 1.f / (float)CHW);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_augment_color(const float *x, const float *mean, const float *color, float *out, int32_t B, int32_t C,
                     int32_t HW, int32_t adjoint, void *stream) {
  if (!x || !mean || !color || !out || B <= 0 || B > 65535 || C <= 0 || HW <= 0) return HG_EINVAL;
  const dim3 grid((HW + 255) / 256, B);
  if (adjoint) hipLaunchKernelGGL(k_aug_color<true>, grid, dim3(256), 0, (hipStream_t)stream, x, mean, color, out, C, HW);
  else hipLaunchKernelGGL(k_aug_color<false>, grid, dim3(256), 0, (hipStream_t)stream, x, mean, color, out, C, HW);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
