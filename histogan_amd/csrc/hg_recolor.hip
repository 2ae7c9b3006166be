// hg_recolor.hip -- the HBM-bound kernels ReHistoGAN adds to the path (include/hg_recolor.h):
//   instance norm + LeakyReLU (encoder blocks), the 3-channel 3x3 stencil of the Sobel / Laplacian reconstruction
//   loss, and the 15x15 depthwise Gaussian of the variance loss -- each with its adjoint.
// One read + one write per element (instance norm: the statistics pass re-reads its chunk from L2).
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_recolor.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float *sm4) { return hg_block_sum_256(v, sm4); }

// element range of chunk j of a plane of HW elements split into `chunks` pieces (multiples of 4 when HW is)
__device__ __forceinline__ void chunk_range(int HW, int chunks, int j, int &lo, int &hi) {
  int len = (HW + chunks - 1) / chunks;
  len = (len + 3) & ~3;
  lo = min(j * len, HW);
  hi = min(lo + len, HW);
}

// ---- instance norm + leaky relu ----------------------------------------------------------------
// grid (P, chunks): per-chunk mean and M2 = sum (x - mean_chunk)^2, two passes over the chunk (second from L2)
__global__ __launch_bounds__(256) void k_in_stats(const float *__restrict__ x, float *__restrict__ part, int HW,
                                                  int chunks) {
  __shared__ float sm[4];
  int lo, hi;
  chunk_range(HW, chunks, blockIdx.y, lo, hi);
  const float *xp = x + (size_t)blockIdx.x * HW;
  const bool vec = (HW & 3) == 0;
  float s = 0.f;
  if (vec) {
    for (int e = lo / 4 + threadIdx.x; e < hi / 4; e += 256) {
      const float4 v = reinterpret_cast<const float4 *>(xp)[e];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) s += xp[e];
  }
  s = block_sum256(s, sm);
  const float mean = hi > lo ? s / (float)(hi - lo) : 0.f;
  float q = 0.f;
  if (vec) {
    for (int e = lo / 4 + threadIdx.x; e < hi / 4; e += 256) {
      const float4 v = reinterpret_cast<const float4 *>(xp)[e];
      const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) { const float a = xp[e] - mean; q += a * a; }
  }
  q = block_sum256(q, sm);
  if (threadIdx.x == 0) {
    float *o = part + ((size_t)blockIdx.x * chunks + blockIdx.y) * 2;
    o[0] = mean; o[1] = q;
  }
}

// combine the per-chunk (mean, M2) pairs of one plane in chunk order (Chan et al.)
__device__ __forceinline__ void in_combine(const float *__restrict__ part, int HW, int chunks, float eps, float &mean,
                                           float &rstd) {
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int j = 0; j < chunks; ++j) {
    int lo, hi;
    chunk_range(HW, chunks, j, lo, hi);
    const float nj = (float)(hi - lo);
    if (nj == 0.f) continue;
    const float d = part[2 * j] - mu, tot = n + nj;
    mu += d * (nj / tot);
    m2 += part[2 * j + 1] + d * d * (n * nj / tot);
    n = tot;
  }
  mean = mu;
  rstd = 1.f / sqrtf(m2 / (float)HW + eps);
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

__global__ __launch_bounds__(256) void k_in_apply(const float *__restrict__ x, const float *__restrict__ part,
                                                  float *__restrict__ out, float *__restrict__ stats, int HW,
                                                  int chunks, float eps, float slope) {
  float mean, rstd;
  in_combine(part + (size_t)blockIdx.x * chunks * 2, HW, chunks, eps, mean, rstd);
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    stats[2 * (size_t)blockIdx.x] = mean;
    stats[2 * (size_t)blockIdx.x + 1] = rstd;
  }
  int lo, hi;
  chunk_range(HW, chunks, blockIdx.y, lo, hi);
  const float *xp = x + (size_t)blockIdx.x * HW;
  float *op = out + (size_t)blockIdx.x * HW;
  if ((HW & 3) == 0) {
    for (int e = lo / 4 + threadIdx.x; e < hi / 4; e += 256) {
      float4 v = reinterpret_cast<const float4 *>(xp)[e];
      v.x = lrelu((v.x - mean) * rstd, slope); v.y = lrelu((v.y - mean) * rstd, slope);
      v.z = lrelu((v.z - mean) * rstd, slope); v.w = lrelu((v.w - mean) * rstd, slope);
      reinterpret_cast<float4 *>(op)[e] = v;
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) op[e] = lrelu((xp[e] - mean) * rstd, slope);
  }
}

// m = gout * lrelu'(out), xhat = out / lrelu'(out)
__device__ __forceinline__ void in_mx(float go, float o, float slope, float islope, float &m, float &xh) {
  const bool pos = o > 0.f;
  m = pos ? go : go * slope;
  xh = pos ? o : o * islope;
}

__global__ __launch_bounds__(256) void k_in_bwd_stats(const float *__restrict__ gout, const float *__restrict__ out,
                                                      float *__restrict__ part, int HW, int chunks, float slope) {
  __shared__ float sm[4];
  int lo, hi;
  chunk_range(HW, chunks, blockIdx.y, lo, hi);
  const float *gp = gout + (size_t)blockIdx.x * HW, *op = out + (size_t)blockIdx.x * HW;
  const float islope = 1.f / slope;
  float s1 = 0.f, s2 = 0.f, m, xh;
  if ((HW & 3) == 0) {
    for (int e = lo / 4 + threadIdx.x; e < hi / 4; e += 256) {
      const float4 g = reinterpret_cast<const float4 *>(gp)[e], o = reinterpret_cast<const float4 *>(op)[e];
      in_mx(g.x, o.x, slope, islope, m, xh); s1 += m; s2 += m * xh;
      in_mx(g.y, o.y, slope, islope, m, xh); s1 += m; s2 += m * xh;
      in_mx(g.z, o.z, slope, islope, m, xh); s1 += m; s2 += m * xh;
      in_mx(g.w, o.w, slope, islope, m, xh); s1 += m; s2 += m * xh;
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) { in_mx(gp[e], op[e], slope, islope, m, xh); s1 += m; s2 += m * xh; }
  }
  s1 = block_sum256(s1, sm);
  s2 = block_sum256(s2, sm);
  if (threadIdx.x == 0) {
    float *o = part + ((size_t)blockIdx.x * chunks + blockIdx.y) * 2;
    o[0] = s1; o[1] = s2;
  }
}

__global__ __launch_bounds__(256) void k_in_bwd_apply(const float *__restrict__ gout, const float *__restrict__ out,
                                                      const float *__restrict__ stats, const float *__restrict__ part,
                                                      float *__restrict__ gx, int HW, int chunks, float slope) {
  float s1 = 0.f, s2 = 0.f;
  const float *pp = part + (size_t)blockIdx.x * chunks * 2;
  for (int j = 0; j < chunks; ++j) { s1 += pp[2 * j]; s2 += pp[2 * j + 1]; }
  const float rstd = stats[2 * (size_t)blockIdx.x + 1], inv = 1.f / (float)HW, islope = 1.f / slope;
  const float c1 = s1 * inv, c2 = s2 * inv;
  int lo, hi;
  chunk_range(HW, chunks, blockIdx.y, lo, hi);
  const float *gp = gout + (size_t)blockIdx.x * HW, *op = out + (size_t)blockIdx.x * HW;
  float *xp = gx + (size_t)blockIdx.x * HW;
  float m, xh;
  if ((HW & 3) == 0) {
    for (int e = lo / 4 + threadIdx.x; e < hi / 4; e += 256) {
      const float4 g = reinterpret_cast<const float4 *>(gp)[e], o = reinterpret_cast<const float4 *>(op)[e];
      float4 r;
      in_mx(g.x, o.x, slope, islope, m, xh); r.x = rstd * (m - c1 - xh * c2);
      in_mx(g.y, o.y, slope, islope, m, xh); r.y = rstd * (m - c1 - xh * c2);
      in_mx(g.z, o.z, slope, islope, m, xh); r.z = rstd * (m - c1 - xh * c2);
      in_mx(g.w, o.w, slope, islope, m, xh); r.w = rstd * (m - c1 - xh * c2);
      reinterpret_cast<float4 *>(xp)[e] = r;
    }
  } else {
    for (int e = lo + threadIdx.x; e < hi; e += 256) {
      in_mx(gp[e], op[e], slope, islope, m, xh);
      xp[e] = rstd * (m - c1 - xh * c2);
    }
  }
}

// ---- 3x3 stencil, C channels -> 1 (and its adjoint) -----------------------------------------------
struct Taps9 { float t[9]; };

template <bool ADJ>
__global__ __launch_bounds__(256) void k_stencil3(const float *__restrict__ x, float *__restrict__ out, Taps9 tp, int C,
                                                  int H, int W) {
  const int xx = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (xx >= W || yy >= H) return;
  const size_t hw = (size_t)H * W;
  if constexpr (!ADJ) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float *p = x + ((size_t)b * C + c) * hw;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int y = yy + i - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int xq = xx + j - 1;
          if (xq >= 0 && xq < W) acc = fmaf(tp.t[3 * i + j], p[(size_t)y * W + xq], acc);
        }
      }
    }
    out[(size_t)b * hw + (size_t)yy * W + xx] = acc;
  } else {
    const float *p = x + (size_t)b * hw;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int y = yy - i + 1;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int xq = xx - j + 1;
        if (xq >= 0 && xq < W) acc = fmaf(tp.t[3 * i + j], p[(size_t)y * W + xq], acc);
      }
    }
    for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * hw + (size_t)yy * W + xx] = acc;
  }
}

// ---- depthwise KSxKS, no padding (and its transpose) ---------------------------------------------
// block = 32x32 outputs of one plane; LDS halo tile (32+KS-1)^2; 4 outputs per thread (rows ty, ty+8, ty+16, ty+24)
constexpr int DW_T = 32, DW_KMAX = 15, DW_HALO = DW_T + DW_KMAX - 1;

template <bool ADJ>
__global__ __launch_bounds__(256) void k_depthwise(const float *__restrict__ x, const float *__restrict__ k,
                                                   float *__restrict__ out, int H, int W, int KS) {
  __shared__ float tile[DW_HALO * DW_HALO];
  __shared__ float kk[DW_KMAX * DW_KMAX];
  const int Hs = H - KS + 1, Ws = W - KS + 1;                  // the filtered (smaller) image
  const int Hi = ADJ ? Hs : H, Wi = ADJ ? Ws : W;              // input of this launch
  const int Ho = ADJ ? H : Hs, Wo = ADJ ? W : Ws;              // output of this launch
  const int x0 = blockIdx.x * DW_T, y0 = blockIdx.y * DW_T;
  const size_t p = blockIdx.z;
  const float *xp = x + p * (size_t)Hi * Wi;
  const int halo = DW_T + KS - 1, oy = ADJ ? y0 - (KS - 1) : y0, ox = ADJ ? x0 - (KS - 1) : x0;
  for (int e = threadIdx.x; e < KS * KS; e += 256) kk[e] = ADJ ? k[KS * KS - 1 - e] : k[e];
  for (int e = threadIdx.x; e < halo * halo; e += 256) {
    const int ly = e / halo, lx = e - ly * halo, gy = oy + ly, gx = ox + lx;
    tile[ly * DW_HALO + lx] = (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) ? xp[(size_t)gy * Wi + gx] : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < KS; ++i)
    for (int j = 0; j < KS; ++j) {
      const float w = kk[i * KS + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(w, tile[(ty + 8 * r + i) * DW_HALO + tx + j], acc[r]);
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + ty + 8 * r, xq = x0 + tx;
    if (y < Ho && xq < Wo) out[p * (size_t)Ho * Wo + (size_t)y * Wo + xq] = acc[r];
  }
}

static inline int in_chunks(long long planes, long long hw) {
  long long c = (1024 + planes - 1) / planes;
  const long long cmax = (hw / 4 + 1023) / 1024;
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return (int)c;
}

}  // namespace

extern "C" {

size_t hg_instnorm_workspace_bytes(int64_t P) { return P > 0 ? (size_t)P * 64 * 2 * sizeof(float) : 0; }

int hg_instnorm_lrelu_fwd(const float *x, float *out, float *stats, int64_t P, int32_t HW, float eps, float slope,
                          void *workspace, size_t workspace_bytes, void *stream) {
  if (!x || !out || !stats || P <= 0 || P > 0x7fffffff || HW <= 0) return HG_EINVAL;
  if (!workspace || workspace_bytes < hg_instnorm_workspace_bytes(P)) return HG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = in_chunks(P, HW);
  const dim3 grid((unsigned)P, chunks);
  hipLaunchKernelGGL(k_in_stats, grid, dim3(256), 0, st, x, (float *)workspace, HW, chunks);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_in_apply, grid, dim3(256), 0, st, x, (const float *)workspace, out, stats, HW, chunks, eps, slope);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_instnorm_lrelu_bwd(const float *gout, const float *out, const float *stats, float *gx, int64_t P, int32_t HW,
                          float slope, void *workspace, size_t workspace_bytes, void *stream) {
  if (!gout || !out || !stats || !gx || P <= 0 || P > 0x7fffffff || HW <= 0 || slope == 0.f) return HG_EINVAL;
  if (!workspace || workspace_bytes < hg_instnorm_workspace_bytes(P)) return HG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = in_chunks(P, HW);
  const dim3 grid((unsigned)P, chunks);
  hipLaunchKernelGGL(k_in_bwd_stats, grid, dim3(256), 0, st, gout, out, (float *)workspace, HW, chunks, slope);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_in_bwd_apply, grid, dim3(256), 0, st, gout, out, stats, (const float *)workspace, gx, HW, chunks,
                     slope);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_stencil3(const float *x, float *out, const float *taps9_host, int32_t B, int32_t C, int32_t H, int32_t W,
                int32_t adjoint, void *stream) {
  if (!x || !out || !taps9_host || B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  Taps9 tp;
  for (int i = 0; i < 9; ++i) tp.t[i] = taps9_host[i];
  const dim3 grid((W + 63) / 64, (H + 3) / 4, B);
  if (adjoint) hipLaunchKernelGGL(k_stencil3<true>, grid, dim3(256), 0, (hipStream_t)stream, x, out, tp, C, H, W);
  else hipLaunchKernelGGL(k_stencil3<false>, grid, dim3(256), 0, (hipStream_t)stream, x, out, tp, C, H, W);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_depthwise_valid(const float *x, const float *k, float *out, int64_t P, int32_t H, int32_t W, int32_t KS,
                       int32_t adjoint, void *stream) {
  if (!x || !k || !out || P <= 0 || P > 65535 || KS < 1 || KS > DW_KMAX || !(KS & 1) || H < KS || W < KS)
    return HG_EINVAL;
  const int Ho = adjoint ? H : H - KS + 1, Wo = adjoint ? W : W - KS + 1;
  const dim3 grid((Wo + DW_T - 1) / DW_T, (Ho + DW_T - 1) / DW_T, (unsigned)P);
  if (adjoint) hipLaunchKernelGGL(k_depthwise<true>, grid, dim3(256), 0, (hipStream_t)stream, x, k, out, H, W, KS);
  else hipLaunchKernelGGL(k_depthwise<false>, grid, dim3(256), 0, (hipStream_t)stream, x, k, out, H, W, KS);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
