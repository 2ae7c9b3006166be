// hg_nets.hip -- memory-bound kernels around the generator's dense contraction, and the optimizer.
//
// The reference's Conv2DMod (histoGAN/histoGAN.py:420-440) materialises per-sample weights
// W*(s+1)*d (4.8 GB for one layer at 256^2/B=32) and runs ONE grouped conv.  Here the modulation is
// moved onto the activations:   out = d[b,o] * conv(x*(s[b,:]+1), W)   with
// d[b,o] = rsqrt(sum_i (s[b,i]+1)^2 * sum_k W[o,i,k]^2 + 1e-8), so the dense contraction uses the
// shared weights.  This file holds the two fused elementwise stages either side of it:
//   prologue  k_modulate_*:        [bilinear x2 upsample ->] x*(s+1)       (+ adjoint, + dL/ds)
//   epilogue  k_demod_noise_lrelu: lrelu(conv*d + noise)                   (+ adjoint, + dL/dd, dL/dnoise-params)
// plus the fused flat-buffer DiffGrad step and EMA.  All HBM-bound: one read + one write per element.
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_nets.h"

namespace {

template <int NT>
__device__ __forceinline__ float block_sum(float v, float *sm) {
  v = hg_wave_sum(v);
  if constexpr (NT == 64) return v;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += sm[i];
  __syncthreads();
  return r;
}

// aten upsample_bilinear2d(scale 2, align_corners=False) taps of output index Y over a source of size H
__device__ __forceinline__ void up2_taps(int Y, int H, int &i0, int &i1, float &l) {
  const float src = fmaxf(0.5f * ((float)Y + 0.5f) - 0.5f, 0.f);  // exact in fp32
  i0 = min((int)src, H - 1);
  l = src - (float)i0;
  i1 = i0 + (i0 < H - 1 ? 1 : 0);
}

// weight of source index k in output index Y
__device__ __forceinline__ float up2_w(int Y, int k, int H) {
  int i0, i1; float l;
  up2_taps(Y, H, i0, i1, l);
  return (i0 == k ? 1.f - l : 0.f) + (i1 == k ? l : 0.f);
}

// ---- prologue -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_modulate_fwd(const float *__restrict__ x, const float *__restrict__ s,
                                                      float *__restrict__ out, long long n4, int hw4) {
  // no upsample; 4 elements per thread (H*W is a multiple of 4 for every layer >= 2x2)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float m = s ? s[i / hw4] + 1.f : 1.f;
    float4 v = reinterpret_cast<const float4 *>(x)[i];
    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
    reinterpret_cast<float4 *>(out)[i] = v;
  }
}

// one thread per SOURCE pixel: writes the 2x2 output block it anchors (polyphase: 3x3 taps -> 4 outputs)
__global__ __launch_bounds__(256) void k_up2_modulate_fwd(const float *__restrict__ x, const float *__restrict__ s,
                                                          float *__restrict__ out, int BC, int H, int W) {
  const long long total = (long long)BC * H * W;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int l = (int)(idx % W), k = (int)((idx / W) % H);
    const long long bc = idx / ((long long)W * H);
    const float m = s ? s[bc] + 1.f : 1.f;
    const float *xp = x + bc * H * W;
    float *op = out + bc * 4LL * H * W;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int Y = 2 * k + dy;
      int y0, y1; float ly;
      up2_taps(Y, H, y0, y1, ly);
      float2 o;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int X = 2 * l + dx;
        int x0, x1; float lx;
        up2_taps(X, W, x0, x1, lx);
        const float t0 = (1.f - lx) * xp[y0 * W + x0] + lx * xp[y0 * W + x1];
        const float t1 = (1.f - lx) * xp[y1 * W + x0] + lx * xp[y1 * W + x1];
        const float v = ((1.f - ly) * t0 + ly * t1) * m;
        if (dx == 0) o.x = v; else o.y = v;
      }
      *reinterpret_cast<float2 *>(op + (long long)Y * (2 * W) + 2 * l) = o;
    }
  }
}

// one block per (b,c) plane: t = up^T(g) (or g), gx = t*(s+1), gs = sum x*t
template <int NT>
__global__ __launch_bounds__(NT) void k_modulate_bwd(const float *__restrict__ gout, const float *__restrict__ x,
                                                     const float *__restrict__ s, float *__restrict__ gx,
                                                     float *__restrict__ gs, int H, int W, int upsample) {
  __shared__ float sm[4];
  const long long bc = blockIdx.x;
  const float m = s ? s[bc] + 1.f : 1.f;
  const float *xp = x + bc * H * W;
  float *gxp = gx + bc * H * W;
  float acc = 0.f;
  if (!upsample) {
    const float *gp = gout + bc * H * W;
    for (int e = threadIdx.x; e < H * W; e += NT) {
      const float t = gp[e];
      gxp[e] = t * m;
      acc = fmaf(xp[e], t, acc);
    }
  } else {
    const int H2 = 2 * H, W2 = 2 * W;
    const float *gp = gout + bc * (long long)H2 * W2;
    for (int e = threadIdx.x; e < H * W; e += NT) {
      const int k = e / W, l = e - k * W;
      float t = 0.f;
#pragma unroll
      for (int dy = -1; dy <= 2; ++dy) {
        const int Y = 2 * k + dy;
        if (Y < 0 || Y >= H2) continue;
        const float wy = up2_w(Y, k, H);
        if (wy == 0.f) continue;
#pragma unroll
        for (int dx = -1; dx <= 2; ++dx) {
          const int X = 2 * l + dx;
          if (X < 0 || X >= W2) continue;
          const float wx = up2_w(X, l, W);
          t = fmaf(wy * wx, gp[(long long)Y * W2 + X], t);
        }
      }
      gxp[e] = t * m;
      acc = fmaf(xp[e], t, acc);
    }
  }
  if (gs) {
    acc = block_sum<NT>(acc, sm);
    if (threadIdx.x == 0) gs[bc] = acc;
  }
}

// ---- epilogue -------------------------------------------------------------------------------
// nzt = the noise image already transposed: nzt[b][i][j] = inoise[b][j][i][0]  (S x S per sample)
__global__ __launch_bounds__(256) void k_dnl_fwd(const float *__restrict__ conv, const float *__restrict__ d,
                                                 const float *__restrict__ nzt, const float *__restrict__ wn,
                                                 const float *__restrict__ bn, float *__restrict__ out, int B, int O,
                                                 int H, int S) {
  const int hw = H * H;
  const long long total = (long long)B * O * hw;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int e = (int)(idx % hw);
    const long long bo = idx / hw;
    const int o = (int)(bo % O), b = (int)(bo / O);
    const int i = e / H, j = e - i * H;
    const float dd = d ? d[bo] : 1.f;
    const float v = fmaf(conv[idx], dd, fmaf(wn[o], nzt[((long long)b * S + i) * S + j], bn[o]));
    out[idx] = v > 0.f ? v : 0.2f * v;
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_dnl_bwd(const float *__restrict__ gout, const float *__restrict__ out,
                                                const float *__restrict__ conv, const float *__restrict__ d,
                                                const float *__restrict__ nzt, float *__restrict__ gconv,
                                                float *__restrict__ gd, float *__restrict__ gwn_part,
                                                float *__restrict__ gbn_part, int O, int H, int S) {
  __shared__ float sm[4];
  const long long bo = blockIdx.x;
  const int b = (int)(bo / O);
  const int hw = H * H;
  const float dd = d ? d[bo] : 1.f;
  const long long base = bo * hw;
  float a_d = 0.f, a_w = 0.f, a_b = 0.f;
  for (int e = threadIdx.x; e < hw; e += NT) {
    const float m = gout[base + e] * (out[base + e] > 0.f ? 1.f : 0.2f);
    gconv[base + e] = m * dd;
    const int i = e / H, j = e - i * H;
    a_d = fmaf(m, conv[base + e], a_d);
    a_w = fmaf(m, nzt[((long long)b * S + i) * S + j], a_w);
    a_b += m;
  }
  a_d = block_sum<NT>(a_d, sm);
  a_w = block_sum<NT>(a_w, sm);
  a_b = block_sum<NT>(a_b, sm);
  if (threadIdx.x == 0) {
    if (gd) gd[bo] = a_d;
    gwn_part[bo] = a_w;
    gbn_part[bo] = a_b;
  }
}

// ---- optimizer ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_diffgrad(float *__restrict__ p, const float *__restrict__ g,
                                                  float *__restrict__ m, float *__restrict__ v,
                                                  float *__restrict__ pg, long long n, float step_size, float b1,
                                                  float b2, float eps) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float dfc = 1.f / (1.f + expf(-fabsf(pg[i] - gi)));
    m[i] = mi; v[i] = vi; pg[i] = gi;
    p[i] -= step_size * (mi * dfc) / (sqrtf(vi) + eps);
  }
}

__global__ __launch_bounds__(256) void k_ema(float *__restrict__ ma, const float *__restrict__ p, long long n,
                                             float beta) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    ma[i] = ma[i] * beta + (1.f - beta) * p[i];
}

inline unsigned grid_for(long long n_threads) {
  long long b = (n_threads + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int hg_modulate_fwd(const float *x, const float *s, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t upsample, void *stream) {
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (upsample) {
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(k_up2_modulate_fwd, dim3(grid_for(total)), dim3(256), 0, st, x, s, out, B * C, H, W);
  } else {
    if ((H * W) % 4) return HG_EUNSUPPORTED;
    const long long n4 = (long long)B * C * H * W / 4;
    hipLaunchKernelGGL(k_modulate_fwd, dim3(grid_for(n4)), dim3(256), 0, st, x, s, out, n4, H * W / 4);
  }
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_modulate_bwd(const float *gout, const float *x, const float *s, float *gx, float *gs, int32_t B,
                    int32_t C, int32_t H, int32_t W, int32_t upsample, void *stream) {
  if (!gout || !x || !gx || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned planes = (unsigned)(B * C);
  if (H * W <= 256)
    hipLaunchKernelGGL((k_modulate_bwd<64>), dim3(planes), dim3(64), 0, st, gout, x, s, gx, gs, H, W, upsample);
  else
    hipLaunchKernelGGL((k_modulate_bwd<256>), dim3(planes), dim3(256), 0, st, gout, x, s, gx, gs, H, W, upsample);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_demod_noise_lrelu_fwd(const float *conv, const float *d, const float *nzt, const float *wn, const float *bn,
                             float *out, int32_t B, int32_t O, int32_t H, int32_t S, void *stream) {
  if (!conv || !nzt || !wn || !bn || !out || B <= 0 || O <= 0 || H <= 0 || S < H) return HG_EINVAL;
  const long long total = (long long)B * O * H * H;
  hipLaunchKernelGGL(k_dnl_fwd, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, conv, d, nzt, wn, bn, out,
                     B, O, H, S);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_demod_noise_lrelu_bwd(const float *gout, const float *out, const float *conv, const float *d,
                             const float *nzt, float *gconv, float *gd, float *gwn_part, float *gbn_part,
                             int32_t B, int32_t O, int32_t H, int32_t S, void *stream) {
  if (!gout || !out || !conv || !nzt || !gconv || !gwn_part || !gbn_part || B <= 0 || O <= 0 || H <= 0 || S < H)
    return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned planes = (unsigned)(B * O);
  if (H * H <= 256)
    hipLaunchKernelGGL((k_dnl_bwd<64>), dim3(planes), dim3(64), 0, st, gout, out, conv, d, nzt, gconv, gd, gwn_part,
                       gbn_part, O, H, S);
  else
    hipLaunchKernelGGL((k_dnl_bwd<256>), dim3(planes), dim3(256), 0, st, gout, out, conv, d, nzt, gconv, gd,
                       gwn_part, gbn_part, O, H, S);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_diffgrad_step(float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *prev_grad, int64_t n,
                     float lr, float beta1, float beta2, float eps, int32_t step, void *stream) {
  if (!p || !g || !exp_avg || !exp_avg_sq || !prev_grad || n <= 0 || step < 1) return HG_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  hipLaunchKernelGGL(k_diffgrad, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq,
                     prev_grad, (long long)n, step_size, beta1, beta2, eps);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_ema_update(float *ma, const float *p, int64_t n, float beta, void *stream) {
  if (!ma || !p || n <= 0) return HG_EINVAL;
  hipLaunchKernelGGL(k_ema, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ma, p, (long long)n, beta);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
