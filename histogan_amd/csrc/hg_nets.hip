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
#include <cstdlib>
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

// Bilinear x2 (align_corners=False, edge clamp) is the separable polyphase filter
//   out[2k] = 1/4 x[k-1] + 3/4 x[k],  out[2k+1] = 3/4 x[k] + 1/4 x[k+1]      (indices clamped to the image)
// evaluated in aten's order ((1-lx) a + lx b along x, then along y) so that the values equal F.interpolate's.
// One thread per source pixel: reads its 3x3 neighbourhood, writes the 2x2 output block it anchors.
__global__ __launch_bounds__(256) void k_up2_modulate_fwd(const float *__restrict__ x, const float *__restrict__ s,
                                                          float *__restrict__ out, int H, int W) {
  const int bc = blockIdx.y;
  const float m = s ? s[bc] + 1.f : 1.f;
  const float *xp = x + (size_t)bc * H * W;
  float *op = out + (size_t)bc * 4 * H * W;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
    const int k = e / W, l = e - k * W;
    const int km = k > 0 ? k - 1 : 0, kp = k < H - 1 ? k + 1 : H - 1;
    const int lm = l > 0 ? l - 1 : 0, lp = l < W - 1 ? l + 1 : W - 1;
    const float a00 = xp[km * W + lm], a01 = xp[km * W + l], a02 = xp[km * W + lp];
    const float a10 = xp[k * W + lm], a11 = xp[k * W + l], a12 = xp[k * W + lp];
    const float a20 = xp[kp * W + lm], a21 = xp[kp * W + l], a22 = xp[kp * W + lp];
    // horizontal pass: even column 2l = (i0=l-1, lambda=.75) except l == 0 (i0=0, lambda=0); odd = (i0=l, lambda=.25)
    const float le = l > 0 ? 0.75f : 0.f;
    const float r0e = (1.f - le) * (l > 0 ? a00 : a01) + le * a01, r0o = 0.75f * a01 + 0.25f * a02;
    const float r1e = (1.f - le) * (l > 0 ? a10 : a11) + le * a11, r1o = 0.75f * a11 + 0.25f * a12;
    const float r2e = (1.f - le) * (l > 0 ? a20 : a21) + le * a21, r2o = 0.75f * a21 + 0.25f * a22;
    const float ke = k > 0 ? 0.75f : 0.f;
    float2 top, bot;
    top.x = ((1.f - ke) * (k > 0 ? r0e : r1e) + ke * r1e) * m;
    top.y = ((1.f - ke) * (k > 0 ? r0o : r1o) + ke * r1o) * m;
    bot.x = (0.75f * r1e + 0.25f * r2e) * m;
    bot.y = (0.75f * r1o + 0.25f * r2o) * m;
    *reinterpret_cast<float2 *>(op + (size_t)(2 * k) * (2 * W) + 2 * l) = top;
    *reinterpret_cast<float2 *>(op + (size_t)(2 * k + 1) * (2 * W) + 2 * l) = bot;
  }
}

// adjoint weight of source index k in output index Y (one axis): Y in {2k-1, 2k, 2k+1, 2k+2}
__device__ __forceinline__ float up2_adj_w(int d /* Y - 2k + 1 in 0..3 */, int k, int H) {
  // interior: 1/4, 3/4, 3/4, 1/4; the clamped edges fold the missing neighbour's share back in
  float w = (d == 0 || d == 3) ? 0.25f : 0.75f;
  if (k == 0 && d == 1) w = 1.f;
  if (k == H - 1 && d == 2) w = 1.f;
  return w;
}

// t = up^T(g) (or g), gx = t*(s+1), gs = sum x*t.  grid = (chunks, planes): partial sums per chunk, summed in
// fixed order by k_plane_sum_finish.
template <int UP>
__global__ __launch_bounds__(256) void k_modulate_bwd(const float *__restrict__ gout, const float *__restrict__ x,
                                                      const float *__restrict__ s, float *__restrict__ gx,
                                                      float *__restrict__ part, int H, int W) {
  __shared__ float sm[4];
  const int bc = blockIdx.y;
  const float m = s ? s[bc] + 1.f : 1.f;
  const float *xp = x + (size_t)bc * H * W;
  float *gxp = gx + (size_t)bc * H * W;
  float acc = 0.f;
  if (UP == 0) {
    const float *gp = gout + (size_t)bc * H * W;
    if (((H * W) & 3) == 0) {
      for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W / 4; e += gridDim.x * 256) {
        const float4 t = reinterpret_cast<const float4 *>(gp)[e];
        const float4 xv = reinterpret_cast<const float4 *>(xp)[e];
        float4 o;
        o.x = t.x * m; o.y = t.y * m; o.z = t.z * m; o.w = t.w * m;
        reinterpret_cast<float4 *>(gxp)[e] = o;
        acc += (xv.x * t.x + xv.y * t.y) + (xv.z * t.z + xv.w * t.w);
      }
    } else {
      for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
        const float t = gp[e];
        gxp[e] = t * m;
        acc = fmaf(xp[e], t, acc);
      }
    }
  } else if (UP == 1) {
    // general widths (odd W): one source pixel per thread, 16 predicated scalar loads
    const int H2 = 2 * H, W2 = 2 * W;
    const float *gp = gout + (size_t)bc * H2 * W2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
      const int k = e / W, l = e - k * W;
      float t = 0.f;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int Y = 2 * k - 1 + dy;
        if (Y < 0 || Y >= H2) continue;
        const float wy = up2_adj_w(dy, k, H);
        const float *row = gp + (size_t)Y * W2;
        float r = 0.f;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
          const int X = 2 * l - 1 + dx;
          if (X < 0 || X >= W2) continue;
          r = fmaf(up2_adj_w(dx, l, W), row[X], r);
        }
        t = fmaf(wy, r, t);
      }
      gxp[e] = t * m;
      acc = fmaf(xp[e], t, acc);
    }
  } else {
    // Adjoint of the bilinear x2: source pixel (k, l) gathers the 4 x 4 output pixels (2k-1 .. 2k+2) x (2l-1 .. 2l+2) with
    // weights (1/4, 3/4, 3/4, 1/4) per axis (edges folded).  A thread takes TWO neighbouring source pixels (l even): their
    // 4 x 6 window is one 16-byte load per row (columns 2l .. 2l+3) plus the left / right neighbour columns, which are
    // the adjacent lanes' loads (wave shuffles; a scalar load only at a wave's ends) -- 4 vector loads per two outputs
    // where the one-pixel-per-thread form issued 32 scalar ones at a stride of two (measured 1.2 TB/s).  Same fma order
    // per output as before: bit-identical gx.
    const int H2 = 2 * H, W2 = 2 * W, Wh = W >> 1;
    const float *gp = gout + (size_t)bc * H2 * W2;
    const int lane = threadIdx.x & 63;
    for (int e0 = blockIdx.x * 256; e0 < H * Wh; e0 += gridDim.x * 256) {    // (block-uniform bound: every lane shuffles)
      const int e = e0 + threadIdx.x;
      const bool live = e < H * Wh;
      const int k = live ? e / Wh : 0, lp = live ? e - k * Wh : 0, l = 2 * lp;
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int Y = 2 * k - 1 + dy;
        const bool yok = live && Y >= 0 && Y < H2;
        const float *row = gp + (size_t)(yok ? Y : 0) * W2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yok) v = *reinterpret_cast<const f32x4 *>(row + 2 * l);      // columns 2l .. 2l+3
        // column 2l-1 = left neighbour's .w, column 2l+4 = right neighbour's .x (same row: lp > 0 / lp < Wh-1)
        float gl = __shfl_up(v[3], 1, 64), gr = __shfl_down(v[0], 1, 64);
        if (yok && lane == 0 && lp > 0) gl = row[2 * l - 1];
        if (yok && lane == 63 && lp < Wh - 1) gr = row[2 * l + 4];
        if (!yok) continue;
        const float wy = up2_adj_w(dy, k, H);
        // pixel l: X = 2l-1 (dx 0), 2l, 2l+1, 2l+2;  pixel l+1: X = 2l+1 (dx 0), 2l+2, 2l+3, 2l+4
        float r0 = 0.f, r1 = 0.f;
        if (l > 0) r0 = fmaf(up2_adj_w(0, l, W), gl, r0);
        r0 = fmaf(up2_adj_w(1, l, W), v[0], r0);
        r0 = fmaf(up2_adj_w(2, l, W), v[1], r0);
        r0 = fmaf(up2_adj_w(3, l, W), v[2], r0);          // 2l+2 <= 2W-2: always inside (l <= W-2)
        r1 = fmaf(up2_adj_w(0, l + 1, W), v[1], r1);
        r1 = fmaf(up2_adj_w(1, l + 1, W), v[2], r1);
        r1 = fmaf(up2_adj_w(2, l + 1, W), v[3], r1);
        if (l + 1 < W - 1) r1 = fmaf(up2_adj_w(3, l + 1, W), gr, r1);
        t0 = fmaf(wy, r0, t0);
        t1 = fmaf(wy, r1, t1);
      }
      if (live) {
        const int o = k * W + l;
        *reinterpret_cast<float2 *>(gxp + o) = make_float2(t0 * m, t1 * m);
        const float2 xv = *reinterpret_cast<const float2 *>(xp + o);
        acc = fmaf(xv.x, t0, acc);
        acc = fmaf(xv.y, t1, acc);
      }
    }
  }
  if (part) {
    acc = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) part[(size_t)bc * gridDim.x + blockIdx.x] = acc;
  }
}

// out[plane] = sum_c part[plane][c]   (fixed order)
__global__ __launch_bounds__(256) void k_plane_sum_finish(const float *__restrict__ part, float *__restrict__ out,
                                                          int planes, int chunks) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= planes) return;
  float v = 0.f;
  for (int c = 0; c < chunks; ++c) v += part[(size_t)p * chunks + c];
  out[p] = v;
}

// ---- epilogue -------------------------------------------------------------------------------
// nzt = the noise image already transposed: nzt[b][i][j] = inoise[b][j][i][0]  (S x S per sample)
// grid = (chunks, B*O); 4 pixels per thread when H % 4 == 0
__global__ __launch_bounds__(256) void k_dnl_fwd(const float *__restrict__ conv, const float *__restrict__ d,
                                                 const float *__restrict__ nzt, const float *__restrict__ wn,
                                                 const float *__restrict__ bn, float *__restrict__ out, int O,
                                                 int H, int S) {
  const int bo = blockIdx.y, o = bo % O, b = bo / O;
  const int hw = H * H;
  const float dd = d ? d[bo] : 1.f, w = wn[o], bb = bn[o];
  const float *cp = conv + (size_t)bo * hw;
  float *op = out + (size_t)bo * hw;
  const float *np = nzt + (size_t)b * S * S;
  if ((H & 3) == 0) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < hw / 4; e += gridDim.x * 256) {
      const int i = (e * 4) / H, j = e * 4 - i * H;
      const float4 c = reinterpret_cast<const float4 *>(cp)[e];
      const float4 n = *reinterpret_cast<const float4 *>(np + (size_t)i * S + j);   // S % 4 == 0 too (S >= H, pow 2)
      float4 v;
      v.x = fmaf(c.x, dd, fmaf(w, n.x, bb)); v.y = fmaf(c.y, dd, fmaf(w, n.y, bb));
      v.z = fmaf(c.z, dd, fmaf(w, n.z, bb)); v.w = fmaf(c.w, dd, fmaf(w, n.w, bb));
      v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
      v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
      reinterpret_cast<float4 *>(op)[e] = v;
    }
  } else {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < hw; e += gridDim.x * 256) {
      const int i = e / H, j = e - i * H;
      const float v = fmaf(cp[e], dd, fmaf(w, np[(size_t)i * S + j], bb));
      op[e] = v > 0.f ? v : 0.2f * v;
    }
  }
}

// m = gout * lrelu'(out): gconv = m*d; partial sums of m*conv, m*nzt, m per (chunk, plane)
// conv == NULL (fused forward: the pre-activation was never stored): conv*d is recovered from out,
//   pre = out > 0 ? out : out / 0.2;  conv*d = pre - (wn*nzt + bn), and the first partial sum is sum m*conv*d
//   (the caller divides by d).
__global__ __launch_bounds__(256) void k_dnl_bwd(const float *__restrict__ gout, const float *__restrict__ out,
                                                 const float *__restrict__ conv, const float *__restrict__ d,
                                                 const float *__restrict__ nzt, const float *__restrict__ wn,
                                                 const float *__restrict__ bn, float *__restrict__ gconv,
                                                 float *__restrict__ part /* [3][planes][chunks] */, int O, int H, int S) {
  __shared__ float sm[4];
  const int bo = blockIdx.y, b = bo / O;
  const int hw = H * H;
  const float dd = d ? d[bo] : 1.f;
  const size_t base = (size_t)bo * hw;
  const float *np = nzt + (size_t)b * S * S;
  const float w_ = conv ? 0.f : wn[bo % O], b_ = conv ? 0.f : bn[bo % O];
  float a_d = 0.f, a_w = 0.f, a_b = 0.f;
  if ((H & 3) == 0 && (S & 3) == 0) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < hw / 4; e += gridDim.x * 256) {
      const int i = (e * 4) / H, j = e * 4 - i * H;
      const float4 g = reinterpret_cast<const float4 *>(gout + base)[e];
      const float4 o = reinterpret_cast<const float4 *>(out + base)[e];
      const float4 n = *reinterpret_cast<const float4 *>(np + (size_t)i * S + j);
      float4 c;
      if (conv) {
        c = reinterpret_cast<const float4 *>(conv + base)[e];
      } else {
        c.x = (o.x > 0.f ? o.x : 5.f * o.x) - fmaf(w_, n.x, b_); c.y = (o.y > 0.f ? o.y : 5.f * o.y) - fmaf(w_, n.y, b_);
        c.z = (o.z > 0.f ? o.z : 5.f * o.z) - fmaf(w_, n.z, b_); c.w = (o.w > 0.f ? o.w : 5.f * o.w) - fmaf(w_, n.w, b_);
      }
      float4 m;
      m.x = g.x * (o.x > 0.f ? 1.f : 0.2f); m.y = g.y * (o.y > 0.f ? 1.f : 0.2f);
      m.z = g.z * (o.z > 0.f ? 1.f : 0.2f); m.w = g.w * (o.w > 0.f ? 1.f : 0.2f);
      float4 r;
      r.x = m.x * dd; r.y = m.y * dd; r.z = m.z * dd; r.w = m.w * dd;
      reinterpret_cast<float4 *>(gconv + base)[e] = r;
      a_d += (m.x * c.x + m.y * c.y) + (m.z * c.z + m.w * c.w);
      a_w += (m.x * n.x + m.y * n.y) + (m.z * n.z + m.w * n.w);
      a_b += (m.x + m.y) + (m.z + m.w);
    }
  } else {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < hw; e += gridDim.x * 256) {
      const float o = out[base + e];
      const float m = gout[base + e] * (o > 0.f ? 1.f : 0.2f);
      gconv[base + e] = m * dd;
      const int i = e / H, j = e - i * H;
      const float nz = np[(size_t)i * S + j];
      const float c = conv ? conv[base + e] : (o > 0.f ? o : 5.f * o) - fmaf(w_, nz, b_);
      a_d = fmaf(m, c, a_d);
      a_w = fmaf(m, nz, a_w);
      a_b += m;
    }
  }
  a_d = block_sum<256>(a_d, sm);
  a_w = block_sum<256>(a_w, sm);
  a_b = block_sum<256>(a_b, sm);
  if (threadIdx.x == 0) {
    const size_t planes = gridDim.y, idx = (size_t)bo * gridDim.x + blockIdx.x;
    part[idx] = a_d;
    part[planes * gridDim.x + idx] = a_w;
    part[2 * planes * gridDim.x + idx] = a_b;
  }
}

// gd / gwn_part / gbn_part [plane] = sum over chunks of the three partial arrays (fixed order)
// ddiv != NULL (conv was recovered from out: the first sum is of m * conv * d): gd = that sum / d
__global__ __launch_bounds__(256) void k_dnl_bwd_finish(const float *__restrict__ part, float *__restrict__ gd,
                                                        float *__restrict__ gw, float *__restrict__ gb,
                                                        const float *__restrict__ ddiv, int planes, int chunks) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= planes) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  for (int c = 0; c < chunks; ++c) {
    v0 += part[(size_t)p * chunks + c];
    v1 += part[(size_t)planes * chunks + (size_t)p * chunks + c];
    v2 += part[2 * (size_t)planes * chunks + (size_t)p * chunks + c];
  }
  if (gd) gd[p] = ddiv ? v0 / ddiv[p] : v0;
  gw[p] = v1;
  gb[p] = v2;
}

// sum over (b, h, w) of a (B, C, HW) tensor -> out[C]  (bias gradient); grid = (chunks, C), partials + finish
__global__ __launch_bounds__(256) void k_channel_sum(const float *__restrict__ g, float *__restrict__ part, int B, int C,
                                                     int HW) {
  __shared__ float sm[4];
  const int c = blockIdx.y;
  float acc = 0.f;
  const long long per = (long long)B * HW;   // elements of channel c: (b, p) -> g[(b*C + c)*HW + p]
  if ((HW & 3) == 0) {
    const int hw4 = HW / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per / 4; e += (long long)gridDim.x * 256) {
      const long long b = e / hw4, p4 = e - b * hw4;
      const float4 v = reinterpret_cast<const float4 *>(g + ((size_t)b * C + c) * HW)[p4];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long long)gridDim.x * 256) {
      const long long b = e / HW, p = e - b * HW;
      acc += g[((size_t)b * C + c) * HW + p];
    }
  }
  acc = block_sum<256>(acc, sm);
  if (threadIdx.x == 0) part[(size_t)c * gridDim.x + blockIdx.x] = acc;
}

// gm = g * (out > 0 ? 1 : slope) (the gradient through LeakyReLU, decided on the OUTPUT: sign(out) == sign(pre-activation))
// and the per-channel sum of gm (the bias gradient of the convolution in front of it) from the same pass.
// grid = (chunks, C), partials + k_plane_sum_finish as k_channel_sum.
__global__ __launch_bounds__(256) void k_lrelu_bwd_csum(const float *__restrict__ g, const float *__restrict__ out,
                                                        float *__restrict__ gm, float *__restrict__ part, float slope,
                                                        int B, int C, int HW) {
  __shared__ float sm[4];
  const int c = blockIdx.y;
  float acc = 0.f;
  const long long per = (long long)B * HW;
  if ((HW & 3) == 0) {
    const int hw4 = HW / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per / 4; e += (long long)gridDim.x * 256) {
      const long long b = e / hw4, p4 = e - b * hw4;
      const size_t off = ((size_t)b * C + c) * HW;
      const float4 v = reinterpret_cast<const float4 *>(g + off)[p4];
      const float4 o = reinterpret_cast<const float4 *>(out + off)[p4];
      float4 r;
      r.x = o.x > 0.f ? v.x : v.x * slope; r.y = o.y > 0.f ? v.y : v.y * slope;
      r.z = o.z > 0.f ? v.z : v.z * slope; r.w = o.w > 0.f ? v.w : v.w * slope;
      reinterpret_cast<float4 *>(gm + off)[p4] = r;
      acc += (r.x + r.y) + (r.z + r.w);
    }
  } else {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long long)gridDim.x * 256) {
      const long long b = e / HW, p = e - b * HW;
      const size_t off = ((size_t)b * C + c) * HW + p;
      const float v = g[off], r = out[off] > 0.f ? v : v * slope;
      gm[off] = r;
      acc += r;
    }
  }
  if (part) {
    acc = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) part[(size_t)c * gridDim.x + blockIdx.x] = acc;
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

// the same update with the bias-corrected step size read from device memory: the launch carries no per-step host
// value, so it can live in a captured hipGraph (the host refreshes *step_size_dev before each replay)
__global__ __launch_bounds__(256) void k_diffgrad_dev(float *__restrict__ p, const float *__restrict__ g,
                                                      float *__restrict__ m, float *__restrict__ v,
                                                      float *__restrict__ pg, long long n,
                                                      const float *__restrict__ step_size_dev, float b1, float b2,
                                                      float eps) {
  const float step_size = *step_size_dev;
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

// Demodulation backward, weight side.  d[b,o] = rsqrt(sum_{i,t} (w[o,i,t] (s[b,i]+1))^2 + eps)  (Conv2DMod,
// histoGAN/histoGAN.py:427-429, on the shared weight), gd = dL/dd:
//   gw[o,i,t] (+)= 2 w[o,i,t] M[o,i],   M[o,i] = sum_b gq[b,o] s1[b,i]^2,   gq = gd * (-0.5) * d^3,   s1 = s + 1
// One block per output channel o: gq's column and M's row in LDS (B-deep dot products, style read coalesced), then one
// coalesced pass over the K*T weights of that channel.  (As aten ops this was a 2048x32x2048 rocBLAS GEMM -- 314 us, the
// library has no kernel for a 32-deep reduction -- plus two element-wise passes over the weight and the gradient add.)
__global__ __launch_bounds__(256) void k_demod_weight_term(const float *__restrict__ w, const float *__restrict__ gd,
                                                           const float *__restrict__ d, const float *__restrict__ s1,
                                                           float *__restrict__ gw, int B, int N, int K, int T,
                                                           int accumulate) {
  extern __shared__ float dsm[];
  float *Mrow = dsm, *gqc = dsm + K;
  const int o = blockIdx.x;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float dd = d[(size_t)b * N + o];
    gqc[b] = gd[(size_t)b * N + o] * (-0.5f) * dd * dd * dd;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += 256) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) {
      const float sv = s1[(size_t)b * K + i];
      m = fmaf(gqc[b], sv * sv, m);
    }
    Mrow[i] = 2.f * m;
  }
  __syncthreads();
  const size_t base = (size_t)o * K * T;
  const int n = K * T;
  if ((n & 3) == 0) {
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(w + base);
    f32x4 *g4 = reinterpret_cast<f32x4 *>(gw + base);
    for (int e = threadIdx.x; e < n / 4; e += 256) {
      f32x4 v = w4[e];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] *= Mrow[(4 * e + c) / T];
      if (accumulate) v += g4[e];
      g4[e] = v;
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) {
      const float v = w[base + e] * Mrow[e / T];
      gw[base + e] = accumulate ? gw[base + e] + v : v;
    }
  }
}

// Demodulation backward, style side:  gy[b,i] = 2 s1[b,i] sum_o gq[b,o] wsq[o,i],  gq = gd * (-0.5) * d^3,
// wsq[o,i] = sum_t w[o,i,t]^2.  A (B x N) @ (N x K) product with B = 32: rocBLAS runs it as 64 workgroups without a split
// over the N-deep reduction (262 us at N = K = 2048, and no faster as a chunked batched GEMM) at the end of the
// generator's backward chain.  Here: grid (K / 64 column tiles, NS row splits), a thread per column with B_T running sums,
// gq's rows of the split in LDS; the NS partial results are summed in fixed order by k_demod_style_grad_finish.
constexpr int DSG_BT = 32;   // batch rows per pass
constexpr int DSG_LD = DSG_BT + 1;   // LDS pitch of a gq row (odd: the transposing fill is conflict-free)
__global__ __launch_bounds__(256) void k_demod_style_grad(const float *__restrict__ gd, const float *__restrict__ d,
                                                          const float *__restrict__ wsq, float *__restrict__ part, int B,
                                                          int N, int K, int rows_per) {
  extern __shared__ float dsm[];
  float *gq = dsm;                       // [rows_per][DSG_LD]
  float *red = dsm + rows_per * DSG_LD;  // [4][DSG_BT][64]
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const int o0 = blockIdx.y * rows_per, o1 = min(N, o0 + rows_per);
  for (int b0 = 0; b0 < B; b0 += DSG_BT) {
    const int nb = min(DSG_BT, B - b0);
    __syncthreads();
    const int rows = o1 - o0;
    for (int e = threadIdx.x; e < rows * DSG_BT; e += 256) {
      const int b = e / rows, r = e - b * rows;      // consecutive threads: consecutive channels of one batch row (coalesced)
      float v = 0.f;
      if (b < nb) {
        const float dd = d[(size_t)(b0 + b) * N + o0 + r];
        v = gd[(size_t)(b0 + b) * N + o0 + r] * (-0.5f) * dd * dd * dd;
      }
      gq[r * DSG_LD + b] = v;
    }
    __syncthreads();
    float acc[DSG_BT];
#pragma unroll
    for (int b = 0; b < DSG_BT; ++b) acc[b] = 0.f;
    if (i < K) {
      for (int o = o0 + ty; o < o1; o += 4) {
        const float w = wsq[(size_t)o * K + i];
        const float *g = gq + (o - o0) * DSG_LD;
#pragma unroll
        for (int b = 0; b < DSG_BT; ++b) acc[b] = fmaf(g[b], w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < DSG_BT; ++b) red[(ty * DSG_BT + b) * 64 + tx] = acc[b];
    __syncthreads();
    // [4][BT][64] -> part[split][b][i]: fixed-order sum of the 4 row groups
    for (int e = threadIdx.x; e < DSG_BT * 64; e += 256) {
      const int b = e >> 6, c = e & 63;
      if (b < nb && blockIdx.x * 64 + c < K)
        part[((size_t)blockIdx.y * B + b0 + b) * K + blockIdx.x * 64 + c] =
            (red[(0 * DSG_BT + b) * 64 + c] + red[(1 * DSG_BT + b) * 64 + c]) +
            (red[(2 * DSG_BT + b) * 64 + c] + red[(3 * DSG_BT + b) * 64 + c]);
    }
  }
}

__global__ __launch_bounds__(256) void k_demod_style_grad_finish(const float *__restrict__ part, const float *__restrict__ s1,
                                                                 float *__restrict__ gy, int BK, int splits) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= BK) return;
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += part[(size_t)z * BK + e];
  gy[e] = 2.f * s1[e] * v;
}

}  // namespace

extern "C" {

// chunks per plane for the partial-sum kernels: ~1024 blocks in total, each thread >= 4 vector iterations
static inline int plane_chunks(long long planes, long long vec_per_plane) {
  static const long long target = [] { const char *e = getenv("HG_NETS_BLOCKS"); return e && atoll(e) > 0 ? atoll(e) : 1024LL; }();
  long long c = (target + planes - 1) / planes;
  const long long cmax = (vec_per_plane + 1023) / 1024;   // >= 4 iterations of 256 threads
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return (int)c;
}

size_t hg_nets_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)3 * B * C * 64 * sizeof(float);   // 3 partial arrays x planes x (<= 64 chunks)
}

int hg_modulate_fwd(const float *x, const float *s, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t upsample, void *stream) {
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (upsample) {
    const int chunks = plane_chunks((long long)B * C, (long long)H * W);
    hipLaunchKernelGGL(k_up2_modulate_fwd, dim3(chunks, B * C), dim3(256), 0, st, x, s, out, H, W);
  } else {
    if ((H * W) % 4) return HG_EUNSUPPORTED;
    const long long n4 = (long long)B * C * H * W / 4;
    hipLaunchKernelGGL(k_modulate_fwd, dim3(grid_for(n4)), dim3(256), 0, st, x, s, out, n4, H * W / 4);
  }
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_modulate_bwd(const float *gout, const float *x, const float *s, float *gx, float *gs, int32_t B,
                    int32_t C, int32_t H, int32_t W, int32_t upsample, void *workspace, size_t workspace_bytes,
                    void *stream) {
  if (!gout || !x || !gx || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * C;
  // even widths with 16-byte-aligned rows: the two-pixels-per-thread adjoint (k_modulate_bwd<2>)
  const bool fast_up = upsample && (W & 1) == 0 && (((uintptr_t)gout | (uintptr_t)gx | (uintptr_t)x) & 15) == 0;
  const int chunks = plane_chunks(planes, (long long)H * W / (upsample ? (fast_up ? 2 : 1) : 4));
  float *part = nullptr;
  if (gs) {
    if (!workspace || workspace_bytes < (size_t)planes * chunks * sizeof(float)) return HG_EWORKSPACE;
    part = (float *)workspace;
  }
  if (upsample && fast_up)
    hipLaunchKernelGGL((k_modulate_bwd<2>), dim3(chunks, planes), dim3(256), 0, st, gout, x, s, gx, part, H, W);
  else if (upsample)
    hipLaunchKernelGGL((k_modulate_bwd<1>), dim3(chunks, planes), dim3(256), 0, st, gout, x, s, gx, part, H, W);
  else
    hipLaunchKernelGGL((k_modulate_bwd<0>), dim3(chunks, planes), dim3(256), 0, st, gout, x, s, gx, part, H, W);
  HG_LAUNCH_CHECK();
  if (gs) {
    hipLaunchKernelGGL(k_plane_sum_finish, dim3((planes + 255) / 256), dim3(256), 0, st, part, gs, planes, chunks);
    HG_LAUNCH_CHECK();
  }
  return HG_OK;
}

int hg_demod_noise_lrelu_fwd(const float *conv, const float *d, const float *nzt, const float *wn, const float *bn,
                             float *out, int32_t B, int32_t O, int32_t H, int32_t S, void *stream) {
  if (!conv || !nzt || !wn || !bn || !out || B <= 0 || O <= 0 || H <= 0 || S < H) return HG_EINVAL;
  const int chunks = plane_chunks((long long)B * O, (long long)H * H / 4);
  hipLaunchKernelGGL(k_dnl_fwd, dim3(chunks, B * O), dim3(256), 0, (hipStream_t)stream, conv, d, nzt, wn, bn, out, O, H, S);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_demod_noise_lrelu_bwd(const float *gout, const float *out, const float *conv, const float *d,
                             const float *nzt, const float *wn, const float *bn, float *gconv, float *gd,
                             float *gwn_part, float *gbn_part, int32_t B, int32_t O, int32_t H, int32_t S,
                             void *workspace, size_t workspace_bytes, void *stream) {
  if (!gout || !out || !nzt || !gconv || !gwn_part || !gbn_part || B <= 0 || O <= 0 || H <= 0 || S < H)
    return HG_EINVAL;
  if (!conv && (!wn || !bn)) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * O;
  const int chunks = plane_chunks(planes, (long long)H * H / 4);
  if (!workspace || workspace_bytes < (size_t)3 * planes * chunks * sizeof(float)) return HG_EWORKSPACE;
  float *part = (float *)workspace;
  hipLaunchKernelGGL(k_dnl_bwd, dim3(chunks, planes), dim3(256), 0, st, gout, out, conv, d, nzt, wn, bn, gconv, part, O, H,
                     S);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dnl_bwd_finish, dim3((planes + 255) / 256), dim3(256), 0, st, part, gd, gwn_part, gbn_part,
                     (!conv && d) ? d : nullptr, planes, chunks);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_channel_sum(const float *g, float *out, int32_t B, int32_t C, int32_t HW, void *workspace, size_t workspace_bytes,
                   void *stream) {
  if (!g || !out || B <= 0 || C <= 0 || HW <= 0) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = plane_chunks(C, (long long)B * HW / 4);
  if (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(float)) return HG_EWORKSPACE;
  hipLaunchKernelGGL(k_channel_sum, dim3(chunks, C), dim3(256), 0, st, g, (float *)workspace, B, C, HW);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_plane_sum_finish, dim3((C + 255) / 256), dim3(256), 0, st, (const float *)workspace, out, C, chunks);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

static int demod_style_splits(int N) {
  int ns = (N + 63) / 64;   // <= 64 rows of gq per block in LDS (up to N = 4096)
  return ns > 64 ? 64 : (ns < 1 ? 1 : ns);
}

size_t hg_demod_style_grad_workspace_bytes(int32_t B, int32_t N, int32_t K) {
  if (B <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)demod_style_splits(N) * B * K * sizeof(float);
}

int hg_demod_style_grad(const float *gd, const float *d, const float *s1, const float *wsq, float *gy, int32_t B, int32_t N,
                        int32_t K, void *workspace, size_t workspace_bytes, void *stream) {
  if (!gd || !d || !s1 || !wsq || !gy || B <= 0 || N <= 0 || K <= 0) return HG_EINVAL;
  if (!workspace || workspace_bytes < hg_demod_style_grad_workspace_bytes(B, N, K)) return HG_EWORKSPACE;
  const int ns = demod_style_splits(N), rows_per = (N + ns - 1) / ns;
  const size_t lds = ((size_t)rows_per * DSG_LD + 4 * DSG_BT * 64) * sizeof(float);
  if (lds > 64 * 1024) return HG_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_demod_style_grad, dim3((unsigned)((K + 63) / 64), (unsigned)ns), dim3(256), lds, st, gd, d, wsq,
                     (float *)workspace, B, N, K, rows_per);
  HG_LAUNCH_CHECK();
  const int BK = B * K;
  hipLaunchKernelGGL(k_demod_style_grad_finish, dim3((unsigned)((BK + 255) / 256)), dim3(256), 0, st,
                     (const float *)workspace, s1, gy, BK, ns);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_demod_weight_term(const float *w, const float *gd, const float *d, const float *s1, float *gw, int32_t B,
                         int32_t N, int32_t K, int32_t taps, int32_t accumulate, void *stream) {
  if (!w || !gd || !d || !s1 || !gw || B <= 0 || N <= 0 || K <= 0 || taps <= 0) return HG_EINVAL;
  const size_t lds = (size_t)(K + B) * sizeof(float);
  if (lds > 64 * 1024) return HG_EUNSUPPORTED;
  hipLaunchKernelGGL(k_demod_weight_term, dim3((unsigned)N), dim3(256), lds, (hipStream_t)stream, w, gd, d, s1, gw, B, N,
                     K, taps, accumulate);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_lrelu_bwd_channel_sum(const float *g, const float *out, float slope, float *gm, float *csum, int32_t B, int32_t C,
                             int32_t HW, void *workspace, size_t workspace_bytes, void *stream) {
  if (!g || !out || !gm || B <= 0 || C <= 0 || HW <= 0 || !(slope >= 0.f)) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = plane_chunks(C, (long long)B * HW / 4);
  if (csum && (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(float))) return HG_EWORKSPACE;
  hipLaunchKernelGGL(k_lrelu_bwd_csum, dim3(chunks, C), dim3(256), 0, st, g, out, gm, csum ? (float *)workspace : nullptr,
                     slope, B, C, HW);
  HG_LAUNCH_CHECK();
  if (csum) {
    hipLaunchKernelGGL(k_plane_sum_finish, dim3((C + 255) / 256), dim3(256), 0, st, (const float *)workspace, csum, C, chunks);
    HG_LAUNCH_CHECK();
  }
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

float hg_diffgrad_step_size(float lr, float beta1, float beta2, int32_t step) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  return (float)((double)lr * sqrt(bc2) / bc1);
}

int hg_diffgrad_step_dev(float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *prev_grad, int64_t n,
                         const float *step_size_dev, float beta1, float beta2, float eps, void *stream) {
  if (!p || !g || !exp_avg || !exp_avg_sq || !prev_grad || !step_size_dev || n <= 0) return HG_EINVAL;
  hipLaunchKernelGGL(k_diffgrad_dev, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq,
                     prev_grad, (long long)n, step_size_dev, beta1, beta2, eps);
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
