// hg_conv_b6.hip -- the 3x3 stride-1 convolution on the bf16 matrix cores at fp32 accuracy ("bf16x6").
//
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 rate.  An fp32 number is exactly the sum of three bf16
// numbers (3 x 8 significand bits: x = h + m + l, h = bf16(x), m = bf16(x-h), l = bf16(x-h-m)), and a
// bf16 x bf16 product is exact in the fp32 accumulator, so
//     x*y = hh + (hm + mh) + (hl + lh + mm) + [ml + lm + ll]
// with the bracket <= 2^-23 |xy|: six v_mfma_f32_32x32x16_bf16 per 16 input channels reproduce the fp32 product sum
// to fp32 rounding level (the accumulation is fp32 as before), at 6/16 of the fp32-MFMA time.
//
//   k_pack_b6   W (Co,Ci,3,3) -> Wb[tap][K/16][split h,m,l][k-half][n][8 bf16]   (one 16-byte MFMA A fragment per
//               (n, k-half): a wave reads its fragments straight from L2, no LDS for the weights)
//   k_conv_b6   D[channel][pixel] += sum over taps and 16-channel chunks; the input halo tile is converted while it
//               is staged: LDS holds [split][k-half][position][8 bf16], so the B fragment of a pixel is ONE
//               conflict-free ds_read_b128 and a tap is a shift of the position.
// Same tiling, pixel-tile geometry and epilogue as k_conv (hg_conv.hip).  Opt-in (see conv.py): parity tests hold it
// to the same bars as the fp32-MFMA kernels.
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_conv.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GeomB {
  int lTW, lTH, lNI;
  int TWp, IMS, HALO;
  float inv_TWp, inv_IMS, inv_HALO;
  int tiles_x, tiles_y, groups;
};

struct B6Args {
  const float *in;
  const bf16x8 *wt;   // [9][KCH][3][2][Np]
  float *out;
  const float *bias;
  int B, K, N, H, W, KCH, Np;
  GeomB g;
};

__device__ __forceinline__ int fdivb(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 hq = (__bf16)v[q];
    const float r1 = v[q] - (float)hq;
    const __bf16 mq = (__bf16)r1;
    const float r2 = r1 - (float)mq;
    h[q] = hq; m[q] = mq; l[q] = (__bf16)r2;
  }
}

// NPROD = 6: the products down to 2^-16 (error 2-4e-6 against fp64 where the fp32 MFMA has 1-2e-6);
// NPROD = 9: all nine -- the bf16 x bf16 partial products are exact in fp32 and sum to the exact fp32 product, so only
// the ACCUMULATION differs from the fp32 MFMA's fma chain ("bf16x9", 9/16 of the fp32-MFMA time on the matrix pipe).
template <int WC, int WP, int TC, int TP, int NPROD>
__global__ __launch_bounds__(WC *WP * 64, 2) void k_conv_b6(const B6Args a) {
  constexpr int NT = WC * WP * 64;
  constexpr int NB = WC * TC * 32, MB = WP * TP * 32;
  constexpr int R16 = MB == 256 ? 22 : (MB == 128 ? 26 : 36);   // halo positions per pixel, in 1/16 (see k_conv)
  constexpr int HMAX = MB * R16 / 16;
  constexpr int NI2 = (2 * HMAX + NT - 1) / NT;   // staging items (position, k-half) per thread

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16x8 *Xs = reinterpret_cast<bf16x8 *>(smem_raw);   // [3 splits][2 k-halves][HALO]

  const GeomB &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, lk = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;
  const int H = a.H, W = a.W, K = a.K, N = a.N, HW = H * W;

  int pt = blockIdx.x;
  const int tx = pt % g.tiles_x;
  pt /= g.tiles_x;
  const int ty = pt % g.tiles_y;
  const int grp = pt / g.tiles_y;
  const int x0 = tx << g.lTW, y0 = ty << g.lTH, b0 = grp << g.lNI;
  const int n0 = blockIdx.y * NB;
  const int TWm = (1 << g.lTW) - 1, THm = (1 << g.lTH) - 1;
  const int HALO = g.HALO;

  // staging items: (k-half, halo position) -> spatial offset of channel 0 of the image (or -1)
  int goff[NI2];
#pragma unroll
  for (int it = 0; it < NI2; ++it) {
    const int item = tid + it * NT;
    goff[it] = -1;
    if (item < 2 * HALO) {
      const int pos = item < HALO ? item : item - HALO;
      const int img = fdivb(pos, g.inv_IMS);
      const int rr = pos - img * g.IMS;
      const int hy = fdivb(rr, g.inv_TWp);
      const int hx = rr - hy * g.TWp;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1, b = b0 + img;
      if (b < a.B && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) goff[it] = (b * K * H + gy) * W + gx;
    }
  }

  int pixoff[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = (wp * TP + j) * 32 + lm;
    const int px = p & TWm, py = (p >> g.lTW) & THm, pi = p >> (g.lTW + g.lTH);
    pixoff[j] = lk * HALO + pi * g.IMS + py * g.TWp + px;
  }
  // A fragment address of (tap 0, chunk 0, split 0): + ((t*KCH + c)*3 + s)*2*Np per (t, c, s)
  const bf16x8 *wa = a.wt + (size_t)lk * a.Np + n0 + wc * TC * 32 + lm;
  const size_t sstr = (size_t)2 * a.Np;   // stride between splits

  f32x16 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float xr[NI2][8];
  auto prefetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI2; ++it) {
      const int item = tid + it * NT;
      const int kh = item < HALO ? 0 : 1;
      const int ch0 = c * 16 + kh * 8;
      const float *src = a.in + (size_t)ch0 * HW;
#pragma unroll
      for (int q = 0; q < 8; ++q) xr[it][q] = (goff[it] >= 0 && ch0 + q < K) ? src[goff[it] + q * HW] : 0.f;
    }
  };
  auto afrag = [&](int t, int c, bf16x8 (&f)[TC][3]) __attribute__((always_inline)) {
    const bf16x8 *p = wa + ((size_t)t * a.KCH + c) * 3 * sstr;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s) f[i][s] = p[s * sstr + i * 32];
  };

  bf16x8 acur[TC][3], anxt[TC][3];
  prefetch(0);
  afrag(0, 0, acur);
  for (int c = 0; c < a.KCH; ++c) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI2; ++it) {
      const int item = tid + it * NT;
      if (item < 2 * HALO) {
        bf16x8 h, m, l;
        split8(xr[it], h, m, l);
        Xs[item] = h;                 // [0][kh][pos] == item
        Xs[2 * HALO + item] = m;
        Xs[4 * HALO + item] = l;
      }
    }
    __syncthreads();
    if (c + 1 < a.KCH) prefetch(c + 1);

#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // next tap's (or next chunk's first) weight fragments are in flight during this tap's MFMAs
      if (t < 8) afrag(t + 1, c, anxt);
      else if (c + 1 < a.KCH) afrag(0, c + 1, anxt);
      const int toff = (t / 3) * g.TWp + (t % 3);
      bf16x8 bf[TP][3];
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int s = 0; s < 3; ++s) bf[j][s] = Xs[s * 2 * HALO + pixoff[j] + toff];
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
          f32x16 v = acc[i][j];
          if constexpr (NPROD == 9) {                                                       // smallest terms first
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][2], bf[j][2], v, 0, 0, 0);   // l*l
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][1], bf[j][2], v, 0, 0, 0);   // m*l
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][2], bf[j][1], v, 0, 0, 0);   // l*m
          }
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][0], bf[j][2], v, 0, 0, 0);   // h*l
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][2], bf[j][0], v, 0, 0, 0);   // l*h
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][1], bf[j][1], v, 0, 0, 0);   // m*m
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][0], bf[j][1], v, 0, 0, 0);   // h*m
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][1], bf[j][0], v, 0, 0, 0);   // m*h
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[i][0], bf[j][0], v, 0, 0, 0);   // h*h
          acc[i][j] = v;
        }
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s) acur[i][s] = anxt[i][s];
    }
  }

  // epilogue (as k_conv): row(i) = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel, col = lane&31 = pixel
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = (wp * TP + j) * 32 + lm;
    const int px = p & TWm, py = (p >> g.lTW) & THm, pi = p >> (g.lTW + g.lTH);
    const int cx = x0 + px, cy = y0 + py, b = b0 + pi;
    if (b >= a.B || cy >= H || cx >= W) continue;
    float *ob = a.out + ((size_t)b * N) * HW + cy * W + cx;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = n0 + (wc * TC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (ch < N) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[ch];
          ob[(size_t)ch * HW] = v;
        }
      }
  }
}

// one thread per 16-byte fragment element (t, c, kh, n): gathers its 8 weights, splits, writes the 3 splits
__global__ __launch_bounds__(256) void k_pack_b6(const float *__restrict__ w, bf16x8 *__restrict__ wt, int Co, int Ci,
                                                 int KCH, int Np, int mode) {
  const long long total = 9LL * KCH * 2 * Np;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int n = (int)(e % Np);
    long long r = e / Np;
    const int kh = (int)(r & 1);
    r >>= 1;
    const int c = (int)(r % KCH), t = (int)(r / KCH);
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = c * 16 + kh * 8 + q;
      float x = 0.f;
      if (n < N && k < K)
        x = mode == HG_CONV_PACK_FWD ? w[((size_t)n * Ci + k) * 9 + t] : w[((size_t)k * Ci + n) * 9 + (8 - t)];
      v[q] = x;
    }
    bf16x8 h, m, l;
    split8(v, h, m, l);
    const size_t base = ((((size_t)t * KCH + c) * 3) * 2 + kh) * Np + n;
    wt[base] = h;
    wt[base + 2 * (size_t)Np] = m;
    wt[base + 4 * (size_t)Np] = l;
  }
}

inline int ceil_log2b(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline int round_upb(int v, int m) { return (v + m - 1) / m * m; }

GeomB make_geom_b(int MB, int B, int H, int W) {
  GeomB g;
  int lTW = ceil_log2b(W < 4 ? 4 : W);
  if (lTW > 5) lTW = 5;
  const int lMB = ceil_log2b(MB);
  if (lTW > lMB - 1) lTW = lMB - 1;
  int lTH = ceil_log2b(H < 4 ? 4 : H);
  if (lTH > lMB - lTW) lTH = lMB - lTW;
  g.lTW = lTW; g.lTH = lTH; g.lNI = lMB - lTW - lTH;
  const int TW = 1 << lTW, TH = 1 << lTH, NI = 1 << g.lNI;
  g.TWp = TW + 2;
  g.IMS = (TH + 2) * g.TWp;
  g.HALO = NI * g.IMS;
  g.inv_TWp = 1.0f / (float)g.TWp;
  g.inv_IMS = 1.0f / (float)g.IMS;
  g.inv_HALO = 1.0f / (float)g.HALO;
  g.tiles_x = (W + TW - 1) / TW;
  g.tiles_y = (H + TH - 1) / TH;
  g.groups = (B + NI - 1) / NI;
  return g;
}

template <int WC, int WP, int TC, int TP, int NPROD>
int launch_b6(B6Args a, hipStream_t st) {
  constexpr int NB = WC * TC * 32, MB = WP * TP * 32, NT = WC * WP * 64;
  a.g = make_geom_b(MB, a.B, a.H, a.W);
  const size_t lds = (size_t)6 * a.g.HALO * 16;
  auto kern = k_conv_b6<WC, WP, TC, TP, NPROD>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid((unsigned)(a.g.tiles_x * a.g.tiles_y * a.g.groups), (unsigned)((a.N + NB - 1) / NB));
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, a);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

template <int NPROD>
int conv2d_bx(const float *in, const void *wt, float *out, const float *bias, int32_t B, int32_t K, int32_t N,
                     int32_t H, int32_t W, void *stream) {
  if (!in || !wt || !out || B <= 0 || K <= 0 || N <= 0 || H <= 0 || W <= 0) return HG_EINVAL;
  if ((long long)B * K * H * W >= 0x7fffffffLL || (long long)B * N * H * W >= 0x7fffffffLL) return HG_EUNSUPPORTED;
  B6Args a;
  a.in = in; a.wt = (const bf16x8 *)wt; a.out = out; a.bias = bias;
  a.B = B; a.K = K; a.N = N; a.H = H; a.W = W;
  a.KCH = round_upb(K, 16) / 16; a.Np = round_upb(N, 128);
  hipStream_t st = (hipStream_t)stream;
  const long long pix = (long long)B * H * W;
  auto blocks = [&](int nb, int mb) { return ((N + nb - 1) / nb) * ((pix + mb - 1) / mb); };
  const bool wide256 = W > 8 && H > 8, wide128 = W > 4 && H > 4;
  if (N <= 32 && wide256) return launch_b6<1, 4, 1, 2, NPROD>(a, st);
  if (N <= 64 && wide256 && blocks(64, 256) >= 384) return launch_b6<1, 4, 2, 2, NPROD>(a, st);
  if (N > 64 && wide128 && blocks(128, 128) >= 256) return launch_b6<2, 2, 2, 2, NPROD>(a, st);
  return launch_b6<2, 2, 1, 1, NPROD>(a, st);
}

}  // namespace

extern "C" {

size_t hg_conv_b6_packed_bytes(int32_t Co, int32_t Ci, int32_t mode) {
  if (Co <= 0 || Ci <= 0) return 0;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  return (size_t)9 * (round_upb(K, 16) / 16) * 3 * 2 * round_upb(N, 128) * 16;
}

int hg_conv_b6_pack_weights(const float *w, void *wt, int32_t Co, int32_t Ci, int32_t mode, void *stream) {
  if (!w || !wt || Co <= 0 || Ci <= 0 || (mode != HG_CONV_PACK_FWD && mode != HG_CONV_PACK_DGRAD)) return HG_EINVAL;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  const int KCH = round_upb(K, 16) / 16, Np = round_upb(N, 128);
  const long long total = 9LL * KCH * 2 * Np;
  long long nb = (total + 255) / 256;
  if (nb > 65535) nb = 65535;
  hipLaunchKernelGGL(k_pack_b6, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, (bf16x8 *)wt, Co, Ci, KCH, Np,
                     mode);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_conv2d_b6(const float *in, const void *wt, float *out, const float *bias, int32_t B, int32_t K, int32_t N,
                 int32_t H, int32_t W, void *stream) {
  return conv2d_bx<6>(in, wt, out, bias, B, K, N, H, W, stream);
}

int hg_conv2d_b9(const float *in, const void *wt, float *out, const float *bias, int32_t B, int32_t K, int32_t N,
                 int32_t H, int32_t W, void *stream) {
  return conv2d_bx<9>(in, wt, out, bias, B, K, N, H, W, stream);
}

}  // extern "C"
