// hg_gstage.hip -- the backward of everything BETWEEN two convolutions of the generator as ONE pass (include/hg_nets.h:
// hg_gstage_bwd).
//
// A generator stage output  out = lrelu_0.2(d[b,c] conv + wn[c] nz + bn[c])  (GeneratorBlock.forward,
// histoGAN/histoGAN.py:461-479; Conv2DMod :420-440) feeds up to two consumers:
//   A  the next modulated 3x3 convolution, either at the same resolution (conv2 of the block: xm = out (sa + 1)) or behind the
//      bilinear x2 of the next block (xm = up2(out) (sa + 1), :447-448, 463-464);
//   R  the block's to-RGB 1x1 modulated convolution without demodulation (RGBBlock, :380-390).
// With autograd these were five to six HBM-bound launches in a row on the critical path of the backward, between the data
// gradient of one convolution and the next: the modulation adjoint (+ bilinear adjoint), the to-RGB adjoint, the sum of the
// two gradients, the LeakyReLU / noise / demodulation adjoint -- 14 ... 20 tensor passes per block where 6 ... 9 are needed:
//
//   t  = ga              (A at the same resolution)      or  up2^T(ga)   (A behind the upsample; edge-folded 4x4 gather)
//   tr = sum_k w_rgb[k,c] g_rgb[b,k,p]
//   G  = t (sa[b,c] + 1) + tr (s_rgb[b,c] + 1)                                   d loss / d out
//   m  = G * (out > 0 ? 1 : 0.2)                                                 d loss / d pre-activation
//   gconv = m d[b,c]                                                             -> the stage's data / weight gradient
//   sums over the pixels of plane (b, c):  out . t  (style gradient of A),  out . g_rgb[k]  (style + weight gradient of R),
//   m . (pre - wn nz - bn)  (d's gradient x d),  m . nz,  m  (noise weight / bias gradients)
//
// Thread layout: PX = 4 pixels per thread (2 source pixels when A is behind the upsample: k_modulate_bwd<2>'s 16-byte row
// loads + neighbour shuffles), LP lanes per plane and pass, 256 / LP planes per block on the small maps (4x4: 4 lanes per
// plane, 64 planes per block -- the one-block-per-plane kernels it replaces ran 65 536 workgroups of 4 ... 16 live lanes
// there); on the large maps a plane is split into chunks (grid y) whose partial sums the finish kernel adds in fixed order.
#include <cstdlib>
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_nets.h"

namespace {

constexpr int kNS = 8;      // partial sums per (plane, chunk): out.t, out.g_rgb[0..3], m.(conv d), m.nz, m
constexpr int kMaxCr = 4;

// adjoint weight of source index k in output index Y (one axis), d = Y - 2k + 1 in 0..3 (as in hg_nets.hip)
__device__ __forceinline__ float up2_adj_w(int d, int k, int H) {
  float w = (d == 0 || d == 3) ? 0.25f : 0.75f;
  if (k == 0 && d == 1) w = 1.f;
  if (k == H - 1 && d == 2) w = 1.f;
  return w;
}

struct GStageArgs {
  const float *out, *ga, *sa, *g_rgb, *w_rgb, *s_rgb, *d, *nzt, *wn, *bn;
  float *gconv, *part;
  int planes, C, H, S, Cr, LP, chunks;
};

// sum over the LP (power of two, 4 ... 256) lanes that share a plane; valid in the group's lane 0
template <int NR>
__device__ __forceinline__ void group_sum(float (&r)[NR], int LP, float *sm /* [4][NR] */) {
  const int w = LP < 64 ? LP : 64;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    float v = r[i];
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    r[i] = v;
  }
  if (LP > 64) {      // (LP == 256: one plane per block, block-uniform branch)
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < NR; ++i) sm[wv * NR + i] = r[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NR; ++i) r[i] = (sm[i] + sm[NR + i]) + (sm[2 * NR + i] + sm[3 * NR + i]);
    }
    __syncthreads();
  }
}

template <int UP>
__global__ __launch_bounds__(256) void k_gstage_bwd(const GStageArgs a) {
  constexpr int PX = UP ? 2 : 4;
  __shared__ float sm[4 * kNS];
  const int H = a.H, W = a.H, HW = H * W, V = HW / PX, LP = a.LP;
  const int gl_ = threadIdx.x & (LP - 1);                 // lane within the plane's group
  const int pl = blockIdx.x * (256 / LP) + threadIdx.x / LP;
  const bool pok = pl < a.planes;
  const int plc = pok ? pl : 0;
  const int b = plc / a.C, c = plc - b * a.C;
  const bool has_a = a.ga != nullptr, has_r = a.g_rgb != nullptr;
  const float ma = has_a && a.sa ? a.sa[plc] + 1.f : 1.f;
  const float mr = has_r && a.s_rgb ? a.s_rgb[plc] + 1.f : 1.f;
  const float dd = a.d ? a.d[plc] : 1.f, wn = a.wn[c], bn = a.bn[c];
  float wk[kMaxCr];
#pragma unroll
  for (int k = 0; k < kMaxCr; ++k) wk[k] = (has_r && k < a.Cr) ? a.w_rgb[k * a.C + c] : 0.f;
  const float *op = a.out + (size_t)plc * HW;
  float *gcp = a.gconv + (size_t)plc * HW;
  const float *np = a.nzt + (size_t)b * a.S * a.S;
  const float *grp = has_r ? a.g_rgb + (size_t)b * a.Cr * HW : nullptr;
  const float *gap = has_a ? a.ga + (size_t)plc * HW * (UP ? 4 : 1) : nullptr;
  const int lane = threadIdx.x & 63;
  float red[kNS];
#pragma unroll
  for (int i = 0; i < kNS; ++i) red[i] = 0.f;

  for (int e0 = blockIdx.y * LP; e0 < V; e0 += gridDim.y * LP) {     // (block-uniform bound: every lane shuffles)
    const int e = e0 + gl_;
    const bool live = pok && e < V;
    float t[PX];
#pragma unroll
    for (int q = 0; q < PX; ++q) t[q] = 0.f;
    int p0 = 0;                       // first pixel of this thread's PX pixels (one row: PX divides W)
    if constexpr (UP) {
      const int H2 = 2 * H, W2 = 2 * W, Wh = W >> 1;
      const int k = live ? e / Wh : 0, lp = live ? e - k * Wh : 0, l = 2 * lp;
      p0 = k * W + l;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int Y = 2 * k - 1 + dy;
        const bool yok = live && has_a && Y >= 0 && Y < H2;
        const float *row = gap + (size_t)(yok ? Y : 0) * W2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yok) v = *reinterpret_cast<const f32x4 *>(row + 2 * l);      // columns 2l .. 2l+3
        // column 2l-1 = left neighbour's .w, column 2l+4 = right neighbour's .x (same row and plane: lp > 0 / lp < Wh-1)
        float gl = __shfl_up(v[3], 1, 64), gr = __shfl_down(v[0], 1, 64);
        if (yok && lane == 0 && lp > 0) gl = row[2 * l - 1];
        if (yok && lane == 63 && lp < Wh - 1) gr = row[2 * l + 4];
        if (!yok) continue;
        const float wy = up2_adj_w(dy, k, H);
        float r0 = 0.f, r1 = 0.f;
        if (l > 0) r0 = fmaf(up2_adj_w(0, l, W), gl, r0);
        r0 = fmaf(up2_adj_w(1, l, W), v[0], r0);
        r0 = fmaf(up2_adj_w(2, l, W), v[1], r0);
        r0 = fmaf(up2_adj_w(3, l, W), v[2], r0);
        r1 = fmaf(up2_adj_w(0, l + 1, W), v[1], r1);
        r1 = fmaf(up2_adj_w(1, l + 1, W), v[2], r1);
        r1 = fmaf(up2_adj_w(2, l + 1, W), v[3], r1);
        if (l + 1 < W - 1) r1 = fmaf(up2_adj_w(3, l + 1, W), gr, r1);
        t[0] = fmaf(wy, r0, t[0]);
        t[1] = fmaf(wy, r1, t[1]);
      }
    } else {
      p0 = e * 4;
      if (live && has_a) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(gap + p0);
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = v[q];
      }
    }
    if (!live) continue;
    const int i = p0 / W, j = p0 - i * W;
    float xv[PX], nz[PX], gk[kMaxCr][PX];
    if constexpr (UP) {
      const f32x2 xo = *reinterpret_cast<const f32x2 *>(op + p0);
      const f32x2 n2 = *reinterpret_cast<const f32x2 *>(np + (size_t)i * a.S + j);
      xv[0] = xo[0]; xv[1] = xo[1]; nz[0] = n2[0]; nz[1] = n2[1];
#pragma unroll
      for (int k = 0; k < kMaxCr; ++k) {
        f32x2 g2 = {0.f, 0.f};
        if (has_r && k < a.Cr) g2 = *reinterpret_cast<const f32x2 *>(grp + (size_t)k * HW + p0);
        gk[k][0] = g2[0]; gk[k][1] = g2[1];
      }
    } else {
      const f32x4 xo = *reinterpret_cast<const f32x4 *>(op + p0);
      const f32x4 n4 = *reinterpret_cast<const f32x4 *>(np + (size_t)i * a.S + j);
#pragma unroll
      for (int q = 0; q < 4; ++q) { xv[q] = xo[q]; nz[q] = n4[q]; }
#pragma unroll
      for (int k = 0; k < kMaxCr; ++k) {
        f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
        if (has_r && k < a.Cr) g4 = *reinterpret_cast<const f32x4 *>(grp + (size_t)k * HW + p0);
#pragma unroll
        for (int q = 0; q < 4; ++q) gk[k][q] = g4[q];
      }
    }
    float gc[PX];
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      float tr = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCr; ++k) {
        tr = fmaf(wk[k], gk[k][q], tr);
        red[1 + k] = fmaf(xv[q], gk[k][q], red[1 + k]);
      }
      red[0] = fmaf(xv[q], t[q], red[0]);
      const float G = fmaf(tr, mr, t[q] * ma);
      const float m = G * (xv[q] > 0.f ? 1.f : 0.2f);
      gc[q] = m * dd;
      const float cv = (xv[q] > 0.f ? xv[q] : 5.f * xv[q]) - fmaf(wn, nz[q], bn);     // conv * d, recovered from out
      red[5] = fmaf(m, cv, red[5]);
      red[6] = fmaf(m, nz[q], red[6]);
      red[7] += m;
    }
    if constexpr (UP) {
      *reinterpret_cast<f32x2 *>(gcp + p0) = f32x2{gc[0], gc[1]};
    } else {
      *reinterpret_cast<f32x4 *>(gcp + p0) = f32x4{gc[0], gc[1], gc[2], gc[3]};
    }
  }
  group_sum<kNS>(red, LP, sm);
  if (gl_ == 0 && pok) {
    float *pp = a.part + ((size_t)pl * a.chunks + blockIdx.y) * kNS;
#pragma unroll
    for (int i = 0; i < kNS; ++i) pp[i] = red[i];
  }
}

// One thread per channel c: chunks and samples in fixed order.
//   gs_a[b,c] = S0;  gs_rgb[b,c] = sum_k w_rgb[k,c] S(1+k);  gw_rgb[k,c] = sum_b (s_rgb[b,c] + 1) S(1+k)
//   gd[b,c] = S5 / d[b,c];  gwn[c] = sum_b S6;  gbn[c] = sum_b S7
__global__ __launch_bounds__(256) void k_gstage_bwd_finish(const float *__restrict__ part, const float *__restrict__ w_rgb,
                                                           const float *__restrict__ s_rgb, const float *__restrict__ d,
                                                           float *__restrict__ gs_a, float *__restrict__ gs_rgb,
                                                           float *__restrict__ gw_rgb, float *__restrict__ gd,
                                                           float *__restrict__ gwn, float *__restrict__ gbn, int B, int C,
                                                           int Cr, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float wk[kMaxCr], gw[kMaxCr];
#pragma unroll
  for (int k = 0; k < kMaxCr; ++k) {
    wk[k] = (w_rgb && k < Cr) ? w_rgb[k * C + c] : 0.f;
    gw[k] = 0.f;
  }
  float a_w = 0.f, a_b = 0.f;
  for (int b = 0; b < B; ++b) {
    const int pl = b * C + c;
    float v[kNS];
#pragma unroll
    for (int i = 0; i < kNS; ++i) v[i] = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
      const float *pp = part + ((size_t)pl * chunks + ch) * kNS;
#pragma unroll
      for (int i = 0; i < kNS; ++i) v[i] += pp[i];
    }
    if (gs_a) gs_a[pl] = v[0];
    if (w_rgb) {
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCr; ++k) r = fmaf(wk[k], v[1 + k], r);
      if (gs_rgb) gs_rgb[pl] = r;
      const float m = s_rgb ? s_rgb[pl] + 1.f : 1.f;
#pragma unroll
      for (int k = 0; k < kMaxCr; ++k) gw[k] = fmaf(m, v[1 + k], gw[k]);
    }
    if (gd) gd[pl] = d ? v[5] / d[pl] : v[5];
    a_w += v[6];
    a_b += v[7];
  }
  if (gw_rgb) {
#pragma unroll
    for (int k = 0; k < kMaxCr; ++k)
      if (k < Cr) gw_rgb[k * C + c] = gw[k];
  }
  gwn[c] = a_w;
  gbn[c] = a_b;
}

struct Geom {
  int LP, PB, chunks, yblocks;
};
inline Geom geom(long long planes, int H, int up) {
  const int PX = up ? 2 : 4;
  const long long V = (long long)H * H / PX;
  Geom g;
  g.LP = 4;
  while (g.LP < 64 && g.LP < V) g.LP <<= 1;
  if (V > 64) g.LP = 256;          // (groups are sub-wave or the whole block: group_sum)
  g.PB = 256 / g.LP;
  g.yblocks = (int)((planes + g.PB - 1) / g.PB);
  g.chunks = 1;
  if (g.LP == 256) {      // large maps: split a plane so that the launch has ~2048 blocks, each with >= 4 passes
    static const long long target = [] { const char *e = getenv("HG_GSTAGE_BLOCKS"); return e && atoll(e) > 0 ? atoll(e) : 2048LL; }();
    long long c = (target + planes - 1) / planes;
    const long long cmax = (V + 1023) / 1024;
    if (c > cmax) c = cmax;
    if (c > 64) c = 64;
    if (c < 1) c = 1;
    g.chunks = (int)c;
  }
  return g;
}

}  // namespace

extern "C" {

size_t hg_gstage_bwd_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t up) {
  if (B <= 0 || C <= 0 || H <= 0) return 0;
  const Geom g = geom((long long)B * C, H, up);
  return (size_t)B * C * g.chunks * kNS * sizeof(float);
}

int hg_gstage_bwd(const float *out, const float *ga, const float *sa, int32_t up, const float *g_rgb, const float *w_rgb,
                  const float *s_rgb, int32_t Cr, const float *d, const float *nzt, const float *wn, const float *bn, int32_t S,
                  float *gconv, float *gs_a, float *gs_rgb, float *gw_rgb, float *gd, float *gwn, float *gbn, int32_t B,
                  int32_t C, int32_t H, void *workspace, size_t workspace_bytes, void *stream) {
  if (!out || !nzt || !wn || !bn || !gconv || !gwn || !gbn || B <= 0 || C <= 0 || H <= 0 || S < H) return HG_EINVAL;
  if (!ga && !g_rgb) return HG_EINVAL;                       // no upstream gradient at all
  if (g_rgb && (!w_rgb || !gw_rgb || Cr <= 0)) return HG_EINVAL;
  if ((gd != nullptr) && !d) return HG_EINVAL;
  if (ga && (gs_a != nullptr) != (sa != nullptr)) return HG_EINVAL;
  if (g_rgb && (gs_rgb != nullptr) != (s_rgb != nullptr)) return HG_EINVAL;
  if (g_rgb && Cr > kMaxCr) return HG_EUNSUPPORTED;
  if ((H & 3) || (S & 3) || (up && !ga)) return HG_EUNSUPPORTED;     // 16-byte rows (every HistoGAN map: powers of two >= 4)
  if ((long long)B * C * H * H * (up ? 4 : 1) >= (1ll << 31)) return HG_EUNSUPPORTED;
  const long long planes = (long long)B * C;
  const Geom g = geom(planes, H, up);
  if (!workspace || workspace_bytes < (size_t)planes * g.chunks * kNS * sizeof(float)) return HG_EWORKSPACE;
  GStageArgs a;
  a.out = out; a.ga = ga; a.sa = sa; a.g_rgb = g_rgb; a.w_rgb = w_rgb; a.s_rgb = s_rgb; a.d = d; a.nzt = nzt; a.wn = wn;
  a.bn = bn; a.gconv = gconv; a.part = (float *)workspace;
  a.planes = (int)planes; a.C = C; a.H = H; a.S = S; a.Cr = g_rgb ? Cr : 0; a.LP = g.LP; a.chunks = g.chunks;
  hipStream_t st = (hipStream_t)stream;
  if (up)
    hipLaunchKernelGGL((k_gstage_bwd<2>), dim3((unsigned)g.yblocks, (unsigned)g.chunks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_gstage_bwd<0>), dim3((unsigned)g.yblocks, (unsigned)g.chunks), dim3(256), 0, st, a);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gstage_bwd_finish, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, (const float *)workspace,
                     g_rgb ? w_rgb : nullptr, g_rgb ? s_rgb : nullptr, d, ga ? gs_a : nullptr, g_rgb ? gs_rgb : nullptr,
                     g_rgb ? gw_rgb : nullptr, gd, gwn, gbn, B, C, g_rgb ? Cr : 0, g.chunks);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
