// hg_torgb.hip -- the generator's to-RGB path as two HBM-bound kernels (include/hg_nets.h: hg_torgb_fwd / hg_torgb_bwd).
//
// RGBBlock.forward (histoGAN/histoGAN.py:380-390): a 1x1 modulated convolution WITHOUT demodulation from O = 32 ... 2048
// channels to C = 3 (4 with transparency), plus the running RGB image of the previous block:
//     rgb[b,c,p] = sum_o w[c,o] (s[b,o] + 1) x[b,o,p]  +  prev[b,c,p]
// As a modulated convolution it was three launches forward (a modulated copy of x, a 3-row matrix launch at 0.6 ... 7 TFLOP/s,
// the residual add) and three backward (data gradient, weight gradient, modulation adjoint): x is read or written seven
// times.  The arithmetic is 6 flops per loaded float -- this is a stream over x:
//   forward   reads x once, writes C channels;
//   backward  reads x and the C-channel gradient once, writes gx, and leaves per-(sample, channel) partial sums of
//             x . t (style gradient) and g_c . x (weight gradient) that a finish kernel combines in fixed order.
// Thread layout (both): a block is PG pixel quads (16-byte loads, lanes along the pixels: coalesced) x CG channel groups
// (256 / PG); small maps get their parallelism from the channel groups (4x4: 4 quads x 64 groups).
#include <cstdlib>
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_nets.h"

namespace {

constexpr int kMaxC = 4;

struct RgbGeom {
  int PG, CG, chunks;   // pixel quads per block (power of two, 4 ... 64), channel groups (256 / PG), pixel chunks per image
};
// Pixel quads per block: 64 on the large maps (1 KB per wave load); fewer on the small ones so that B x chunks still gives a
// few hundred blocks (8 quads = one 128-byte line per channel row), the parallelism moving to the channel groups.
inline RgbGeom rgb_geom(int B, int HW) {
  const int q = HW / 4;
  RgbGeom g;
  g.PG = 64;
  while (g.PG > 8 && (long long)B * (q / g.PG) < 256) g.PG >>= 1;
  while (g.PG > 1 && g.PG > q) g.PG >>= 1;
  g.CG = 256 / g.PG;
  g.chunks = (q + g.PG - 1) / g.PG;
  return g;
}
constexpr int kBwdBlocksPerImage = 8;   // pixel blocks per image of the adjoint (each loops over its share of the chunks)

// out[b,c,p] = sum_o wm[c][o] x[b,o,p] + prev[b,c,p],  wm[c][o] = w[c][o] (s[b,o] + 1) staged in LDS
__global__ __launch_bounds__(256) void k_torgb_fwd(const float *__restrict__ x, const float *__restrict__ s,
                                                   const float *__restrict__ w, const float *__restrict__ prev,
                                                   float *__restrict__ out, int O, int C, int HW, int PG, int CG) {
  extern __shared__ float sm[];          // wm [C][O], then the channel-group partials [CG][PG][C] float4
  const int b = blockIdx.y, tid = threadIdx.x;
  const int pg = tid % PG, cg = tid / PG;
  const int q4 = HW / 4, pq = blockIdx.x * PG + pg;
  for (int e = tid; e < C * O; e += 256) {
    const int o = e % O;
    sm[e] = w[e] * (s ? s[b * O + o] + 1.f : 1.f);
  }
  __syncthreads();
  f32x4 acc[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (pq < q4) {
    const f32x4 *xp = reinterpret_cast<const f32x4 *>(x + (size_t)b * O * HW) + pq;
    int o = cg;
    for (; o + 3 * CG < O; o += 4 * CG) {      // four independent loads in flight
      const f32x4 v0 = xp[(size_t)o * q4], v1 = xp[(size_t)(o + CG) * q4], v2 = xp[(size_t)(o + 2 * CG) * q4],
                  v3 = xp[(size_t)(o + 3 * CG) * q4];
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) {
          acc[c] += sm[c * O + o] * v0;
          acc[c] += sm[c * O + o + CG] * v1;
          acc[c] += sm[c * O + o + 2 * CG] * v2;
          acc[c] += sm[c * O + o + 3 * CG] * v3;
        }
    }
    for (; o < O; o += CG) {
      const f32x4 v0 = xp[(size_t)o * q4];
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) acc[c] += sm[c * O + o] * v0;
    }
  }
  // combine the channel groups in fixed order (deterministic)
  f32x4 *part = reinterpret_cast<f32x4 *>(sm + ((C * O + 3) & ~3));
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kMaxC; ++c)
    if (c < C) part[(cg * PG + pg) * C + c] = acc[c];
  __syncthreads();
  for (int e = tid; e < PG * C; e += 256) {
    const int p2 = e / C, c = e % C, pq2 = blockIdx.x * PG + p2;
    if (pq2 >= q4) continue;
    f32x4 v = part[p2 * C + c];
    for (int g = 1; g < CG; ++g) v += part[(g * PG + p2) * C + c];
    const size_t off = ((size_t)b * C + c) * HW + (size_t)pq2 * 4;
    if (prev) v += *reinterpret_cast<const f32x4 *>(prev + off);
    *reinterpret_cast<f32x4 *>(out + off) = v;
  }
}

// t[b,o,p] = sum_c w[c,o] g[b,c,p];  gx = (s + 1) t;  per (pixel block, channel) partial sums of x . t and g_c . x:
// part[((b * nblk + blk) * O + o) * (1 + C) + {0, 1 + c}].  grid = (nblk pixel blocks, B, channel splits): nothing is summed
// over channels here, so the channels split freely over blocks; a thread walks the pixel quads of its block for one channel
// at a time (g re-read from L1 / L2: 3 channels against x's O) and the PG lanes of a channel group meet ONCE per channel.
__global__ __launch_bounds__(256) void k_torgb_bwd(const float *__restrict__ g, const float *__restrict__ x,
                                                   const float *__restrict__ s, const float *__restrict__ w,
                                                   float *__restrict__ gx, float *__restrict__ part, int O, int C, int HW,
                                                   int PG, int CG, int nblk, int o_per) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int pg = tid % PG, cg = tid / PG;
  const int q4 = HW / 4;
  const int o_begin = blockIdx.z * o_per, o_end = o_begin + o_per < O ? o_begin + o_per : O;
  const f32x4 *gq = reinterpret_cast<const f32x4 *>(g + (size_t)b * C * HW);
  const f32x4 *xq = reinterpret_cast<const f32x4 *>(x + (size_t)b * O * HW);
  f32x4 *gxq = reinterpret_cast<f32x4 *>(gx + (size_t)b * O * HW);
  float *pp = part + ((size_t)(b * nblk + blockIdx.x) * O) * (1 + C);
  for (int o = o_begin + cg; o < o_end; o += CG) {   // (the PG lanes of a channel group share o: uniform for the shuffles)
    float wc[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) wc[c] = c < C ? w[c * O + o] : 0.f;
    const float m = s ? s[b * O + o] + 1.f : 1.f;
    float red[1 + kMaxC];
#pragma unroll
    for (int r = 0; r < 1 + kMaxC; ++r) red[r] = 0.f;
    for (int pq = blockIdx.x * PG + pg; pq < q4; pq += nblk * PG) {
      const f32x4 xv = xq[(size_t)o * q4 + pq];
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) {
          const f32x4 gc = gq[(size_t)c * q4 + pq];
          t += wc[c] * gc;
          const f32x4 mm = gc * xv;
          red[1 + c] += (mm[0] + mm[1]) + (mm[2] + mm[3]);
        }
      gxq[(size_t)o * q4 + pq] = m * t;
      const f32x4 mm = xv * t;
      red[0] += (mm[0] + mm[1]) + (mm[2] + mm[3]);
    }
#pragma unroll
    for (int r = 0; r < 1 + kMaxC; ++r)
      if (r < 1 + C) {
        float v = red[r];
        for (int off = PG >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        red[r] = v;
      }
    if (pg == 0) {
#pragma unroll
      for (int r = 0; r < 1 + kMaxC; ++r)
        if (r < 1 + C) pp[(size_t)o * (1 + C) + r] = red[r];
    }
  }
}

// gs[b,o] = sum_blk part[..][0];  gw[c,o] = sum_b (s[b,o] + 1) sum_blk part[..][1 + c].  One block per channel o, one thread
// per sample (fixed order over the pixel blocks, then over the samples through LDS: deterministic).
__global__ __launch_bounds__(256) void k_torgb_bwd_finish(const float *__restrict__ part, const float *__restrict__ s,
                                                          float *__restrict__ gs, float *__restrict__ gw, int B, int O, int C,
                                                          int nblk) {
  __shared__ float acc[256][kMaxC];
  const int o = blockIdx.x, tid = threadIdx.x;
  float wsum[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) wsum[c] = 0.f;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + tid;
    float a0 = 0.f, ac[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) ac[c] = 0.f;
    if (b < B) {
      for (int k = 0; k < nblk; ++k) {
        const float *pp = part + ((size_t)(b * nblk + k) * O + o) * (1 + C);
        a0 += pp[0];
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) ac[c] += pp[1 + c];
      }
      if (gs) gs[b * O + o] = a0;
      const float m = s ? s[b * O + o] + 1.f : 1.f;
#pragma unroll
      for (int c = 0; c < kMaxC; ++c) ac[c] *= m;
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[tid][c] = ac[c];
    __syncthreads();
    if (tid < kMaxC && tid < C) {
      float v = wsum[0];      // (thread c keeps the running sum of channel c in wsum[0])
      const int nb = B - b0 < 256 ? B - b0 : 256;
      for (int i = 0; i < nb; ++i) v += acc[i][tid];
      wsum[0] = v;
    }
    __syncthreads();
  }
  if (tid < kMaxC && tid < C) gw[tid * O + o] = wsum[0];
}

inline bool rgb_ok(int B, int O, int C, int HW) { return B > 0 && O > 0 && C > 0 && C <= kMaxC && HW > 0 && (HW & 3) == 0; }

}  // namespace

extern "C" {

int hg_torgb_fwd(const float *x, const float *s, const float *w, const float *prev, float *out, int32_t B, int32_t O,
                 int32_t C, int32_t HW, void *stream) {
  if (!x || !w || !out || B <= 0 || O <= 0 || C <= 0 || HW <= 0) return HG_EINVAL;
  if (!rgb_ok(B, O, C, HW)) return HG_EUNSUPPORTED;
  const RgbGeom g = rgb_geom(B, HW);
  const size_t lds = ((size_t)((C * O + 3) & ~3) + (size_t)256 * C * 4) * sizeof(float);
  if (lds > 64 * 1024) return HG_EUNSUPPORTED;
  hipLaunchKernelGGL(k_torgb_fwd, dim3((unsigned)g.chunks, (unsigned)B), dim3(256), lds, (hipStream_t)stream, x, s, w, prev, out,
                     O, C, HW, g.PG, g.CG);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

static inline int bwd_blocks(const RgbGeom &g) { return g.chunks < kBwdBlocksPerImage ? g.chunks : kBwdBlocksPerImage; }

size_t hg_torgb_bwd_workspace_bytes(int32_t B, int32_t O, int32_t C, int32_t HW) {
  if (!rgb_ok(B, O, C, HW)) return 0;
  return (size_t)B * bwd_blocks(rgb_geom(B, HW)) * O * (1 + C) * sizeof(float);
}

int hg_torgb_bwd(const float *g, const float *x, const float *s, const float *w, float *gx, float *gs, float *gw, int32_t B,
                 int32_t O, int32_t C, int32_t HW, void *workspace, size_t workspace_bytes, void *stream) {
  if (!g || !x || !w || !gx || !gw || !workspace || B <= 0 || O <= 0 || C <= 0 || HW <= 0) return HG_EINVAL;
  if (!rgb_ok(B, O, C, HW)) return HG_EUNSUPPORTED;
  if ((gs != nullptr) != (s != nullptr)) return HG_EINVAL;
  if (workspace_bytes < hg_torgb_bwd_workspace_bytes(B, O, C, HW)) return HG_EWORKSPACE;
  const RgbGeom gm = rgb_geom(B, HW);
  const int nblk = bwd_blocks(gm);
  // channel splits: enough blocks to fill the chip (~1024), each split a multiple of the channel groups
  int osplit = (1024 + B * nblk - 1) / (B * nblk);
  const int maxs = (O + gm.CG - 1) / gm.CG;
  if (osplit > maxs) osplit = maxs;
  if (osplit < 1) osplit = 1;
  int o_per = (O + osplit - 1) / osplit;
  o_per = (o_per + gm.CG - 1) / gm.CG * gm.CG;
  osplit = (O + o_per - 1) / o_per;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_torgb_bwd, dim3((unsigned)nblk, (unsigned)B, (unsigned)osplit), dim3(256), 0, st, g, x, s, w, gx,
                     (float *)workspace, O, C, HW, gm.PG, gm.CG, nblk, o_per);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_torgb_bwd_finish, dim3((unsigned)O), dim3(256), 0, st, (const float *)workspace, s, gs, gw, B, O, C, nblk);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
