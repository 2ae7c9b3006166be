// hg_conv.hip -- fp32 MFMA implicit-GEMM convolutions (3x3 / 1x1, stride 1 or 2, padding k/2) for gfx950.
//
// The reference runs every convolution of the generator as ONE grouped F.conv2d over per-sample
// weights (histoGAN/histoGAN.py:420-440) and the discriminator's as nn.Conv2d (:510-518).  Here the
// contraction is a hand-written implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain,
// 157.3 TFLOP/s peak) with the modulation / demodulation / bias fused as input / output scales:
//
//   k_conv   D[channel][pixel] = sum_{tap,k} Wt[tap][k][channel] * X[k][pixel*IS + tap]
//            A operand = packed weights (LDS, [tap][k][channel], channel contiguous -> conflict-free),
//            B operand = a halo tile of the input (LDS, [k][image][rows][cols]); the taps are shifted
//            reads of the same halo tile, so the input is fetched once per k-chunk, not 9x.
//            Output rows (channels) x columns (pixels): a lane owns one pixel, 32 consecutive lanes
//            write 32 consecutive pixels of one channel (128 B segments).
//            The same kernel computes the data gradient: stride 1 = the forward kernel on transposed +
//            flipped weights; stride 2 = four launches, one per output parity class, each with the
//            1/2/2/4 taps that reach that class (no multiplications by inserted zeros).
//   k_wgrad  dW[tap][n][k] = sum_pixels gout[n][pixel] * X[k][pixel*IS + tap]   (K-dim = pixels, split-K
//            over pixel chunks into slabs, then k_wgrad_reduce sums the slabs in fixed order and
//            writes the (N,K,kh,kw) layout).
//   k_pack   W (Co,Ci,kh,kw) -> Wt[tap][K][N] (LDS-tiled transpose).
//
// Pixel tiles are NI images x TH rows x TW columns (all powers of two, chosen on the host per layer:
// 32-wide rows for big maps, several whole images per tile for 4x4 / 8x8 maps).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_conv.h"

// tuning knobs (experiment builds: HG_CFLAGS='-DHG_CONV_MINW=3' python -m histogan_amd.build, see build.py)
#ifndef HG_CONV_MINW
#define HG_CONV_MINW 2   // __launch_bounds__ minimum waves per SIMD of k_conv (register budget 512 / MINW)
#endif
#ifndef HG_CONV_SETPRIO
#define HG_CONV_SETPRIO 0
#endif
#ifndef HG_CONV_OPIPE
#define HG_CONV_OPIPE 1  // explicit one-step-ahead operand pipeline in the MFMA loop of k_conv (+2.5 % on the generator layers)
#endif
#ifndef HG_CONV_KC
#define HG_CONV_KC 4     // input channels per K chunk of the stride-1 tiles (64x64 tile: twice that); 2/4/8 measure within 3 %
#endif
#ifndef HG_WGRAD_TS_DBUF
#define HG_WGRAD_TS_DBUF 0   // double-buffering the tap-split pixel-split tiles: measured no gain
#endif
#ifndef HG_WGRAD_TAPSPLIT
#define HG_WGRAD_TAPSPLIT 2   // k_wgrad: the three kernel rows of a tile on three waves (1: 2x2-tile blocks only, 2: all 3x3 tiles)
#endif
#ifndef HG_WGRAD_PC128
#define HG_WGRAD_PC128 1   // 128-pixel chunks for the pixel-split 3x3 weight-gradient tiles (<= 32 channels on one side)
#endif
// (Rounds 3-4 built the K-split combination INSIDE k_conv three times and removed it each time: with device-scope fences
// 53 instead of 38 ms of convolutions per step (a release / acquire pair writes back / invalidates a whole L2 on this
// multi-XCD part); fence-free with a tile's splits on one XCD 48.5 ms per plain step against 46.1 (the finishers run
// alone at the tail of the launch); with the splits adjacent in dispatch order 50.9 (profiles/r04_xcd_splitk.json).  The
// two-launch form -- slabs + k_splitk_reduce -- stays.  The code is in the history: commits 547ecd1, 19f73b6.)
#ifndef HG_CONV_BIGTILE_SPLITK
#define HG_CONV_BIGTILE_SPLITK 2   // 128x128 tile + K split for 8x8 maps (1) and 4x4 maps (2)
#endif

namespace {

struct Geom {
  int lTW, lTH, lNI;          // log2 of tile width / height / images per tile
  int TWp, IMS, HALO, CHS;    // halo row length, floats per image halo, NI*IMS, LDS channel stride
  float inv_TWp, inv_IMS, inv_HALO;
  int tiles_x, tiles_y, groups;
  int lo_y, lo_x;             // input coordinate of halo element (0,0) = tile_origin*IS + lo
};

struct ConvArgs {
  const float *in, *wt;
  float *out;
  const float *iscale, *oscale, *bias;
  const float *addend;   // optional, out layout: out = conv + bias + addend (the discriminator block's residual sum)
  int B, K, N, Kp, Np;
  int Hi, Wi;      // input image
  int Ho, Wo;      // output image
  int Hc, Wc;      // compute grid: pixel (y,x) reads input (y*IS + dy, x*IS + dx), writes (y*os + oy, x*os + ox)
  int os, oy, ox;
  int toff[9];     // LDS offset of tap t inside the halo tile
  int ntx, wrow0, wrow_dy, wrow_dx;  // packed-weight row of tap t = wrow0 + (t / ntx)*wrow_dy + (t % ntx)*wrow_dx
  int ksplit;      // > 1: blockIdx.z handles a K range and writes raw partial sums to slab[z] (out layout)
  float *slab;
  // fused generator epilogue (hg_modconv2d_fwd): v = acc*oscale + bias[n] + noise_w[n]*noise_img[b][y][x]; lrelu
  const float *noise_w, *noise_img;
  int noise_S;     // noise_img is (B, noise_S, noise_S)
  float slope;     // > 0: LeakyReLU slope applied last; 0: none
  Geom g;
};

__device__ __forceinline__ int fdiv(int e, float inv) {  // floor(e / d) for 0 <= e < 2^20, d < 2^12 (see host)
  return (int)(((float)e + 0.5f) * inv);
}

// Buffer (SRSRC) loads for the operand staging: a 32-bit byte offset per lane against a wave-uniform base, and the
// hardware range check turns an offset of 0xFFFFFFFF into a load of 0.0 without touching memory -- zero padding,
// batch tails and lanes without an element cost no predicate, no branch and no s_waitcnt behind the load.
constexpr unsigned kOOB = 0xFFFFFFFFu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, unsigned bytes = kOOB) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
// a wave-uniform pointer the compiler may have parked in vector registers, back in scalar ones (a buffer descriptor built
// from vector registers gets every load wrapped in a readfirstlane "waterfall" loop: the iscale loads of the fused-extras
// kernel were)
__device__ __forceinline__ const float *uniform_ptr(const float *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float *)(((unsigned long long)hi << 32) | lo);
}
constexpr unsigned kFar = 0x80000000u;   // an out-of-range offset that stays out of range when < 2 GB is added to it
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}

// ------------------------------------------------------------------------------------------------
// output / data-gradient kernel
// MT = MFMA tile: 32 (v_mfma_f32_32x32x2_f32, 2 channels per instruction) or 16 (v_mfma_f32_16x16x4_f32, 4 channels
// per instruction, same flop rate) -- the 16-wide tile serves layers with <= 16 output channels (the first
// discriminator block, the to-RGB convolutions) without multiplying 16 empty rows.
template <int MT>
struct MfmaTile {
  typedef f32x16 acc_t;
  static constexpr int NR = 16, KS = 2;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
};
template <>
struct MfmaTile<16> {
  typedef f32x4 acc_t;
  static constexpr int NR = 4, KS = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int r, int lk) { return 4 * lk + r; }
};

// FE (fused extras) = input scale / output scale / noise / LeakyReLU support; the plain instantiation (bias only)
// keeps ~90 fewer registers and is what the training hot path runs.
template <int WC, int WP, int TC, int TP, int TAPS, int KC, int IS, bool SM, int MT, bool FE>
__device__ __forceinline__ void conv_body(const ConvArgs &a, const int bidx, const int bidy, const int bidz) {
  typedef MfmaTile<MT> M;
  constexpr int KS = M::KS;         // channels per MFMA
  static_assert(KC % KS == 0, "chunk must hold whole MFMA k-steps");
  constexpr int NT = WC * WP * 64;
  constexpr int NB = WC * TC * MT;  // channels per block
  constexpr int MB = WP * TP * MT;  // pixels per block
  constexpr int WPT = KC * NB / 4;  // float4 per tap of a weight chunk
  constexpr int WTOT = TAPS * WPT;
  constexpr int NW = (WTOT + NT - 1) / NT;
  // Upper bound (in 1/16 per pixel) of the halo size over the geometries the host builds for this tile shape:
  // 256-pixel tiles are only used with rows >= 16 wide (32x8 / 16x16: <= 1.33), 128-pixel tiles with rows >= 8
  // wide (<= 1.6), 64-pixel tiles with anything down to 4x4 (2.25) or, with SM, 2x2 maps (4.0); stride 2: 5.08.
  // (stride 2 with SM: 2x2 output tiles, a 5x5 halo per 4 pixels = 6.25)
  constexpr int R16 = IS == 2 ? (SM ? 100 : 83) : (TAPS == 1 ? 16 : (SM ? 64 : (MB == 256 ? 22 : (MB == 128 ? 26 : 36))));
  constexpr int HMAX = KC * MB * R16 / 16;
  constexpr int NH = (HMAX + NT - 1) / NT;

  extern __shared__ float smem[];
  float *Ws = smem;                  // [TAPS][KC][NB]
  float *Xs = smem + TAPS * KC * NB; // [KC][CHS]

  const Geom &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane % MT, lk = lane / MT;   // position inside the MFMA tile / k index of the operand
  const int wc = wave % WC, wp = wave / WC;
  const int Hi = a.Hi, Wi = a.Wi, K = a.K, N = a.N;

  int pt = bidx;
  const int tx = pt % g.tiles_x;
  pt /= g.tiles_x;
  const int ty = pt % g.tiles_y;
  const int grp = pt / g.tiles_y;
  const int x0 = tx << g.lTW, y0 = ty << g.lTH, b0 = grp << g.lNI;
  const int n0 = bidy * NB;
  const int TWm = (1 << g.lTW) - 1, THm = (1 << g.lTH) - 1;

  // ---- staging descriptors of the halo tile (chunk-invariant): byte offset of the element against this block's first
  //      image (kOOB: zero padding / no element), and the LDS slot it is written to (spare lanes: a dump row behind Xs)
  unsigned voff[NH];
  int xso[NH];
  const int htot = KC * g.HALO;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int e = tid + i * NT;
    voff[i] = kOOB;
    xso[i] = KC * g.CHS + lane;
    if (e < htot) {
      const int kc = fdiv(e, g.inv_HALO);
      const int r = e - kc * g.HALO;
      const int img = fdiv(r, g.inv_IMS);
      const int rr = r - img * g.IMS;
      const int hy = fdiv(rr, g.inv_TWp);
      const int hx = rr - hy * g.TWp;
      const int gy = y0 * IS + g.lo_y + hy, gx = x0 * IS + g.lo_x + hx, b = b0 + img;
      xso[i] = kc * g.CHS + r;
      if (b < a.B && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi)
        voff[i] = (unsigned)(((img * K + kc) * Hi + gy) * Wi + gx) * 4u;
    }
  }
  // weight staging: element id = tid + i*NT -> (tap, float4 in the tap's [KC][NB] slice); byte offset in the packed
  // weights for chunk 0 (the chunk advances through the scalar offset of the load)
  unsigned wv[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int id = tid + i * NT;
    const int t = id / WPT, e4 = id % WPT;
    const int kc = e4 / (NB / 4), c4 = e4 % (NB / 4);
    const int wrow = t < TAPS ? a.wrow0 + (t / a.ntx) * a.wrow_dy + (t % a.ntx) * a.wrow_dx : 0;
    wv[i] = (NW * NT == WTOT || id < WTOT) ? (unsigned)((wrow + kc) * a.Np + n0 + c4 * 4) * 4u : kOOB;
  }
  // LDS slot (in float4) of weight element i: spare lanes of the last pass write a dump row behind the operands
  const int wdump = (TAPS * KC * NB + KC * g.CHS + 64) / 4 + 1 + lane;

  // ---- operand read offsets
  int pixoff[TP];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    const int p = (wp * TP + tp) * MT + lm;
    const int px = p & TWm, py = (p >> g.lTW) & THm, pi = p >> (g.lTW + g.lTH);
    pixoff[tp] = pi * g.IMS + py * IS * g.TWp + px * IS + lk * g.CHS;
  }
  const int aoff = lk * NB + wc * TC * MT + lm;

  typename M::acc_t acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int r = 0; r < M::NR; ++r) acc[i][j][r] = 0.f;

  float xr[NH];
  float xs[FE ? NH : 1];     // modulation scale of each staged element (only live when iscale is given)
  unsigned soff[FE ? NH : 1];     // byte offset of (b*K + kc) of each staged element in iscale (kOOB: none)
  if (FE && a.iscale != nullptr) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int e = tid + i * NT;
      soff[i] = kOOB;
      if (e < htot) {
        const int kc = fdiv(e, g.inv_HALO);
        const int bb = b0 + fdiv(e - kc * g.HALO, g.inv_IMS);
        if (bb < a.B) soff[i] = (unsigned)(bb * K + kc) * 4u;
      }
    }
  }
  f32x4 wr[NW];
  const int nchunks_all = (K + KC - 1) / KC;
  const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;   // chunks per K split
  const int c_begin = bidz * cps;
  const int nchunks = c_begin + cps < nchunks_all ? c_begin + cps : nchunks_all;
  const int HWi = Hi * Wi;
  const float *inblk = a.in + (size_t)b0 * K * HWi;          // first image of this block (wave-uniform)
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wt), rs = make_rsrc(uniform_ptr(FE && a.iscale ? a.iscale : a.wt));

  // All loads are unconditional (a `valid ? load : 0` select makes the compiler branch around every load and wait for
  // it at once: the global latency then runs in series with the MFMAs of the chunk).
  auto prefetch = [&](int c) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(inblk + (size_t)c * KC * HWi);
    if (K - c * KC < KC) {   // the last, partial chunk (once per block): channels past the end read as zero from here on
      const int krem = K - c * KC;
#pragma unroll
      for (int i = 0; i < NH; ++i)
        if (fdiv(tid + i * NT, g.inv_HALO) >= krem) {
          voff[i] = kOOB;
          if constexpr (FE) soff[i] = kOOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) xr[i] = buf_load(rx, voff[i], 0);
    if constexpr (FE) {
      if (a.iscale != nullptr) {
#pragma unroll
        for (int i = 0; i < NH; ++i) xs[i] = buf_load(rs, soff[i], c * KC * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) wr[i] = buf_load4(rw, wv[i], c * KC * a.Np * 4);
  };

  for (int c = c_begin - 1; c < nchunks; ++c) {
    if (c >= c_begin) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NH; ++i) Xs[xso[i]] = (FE && a.iscale != nullptr) ? xr[i] * xs[FE ? i : 0] : xr[i];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int id = tid + i * NT;
        if (NW * NT == WTOT || i + 1 < NW) reinterpret_cast<f32x4 *>(Ws)[id] = wr[i];
        else reinterpret_cast<f32x4 *>(Ws)[id < WTOT ? id : wdump] = wr[i];
      }
      __syncthreads();
    }
    if (c + 1 < nchunks) prefetch(c + 1);
    if (c < c_begin) continue;

#if HG_CONV_OPIPE
    // operand reads one MFMA step ahead (explicit two-slot pipeline): the ds_reads of step s+1 are issued before the
    // MFMAs of step s, so a wave never waits a full LDS latency between two MFMA groups
    constexpr int NS = TAPS * (KC / KS);
    float av[2][TC], bv[2][TP];
    auto ldop = [&](int s_, int slot) __attribute__((always_inline)) {
      const int t = s_ / (KC / KS), kk = s_ % (KC / KS);
#pragma unroll
      for (int i = 0; i < TC; ++i) av[slot][i] = Ws[(t * KC + kk * KS) * NB + aoff + i * MT];
      if constexpr (TAPS == 9) {
        // the 3x3 tap grid: the three taps of a kernel row are consecutive floats of the halo row -- one address per
        // (pixel tile, kernel row, k step) with the column as the instruction's immediate offset (a third of the address
        // registers of the general form below)
        const int roff = a.toff[(t / 3) * 3];
#pragma unroll
        for (int j = 0; j < TP; ++j) bv[slot][j] = (Xs + pixoff[j] + roff + kk * KS * g.CHS)[t % 3];
      } else {
        const int toff = a.toff[t];
#pragma unroll
        for (int j = 0; j < TP; ++j) bv[slot][j] = Xs[pixoff[j] + toff + kk * KS * g.CHS];
      }
    };
    ldop(0, 0);
    // pin the schedule: [operand reads of step s+1] [MFMAs of step s] (the compiler otherwise batches the pixel-operand
    // reads of several steps and issues the weight reads just in time, behind an s_waitcnt lgkmcnt(0) per step)
    constexpr int DSN = (MT == 32 ? (TC + 1) / 2 : TC) + TP;
    __builtin_amdgcn_sched_group_barrier(0x100, DSN, 0);
#if HG_CONV_SETPRIO > 0
    __builtin_amdgcn_s_setprio(HG_CONV_SETPRIO);   // waves in their MFMA phase go first: the staging of other waves fills in
#elif HG_CONV_SETPRIO < 0
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (s_ + 1 < NS) ldop(s_ + 1, (s_ + 1) & 1);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = M::mma(av[s_ & 1][i], bv[s_ & 1][j], acc[i][j]);
      if (s_ + 1 < NS) {
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);     // address of the next weight read
        __builtin_amdgcn_sched_group_barrier(0x100, DSN, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TC * TP, 0);
    }
#if HG_CONV_SETPRIO > 0
    __builtin_amdgcn_s_setprio(0);
#elif HG_CONV_SETPRIO < 0
    __builtin_amdgcn_s_setprio(-(HG_CONV_SETPRIO));   // staging / barrier phases go first, so waves return to their MFMAs sooner
#endif
#else
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int toff = a.toff[t];
#pragma unroll
      for (int kk = 0; kk < KC / KS; ++kk) {
        float av[TC], bv[TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) av[i] = Ws[(t * KC + kk * KS) * NB + aoff + i * MT];
#pragma unroll
        for (int j = 0; j < TP; ++j) bv[j] = Xs[pixoff[j] + toff + kk * KS * g.CHS];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) acc[i][j] = M::mma(av[i], bv[j], acc[i][j]);
      }
    }
#endif
  }

  // ---- epilogue: D[i = channel][j = pixel]; 32x32: row(i) = (r&3) + 8*(r>>2) + 4*(lane>>5), col(j) = lane&31;
  //      16x16: row = 4*(lane>>4) + r, col = lane&15
  const int HWo = a.Ho * a.Wo;
  // fin: the finished sums (bias / scales / noise / activation applied, written to `out`); else raw partial sums to slab[z]
  auto epilogue = [&](const bool fin) __attribute__((always_inline)) {
  if constexpr (!FE) {
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      const int p = (wp * TP + j) * MT + lm;
      const int px = p & TWm, py = (p >> g.lTW) & THm, pi = p >> (g.lTW + g.lTH);
      const int cx = x0 + px, cy = y0 + py, b = b0 + pi;
      if (b >= a.B || cy >= a.Hc || cx >= a.Wc) continue;
      const size_t pofs = ((size_t)b * N) * HWo + (cy * a.os + a.oy) * a.Wo + cx * a.os + a.ox;
      float *ob = (fin ? a.out : a.slab + (size_t)bidz * a.B * N * HWo) + pofs;
#pragma unroll
      for (int i = 0; i < TC; ++i) {
#pragma unroll
        for (int r = 0; r < M::NR; ++r) {
          const int ch = n0 + (wc * TC + i) * MT + M::row(r, lk);
          if (ch < N) {
            float v = acc[i][j][r];
            if (fin && a.bias) v += a.bias[ch];
            if (fin && a.addend) v += a.addend[pofs + (size_t)ch * HWo];
            ob[(size_t)ch * HWo] = v;
          }
        }
      }
    }
  } else {
    // One channel tile at a time: its 2 x NR per-channel parameters are loaded before that tile's stores (the compiler
    // cannot hoist loads past possibly aliasing stores itself), the demodulation scale per (pixel tile, channel).  Holding
    // the parameters of ALL tiles at once made this epilogue the register peak of the kernel (238 VGPRs = 2 blocks per
    // CU); now the K loop is, as in the plain instantiation.
    const float *__restrict__ pbias_ = a.bias, *__restrict__ pnw_ = a.noise_w, *__restrict__ posc_ = a.oscale;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      float pbias[M::NR], pnw[M::NR];
#pragma unroll
      for (int r = 0; r < M::NR; ++r) {
        const int ch = n0 + (wc * TC + i) * MT + M::row(r, lk);
        const int cc = ch < N ? ch : N - 1;
        pbias[r] = (fin && pbias_) ? pbias_[cc] : 0.f;
        pnw[r] = (fin && a.noise_img) ? pnw_[cc] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const int p = (wp * TP + j) * MT + lm;
        const int px = p & TWm, py = (p >> g.lTW) & THm, pi = p >> (g.lTW + g.lTH);
        const int cx = x0 + px, cy = y0 + py, b = b0 + pi;
        if (b >= a.B || cy >= a.Hc || cx >= a.Wc) continue;
        const size_t pofs = ((size_t)b * N) * HWo + (cy * a.os + a.oy) * a.Wo + cx * a.os + a.ox;
        float *ob = (fin ? a.out : a.slab + (size_t)bidz * a.B * N * HWo) + pofs;
        const float nz = (a.noise_img != nullptr && fin)
                             ? a.noise_img[((size_t)b * a.noise_S + cy * a.os + a.oy) * a.noise_S + cx * a.os + a.ox] : 0.f;
        float posc[M::NR];
#pragma unroll
        for (int r = 0; r < M::NR; ++r) {
          const int ch = n0 + (wc * TC + i) * MT + M::row(r, lk);
          posc[r] = (fin && posc_) ? posc_[b * N + (ch < N ? ch : N - 1)] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < M::NR; ++r) {
          const int ch = n0 + (wc * TC + i) * MT + M::row(r, lk);
          if (ch < N) {
            float v = fmaf(acc[i][j][r], posc[r], fmaf(pnw[r], nz, pbias[r]));
            if (fin && a.slope > 0.f) v = v > 0.f ? v : a.slope * v;
            ob[(size_t)ch * HWo] = v;
          }
        }
      }
    }
  }
  };
  // split-K partials get their epilogue in k_splitk_reduce.  (Round 3 built the combination into this kernel -- the last-
  // arriving block of a tile summing the slabs behind arrival flags -- and measured it 15 ms per step SLOWER: the
  // device-scope release / acquire fences it needs write back and invalidate the XCD's whole L2, per block, under the
  // other blocks' operand reuse.  DESIGN.md section 8; the fence-free form below is section 12.5's experiment.)
  epilogue(a.ksplit == 1);
}

template <int WC, int WP, int TC, int TP, int TAPS, int KC, int IS, bool SM, int MT, bool FE>
__global__ __launch_bounds__(WC *WP * 64, HG_CONV_MINW) void k_conv(const ConvArgs a) {
  conv_body<WC, WP, TC, TP, TAPS, KC, IS, SM, MT, FE>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The stride-2 data gradient's four output-parity classes (1 / 2 / 2 / 4 taps: hg_conv2d_dgrad) as ONE launch: block
// (4 t + c) is tile t of class c.  For the small maps of the discriminator's deep blocks the four launches were a few
// dozen blocks and ~20 us each (launch latency, not work); one launch fills the chip four times better.
struct ConvArgs4 {
  ConvArgs c[4];
  int tiles[4];    // pixel tiles (grid x extent) of each class
  int xcd_map;     // block id -> (tile, class) mapping, see k_conv_parity4
};
// Block -> (tile, class): workgroups go to the 8 XCDs round robin by linear id, and the two classes of one row parity
// write the even / odd floats of the same cache lines.  With id = 32 q + 8 c + x the four classes of tile 8 q + x run on
// XCD x within 32 ids of each other, so the half-written lines of one class meet the other half in that XCD's L2 and
// leave it as whole lines (the classes on different XCDs -- or in separate launches -- send masked partial lines to
// memory: 1.6 TB/s on the 256^2 maps of the first discriminator block, 2.26 TB/s paired; tools/s2_dgrad_probe.py).
template <int WC, int WP, int TC, int TP, int KC, bool SM, int MT, bool FE>
__global__ __launch_bounds__(WC *WP * 64, HG_CONV_MINW) void k_conv_parity4(const ConvArgs4 a) {
  const int cls = a.xcd_map ? (blockIdx.x >> 3) & 3 : blockIdx.x & 3;
  const int t = a.xcd_map ? (blockIdx.x >> 5) * 8 + (blockIdx.x & 7) : blockIdx.x >> 2;
  if (t >= a.tiles[cls]) return;
  if (cls == 0) conv_body<WC, WP, TC, TP, 1, KC, 1, SM, MT, FE>(a.c[0], t, blockIdx.y, blockIdx.z);
  else if (cls == 1) conv_body<WC, WP, TC, TP, 2, KC, 1, SM, MT, FE>(a.c[1], t, blockIdx.y, blockIdx.z);
  else if (cls == 2) conv_body<WC, WP, TC, TP, 2, KC, 1, SM, MT, FE>(a.c[2], t, blockIdx.y, blockIdx.z);
  else conv_body<WC, WP, TC, TP, 4, KC, 1, SM, MT, FE>(a.c[3], t, blockIdx.y, blockIdx.z);
}

// out[b][n][p] = epilogue( sum_z slab[z][b][n][p] )   (fixed order: deterministic); epilogue as in k_conv
__global__ __launch_bounds__(256) void k_splitk_reduce(const float *__restrict__ slab, float *__restrict__ out,
                                                       const float *__restrict__ oscale, const float *__restrict__ bias,
                                                       const float *__restrict__ noise_w, const float *__restrict__ noise_img,
                                                       const float *__restrict__ addend, int noise_S, float slope,
                                                       long long total, int HWo, int Wo, int N, int ksplit) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < ksplit; ++z) v += slab[(size_t)z * total + i];
    const long long bn = i / HWo;
    const int n = (int)(bn % N);
    float add = bias ? bias[n] : 0.f;
    if (noise_img) {
      const int p = (int)(i - bn * HWo), y = p / Wo, x = p - y * Wo;
      add = fmaf(noise_w[n], noise_img[((size_t)(bn / N) * noise_S + y) * noise_S + x], add);
    }
    v = oscale ? fmaf(v, oscale[bn], add) : v + add;
    if (addend) v += addend[i];
    if (slope > 0.f) v = v > 0.f ? v : slope * v;
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
struct WgradArgs {
  const float *in, *gout;
  float *slab;  // [splits*WS][TAPS][Np32][Kp32]
  const float *iscale, *gscale;
  int B, K, N, Hi, Wi, Ho, Wo, Kp32, Np32;
  int tiles_x, tiles_y, nchunks, splits, ktiles;
  float *gw;  // non-NULL: a single slab would be written -> store straight into gw (N,K,taps) instead
};

constexpr int WG_TP = 32 * 9 + 4;   // LDS row pitch of the single-slab store transpose (k_wgrad)

// compile-time pixel-chunk geometry of the weight-gradient kernel: PC (output) pixels = NI images x TH x TW
template <int PC, int LTW, int PAD, int IS>
struct CGeom {
  static constexpr int TW = 1 << LTW;
  static constexpr int TH = (PC / TW) < TW ? (PC / TW) : TW;
  static constexpr int NI = PC / (TW * TH);
  static constexpr int TWp = (TW - 1) * IS + 1 + 2 * PAD, THp = (TH - 1) * IS + 1 + 2 * PAD;
  static constexpr int IMS = TWp * THp, HALO = NI * IMS, CHS = HALO | 1;  // odd pitch: lanes vary the channel
};

// Block = WN x WK x WS waves.  A wave owns a 32(n) x 32(k) x TAPS accumulator tile (TAPS*16 registers); the WS
// waves of a tile split the pixel pairs of each chunk between them and write separate slabs.  The next
// chunk is fetched into registers while the MFMAs of the current one run (the kernel is allowed the full
// 512-register budget: accumulators in AGPRs, staging in VGPRs).
// TS = 3 (3x3 only): the three kernel ROWS of a tile go to three different waves (3 x 16 accumulator registers each,
// 12 waves per 2x2-tile block).  The 9-tap tile leaves room for ONE wave per SIMD (144 accumulators + staging), so every
// s_waitcnt / barrier / staging instruction of that wave idles the matrix pipe (measured MFMA utilisation 0.62); with
// three lighter waves per SIMD another wave's MFMAs fill those gaps.
// SC: operand scales (iscale / gscale of the fused modulated-convolution backward) -- its own instantiation: the scale
// lookups cost the unscaled kernel 190 spilled SGPRs (v_readlane / v_writelane in the chunk loop) when they shared one.
template <int WN, int WK, int WS, int TAPS, int PC, int LTW, int IS, int MT, int TS = 1, bool SC = false>
__global__ __launch_bounds__(WN *WK *WS *TS * 64) void k_wgrad(const WgradArgs a) {
  typedef MfmaTile<MT> M;
  static_assert(TS == 1 || (TS == 3 && TAPS == 9), "tap split: rows of the 3x3 kernel");
  constexpr int TPW = TAPS / TS;   // taps per wave
  constexpr int KS = M::KS;     // pixels per MFMA
  constexpr int NT = WN * WK * WS * TS * 64;
  constexpr int NBW = WN * MT;  // out channels (gout) per block
  constexpr int KBW = WK * MT;  // in channels per block
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  using G = CGeom<PC, LTW, PAD, IS>;
  constexpr int GP = PC + 1;  // odd pitch

  extern __shared__ float smem[];
  float *Gs = smem;             // [NBW][GP]
  float *Xs = smem + NBW * GP;  // [KBW][CHS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane % MT, lk = lane / MT;
  static_assert((1 << LTW) >= KS, "the pixels of one MFMA k-step must lie in one tile row");
  const int wn = wave % WN, wk = (wave / WN) % WK, ws = (wave / (WN * WK)) % WS, wt = wave / (WN * WK * WS);
  const int t0 = wt * TPW;   // first tap of this wave
  const int Hi = a.Hi, Wi = a.Wi, Ho = a.Ho, Wo = a.Wo, K = a.K, N = a.N;
  const int HWi = Hi * Wi, HWo = Ho * Wo;
  const int k0 = (blockIdx.x % a.ktiles) * KBW, n0 = (blockIdx.x / a.ktiles) * NBW;

  typename M::acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < M::NR; ++r) acc[t][r] = 0.f;

  const float *Ga = Gs + (wn * MT + lm) * GP + lk;
  const float *Xa = Xs + (wk * MT + lm) * G::CHS + lk * IS;

  // Staging maps: a pass moves CPI whole channels; thread -> (channel slot cs, position r) is fixed, so the
  // per-chunk address work is ONE offset per thread and each element costs one load + one LDS store.
  constexpr int GCPI = NT / PC, NGI = (NBW + GCPI - 1) / GCPI;            // gout: PC pixels per channel
  constexpr int HCPI = NT / G::HALO, NHI = (KBW + HCPI - 1) / HCPI;      // halo: HALO floats per channel
  static_assert(HCPI >= 1, "halo tile wider than the block");
  const int gcs = tid / PC, gp = tid % PC;
  const int gpx = gp % G::TW, gpy = (gp / G::TW) % G::TH, gpi = gp / (G::TW * G::TH);
  const int hcs = tid / G::HALO, hrr = tid % G::HALO;
  const int himg = hrr / G::IMS, hy = (hrr % G::IMS) / G::TWp, hx = hrr % G::TWp;
  const bool hlane = hcs < HCPI;

  // Prefetch loads are UNCONDITIONAL buffer loads (a `valid ? load : 0` select puts an s_waitcnt right behind every
  // load and serialises the global latency with the MFMAs -- with one block per CU nothing else hides it): a position
  // outside the image, a channel past the end or a spare lane carries the byte offset kOOB, for which the hardware
  // range check returns 0.0 without a memory access -- no clamped coordinates, no select at the LDS store.  The base of
  // the descriptor is the chunk's first image and this block's first channel (wave-uniform), the channel of pass i
  // goes through the scalar offset; per chunk a thread computes ONE offset for its gout pixel and ONE for its halo
  // position.
  unsigned gstat = 0;
  unsigned long long hstat = 0;
  static_assert(NGI <= 32 && NHI <= 64, "validity masks");
#pragma unroll
  for (int i = 0; i < NGI; ++i)
    if (n0 + gcs + i * GCPI < N && (NGI * GCPI == NBW || gcs + i * GCPI < NBW)) gstat |= 1u << i;
#pragma unroll
  for (int i = 0; i < NHI; ++i)
    if (hlane && k0 + hcs + i * HCPI < K && (NHI * HCPI == KBW || hcs + i * HCPI < KBW)) hstat |= 1ull << i;

  float gr[NGI], hr[NHI];
  unsigned gmask = 0;
  unsigned long long hmask = 0;
  int pb0 = 0;   // image-group origin of the prefetched chunk (for the scale lookups at the store)
  auto prefetch = [&](int chunk) __attribute__((always_inline)) {
    int pt = chunk;
    const int tx = pt % a.tiles_x;
    pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int grp = pt / a.tiles_y;
    const int x0 = tx * G::TW, y0 = ty * G::TH, b0 = grp * G::NI;
    pb0 = b0;
    if constexpr (G::NI == 1) {
      // one image per chunk: the byte offset grows with the channel, so the descriptor's size ends the valid channels
      // (offset < (N - n0) * HWo * 4  <=>  channel < N) and the passes just step the offset -- no per-pass masks
      const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gout + ((size_t)b0 * N + n0) * HWo, (unsigned)(N - n0) * (unsigned)HWo * 4u);
      const int gx = x0 + gpx, gy = y0 + gpy;
      const bool ok = gy < Ho && gx < Wo;
      unsigned off = ok ? (unsigned)(gcs * HWo + gy * Wo + gx) * 4u : kFar;
      if constexpr (SC) gmask = ok ? gstat : 0u;
      const unsigned gstep = (unsigned)(GCPI * HWo) * 4u;
#pragma unroll
      for (int i = 0; i < NGI; ++i) {
        gr[i] = buf_load(rg, off, 0);
        off += gstep;
      }
      const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.in + ((size_t)b0 * K + k0) * HWi, (unsigned)(K - k0) * (unsigned)HWi * 4u);
      const int hgy = y0 * IS + hy - PAD, hgx = x0 * IS + hx - PAD;
      const bool hok = hlane && (unsigned)hgy < (unsigned)Hi && (unsigned)hgx < (unsigned)Wi;
      unsigned hoff = hok ? (unsigned)(hcs * HWi + hgy * Wi + hgx) * 4u : kFar;
      if constexpr (SC) hmask = hok ? hstat : 0ull;
      const unsigned hstep = (unsigned)(HCPI * HWi) * 4u;
#pragma unroll
      for (int i = 0; i < NHI; ++i) {
        hr[i] = buf_load(rh, hoff, 0);
        hoff += hstep;
      }
    } else {
      {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gout + ((size_t)b0 * N + n0) * HWo);
        const int gx = x0 + gpx, gy = y0 + gpy, b = b0 + gpi;
        const bool ok = b < a.B && gy < Ho && gx < Wo;
        const unsigned off = ok ? (unsigned)((gpi * N + gcs) * HWo + gy * Wo + gx) * 4u : kOOB;
        gmask = ok ? gstat : 0u;
#pragma unroll
        for (int i = 0; i < NGI; ++i) gr[i] = buf_load(rg, (gmask >> i) & 1u ? off + (unsigned)(i * GCPI * HWo) * 4u : kOOB, 0);
      }
      {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.in + ((size_t)b0 * K + k0) * HWi);
        const int gy = y0 * IS + hy - PAD, gx = x0 * IS + hx - PAD, b = b0 + himg;
        const bool ok = b < a.B && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi;
        const unsigned off = ok ? (unsigned)((himg * K + hcs) * HWi + gy * Wi + gx) * 4u : kOOB;
        hmask = ok ? hstat : 0ull;
#pragma unroll
        for (int i = 0; i < NHI; ++i) hr[i] = buf_load(rh, (hmask >> i) & 1ull ? off + (unsigned)(i * HCPI * HWi) * 4u : kOOB, 0);
      }
    }
  };

  // Double-buffered LDS (when two buffers fit in 160 KB), ONE barrier per chunk: while chunk c is multiplied out of
  // buffer (it&1), the registers holding chunk c+1 are stored into the other buffer and re-filled with chunk c+2.
  constexpr int BUFSZ = NBW * GP + KBW * G::CHS;   // floats of one (gout + halo) buffer
  // (measured: +3..5 % for the 4-wave tiles; the pixel-split tiles (WS > 1) are faster single-buffered)
  constexpr int NBUF = ((WS == 1 || (TS > 1 && HG_WGRAD_TS_DBUF)) && 2 * BUFSZ * 4 <= 160 * 1024) ? 2 : 1;
  auto store = [&](int buf) __attribute__((always_inline)) {
    float *G2 = smem + buf * BUFSZ, *X2 = G2 + NBW * GP;
    if constexpr (!SC) {
      // the training hot path: registers straight to LDS (invalid positions were loaded as 0.0); only the last pass of
      // each operand can address a channel row outside the block tile
#pragma unroll
      for (int i = 0; i < NGI; ++i)
        if (NGI * GCPI == NBW || i + 1 < NGI || gcs + i * GCPI < NBW) G2[(gcs + i * GCPI) * GP + gp] = gr[i];
      if (hlane) {
#pragma unroll
        for (int i = 0; i < NHI; ++i)
          if (NHI * HCPI == KBW || i + 1 < NHI || hcs + i * HCPI < KBW) X2[(hcs + i * HCPI) * G::CHS + hrr] = hr[i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NGI; ++i)
      if (NGI * GCPI == NBW || gcs + i * GCPI < NBW) {
        float v = gr[i];   // invalid positions were loaded as 0
        if (a.gscale != nullptr && ((gmask >> i) & 1u)) v *= a.gscale[(pb0 + gpi) * N + n0 + gcs + i * GCPI];
        G2[(gcs + i * GCPI) * GP + gp] = v;
      }
#pragma unroll
    for (int i = 0; i < NHI; ++i)
      if (hlane && (NHI * HCPI == KBW || hcs + i * HCPI < KBW)) {
        float v = hr[i];
        if (a.iscale != nullptr) {
          if constexpr (G::NI == 1) {
            // one image per chunk: the HCPI candidate scales of pass i are wave-uniform -> scalar loads + select
            const float *sp = a.iscale + (size_t)pb0 * K + k0;
            float sc = 1.f;
#pragma unroll
            for (int c = 0; c < HCPI; ++c) {
              const int kk = i * HCPI + c;
              const float sv = sp[k0 + kk < K ? kk : K - 1 - k0];
              sc = hcs == c ? sv : sc;
            }
            v *= sc;
          } else if ((hmask >> i) & 1ull) {
            v *= a.iscale[(pb0 + himg) * K + k0 + hcs + i * HCPI];
          }
        }
        X2[(hcs + i * HCPI) * G::CHS + hrr] = v;
      }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float *Gc = Ga + buf * BUFSZ, *Xc = Xa + buf * BUFSZ;
#pragma unroll
    for (int q = 0; q < PC / KS / WS; ++q) {
      // this wave's pixel group (KS pixels of one row): compile-time when WS == 1, else one of WS alternatives
      float av;
      float bv[TPW];
      auto rd = [&](int p0) __attribute__((always_inline)) {
        const int hoff = (p0 / (G::TW * G::TH)) * G::IMS + ((p0 / G::TW) % G::TH) * IS * G::TWp + (p0 % G::TW) * IS;
        av = Gc[p0];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          if constexpr (TS == 1) bv[t] = Xc[hoff + (PAD ? (t / 3) * G::TWp + (t % 3) : 0)];
          else bv[t] = Xc[hoff + wt * G::TWp + t];   // kernel row wt, column t
        }
      };
      if constexpr (WS == 1) rd(q * KS);
      else rd((q * WS + ws) * KS);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = M::mma(av, bv[t], acc[t]);
    }
  };

  const int c0 = blockIdx.y, cs = a.splits;
  if constexpr (NBUF == 2) {
    if (c0 < a.nchunks) {
      prefetch(c0);
      store(0);
      if (c0 + cs < a.nchunks) prefetch(c0 + cs);
    }
    __syncthreads();
    int it = 0;
    for (int chunk = c0; chunk < a.nchunks; chunk += cs, ++it) {
      if (chunk + cs < a.nchunks) {
        store((it & 1) ^ 1);
        if (chunk + 2 * cs < a.nchunks) prefetch(chunk + 2 * cs);
      }
      compute(it & 1);
      __syncthreads();
    }
  } else {
    if (c0 < a.nchunks) prefetch(c0);
    for (int chunk = c0; chunk < a.nchunks; chunk += cs) {
      __syncthreads();
      store(0);
      __syncthreads();
      if (chunk + cs < a.nchunks) prefetch(chunk + cs);
      compute(0);
    }
  }

  // The WS waves of a tile each hold a partial sum over their share of the pixels: combine them in LDS (fixed order
  // ws = 0, 1, ..) so that the block leaves ONE slab -- half / a quarter of the slab traffic of the pixel-split tiles,
  // written and read back by k_wgrad_reduce (37.7 MB -> 18.9 / 9.4 MB per launch on the 256^2 / 128^2 layers).
  if constexpr (WS > 1) {
    constexpr int NTW = WN * WK * TS;                  // wave tiles of the block (per pixel split)
    constexpr int WSZ = TPW * M::NR * 64;              // floats one wave holds
    const int wtile = wn + WN * (wk + WK * wt);
    __syncthreads();                                   // operand buffers are dead
    if (ws > 0) {
      float *R = smem + ((size_t)(ws - 1) * NTW + wtile) * WSZ + lane;
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < M::NR; ++r) R[(t * M::NR + r) * 64] = acc[t][r];
    }
    __syncthreads();
    if (ws > 0) return;
#pragma unroll
    for (int w2 = 1; w2 < WS; ++w2) {
      const float *R = smem + ((size_t)(w2 - 1) * NTW + wtile) * WSZ + lane;
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < M::NR; ++r) acc[t][r] += R[(t * M::NR + r) * 64];
    }
  }

  if (a.gw != nullptr) {  // one split: this block's tile IS the result
    if constexpr (WS == 1 && MT == 32 && TAPS == 9) {
      if ((K & 3) == 0) {
        // The (n, k, tap) layout makes a wave's 32n x 32k x 9 tile 32 contiguous 1152-byte runs: transpose it through
        // LDS and store 16-byte pieces (the direct store below writes 4-byte pieces at a 36-byte stride: 0.5 TB/s on
        // the 151 MB gradient of a 2048x2048x3x3 layer).  The launch reserved 4 x 32 x WG_TP floats for this.
        __syncthreads();   // operand buffers are dead
        float *T = smem + (wave % (WN * WK)) * 32 * WG_TP;   // the TS waves of a tile fill one transpose buffer
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int r = 0; r < M::NR; ++r) T[M::row(r, lk) * WG_TP + lm * 9 + t0 + t] = acc[t][r];
        __syncthreads();
        const int kbase = k0 + wk * 32;
        const int nk = K - kbase < 32 ? K - kbase : 32;          // valid k's (multiple of 4)
        const int nf4 = nk > 0 ? nk * 9 / 4 : 0;                 // 16-byte pieces per row
        for (int row = wt; row < 32; row += TS) {
          const int n = n0 + wn * 32 + row;
          if (n >= N) break;
          float *dst = a.gw + ((size_t)n * K + kbase) * 9;
          for (int f = lane; f < nf4; f += 64)
            *reinterpret_cast<f32x4 *>(dst + 4 * f) = *reinterpret_cast<const f32x4 *>(T + row * WG_TP + 4 * f);
        }
        return;
      }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < M::NR; ++r) {
        const int n = n0 + wn * MT + M::row(r, lk);
        const int k = k0 + wk * MT + lm;
        if (n < N && k < K) a.gw[((size_t)n * K + k) * TAPS + t0 + t] = acc[t][r];
      }
    return;
  }
  // slab[split][t][n][k]: D[i = n][j = k]
  float *sb = a.slab + (size_t)blockIdx.y * TAPS * a.Np32 * a.Kp32;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < M::NR; ++r) {
      const int n = n0 + wn * MT + M::row(r, lk);
      const int k = k0 + wk * MT + lm;
      sb[((size_t)(t0 + t) * a.Np32 + n) * a.Kp32 + k] = acc[t][r];
    }
}

// gw[n][k][t] = sum_s slab[s][t][n][k].  One block per (n, tap, 32 k's): 8 lanes x 16-byte loads along k, 32 groups of
// splits with four independent loads in flight each (the slabs of the 256^2 / 128^2 layers are 10-20 MB per launch: the
// kernel is a bandwidth problem -- 4-byte loads in 8 groups ran at 1 TB/s), fixed-order combine through LDS (deterministic).
template <int TAPS>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ slab, float *__restrict__ gw, int N, int K,
                                                      int Np32, int Kp32, int splits) {
  __shared__ float part[32][33];
  const int k4 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int kb = blockIdx.x * 32 + k4 * 4, t = blockIdx.y, n = blockIdx.z;
  const size_t sstride = (size_t)TAPS * Np32 * Kp32;
  const bool in = kb < Kp32;                                   // Kp32 is a multiple of the tile (16 / 32), kb of 4
  const float *p = slab + ((size_t)t * Np32 + n) * Kp32 + (in ? kb : 0);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (in) {
    int sp = grp;
    for (; sp + 96 < splits; sp += 128) {
      s0 += *reinterpret_cast<const f32x4 *>(p + (size_t)sp * sstride);
      s1 += *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 32) * sstride);
      s2 += *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 64) * sstride);
      s3 += *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 96) * sstride);
    }
    for (; sp < splits; sp += 32) s0 += *reinterpret_cast<const f32x4 *>(p + (size_t)sp * sstride);
  }
  const f32x4 v4 = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int c = 0; c < 4; ++c) part[grp][k4 * 4 + c] = v4[c];
  __syncthreads();
  const int kx = threadIdx.x, k = blockIdx.x * 32 + kx;
  if (kx < 32 && k < K) {
    float v = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 32; ++g2) v += part[g2][kx];
    gw[((size_t)n * K + k) * TAPS + t] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: W (Co,Ci,T) -> Wt[T][Kp][Np]; one block = 32 co x 32 ci x T, LDS transpose
template <int TAPS>
__global__ __launch_bounds__(256) void k_pack(const float *__restrict__ w, float *__restrict__ wt, int Co, int Ci, int Kp,
                                              int Np, int mode) {
  constexpr int RW = 32 * TAPS;  // floats per co row of the tile
  __shared__ float tile[32][RW + 1];
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  for (int e = threadIdx.x; e < 32 * RW; e += 256) {
    const int i = e / RW, q = e % RW;  // q = j*TAPS + t
    const int co = co0 + i, ci = ci0 + q / TAPS;
    tile[i][q] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci0) * TAPS + q] : 0.f;
  }
  __syncthreads();
  // output element (t, a, b) with b fastest: fwd: a = ci (K), b = co (N); dgrad: a = co (K), b = ci (N)
  for (int e = threadIdx.x; e < 32 * RW; e += 256) {
    const int b = e & 31, a_ = (e >> 5) & 31, t = e >> 10;
    int kk, nn;
    float v;
    if (mode == HG_CONV_PACK_FWD) {
      kk = ci0 + a_; nn = co0 + b;
      v = tile[b][a_ * TAPS + t];
    } else {
      kk = co0 + a_; nn = ci0 + b;
      v = tile[a_][b * TAPS + (TAPS - 1 - t)];
    }
    if (kk < Kp && nn < Np) wt[((size_t)t * Kp + kk) * Np + nn] = v;
  }
}

// both packings from ONE read of the weights (the training step needs the forward and the data-gradient operand of
// every convolution once per optimizer step)
// one 32 (co) x 32 (ci) x TAPS tile of both packed operands; `tile_mem`: 32 * (32 * TAPS + 1) floats of LDS
template <int TAPS>
__device__ __forceinline__ void pack_both_tile(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wd,
                                               int Co, int Ci, int Kpf, int Npf, int Kpd, int Npd, int bx, int by,
                                               float *tile_mem, float *__restrict__ wsq = nullptr) {
  constexpr int RW = 32 * TAPS;
  float(*tile)[RW + 1] = reinterpret_cast<float(*)[RW + 1]>(tile_mem);
  const int ci0 = bx * 32, co0 = by * 32;
  if (co0 >= Co || ci0 >= Ci) {   // padding-only tile: zeros, no loads
    for (int e = threadIdx.x; e < 32 * RW; e += 256) {
      const int b = e & 31, a_ = (e >> 5) & 31, t = e >> 10;
      if (ci0 + a_ < Kpf && co0 + b < Npf) wf[((size_t)t * Kpf + ci0 + a_) * Npf + co0 + b] = 0.f;
      if (co0 + a_ < Kpd && ci0 + b < Npd) wd[((size_t)t * Kpd + co0 + a_) * Npd + ci0 + b] = 0.f;
    }
    return;
  }
  float v[4 * TAPS];   // 32*RW / 256 loads in flight (a rolled loop pays one global latency per iteration)
#pragma unroll
  for (int it = 0; it < 4 * TAPS; ++it) {
    const int e = threadIdx.x + it * 256;
    const int i = e / RW, q = e % RW;
    const int co = co0 + i, ci = ci0 + q / TAPS;
    v[it] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci0) * TAPS + q] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < 4 * TAPS; ++it) {
    const int e = threadIdx.x + it * 256;
    tile[e / RW][e % RW] = v[it];
  }
  __syncthreads();
  if (wsq != nullptr) {   // sum of squares over the taps of every (co, ci) of the tile (fixed tap order)
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int i = e >> 5, j = e & 31;
      if (co0 + i < Co && ci0 + j < Ci) {
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) q = fmaf(tile[i][j * TAPS + t], tile[i][j * TAPS + t], q);
        wsq[(size_t)(co0 + i) * Ci + ci0 + j] = q;
      }
    }
  }
#pragma unroll 4
  for (int e = threadIdx.x; e < 32 * RW; e += 256) {
    const int b = e & 31, a_ = (e >> 5) & 31, t = e >> 10;
    {  // forward: Wt[t][ci][co]
      const int kk = ci0 + a_, nn = co0 + b;
      if (kk < Kpf && nn < Npf) wf[((size_t)t * Kpf + kk) * Npf + nn] = tile[b][a_ * TAPS + t];
    }
    {  // data gradient: Wt[t][co][ci] with flipped taps
      const int kk = co0 + a_, nn = ci0 + b;
      if (kk < Kpd && nn < Npd) wd[((size_t)t * Kpd + kk) * Npd + nn] = tile[a_][b * TAPS + (TAPS - 1 - t)];
    }
  }
}

template <int TAPS>
__global__ __launch_bounds__(256) void k_pack_both(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wd,
                                                   int Co, int Ci, int Kpf, int Npf, int Kpd, int Npd) {
  __shared__ float tile_mem[32 * (32 * TAPS + 1)];
  pack_both_tile<TAPS>(w, wf, wd, Co, Ci, Kpf, Npf, Kpd, Npd, blockIdx.x, blockIdx.y, tile_mem);
}

// Every convolution weight of a model in ONE launch (hg_conv_pack_weights_multi): block -> (item, tile) through the items'
// first-block table.  ~50 launches of 6-80 us per optimizer step become two.
__global__ __launch_bounds__(256) void k_pack_multi(const hg_pack_item *__restrict__ items, int n_items) {
  __shared__ float tile_mem[32 * (32 * 9 + 1)];
  int it = 0;
  while (it + 1 < n_items && (int)blockIdx.x >= items[it + 1].block_begin) ++it;   // wave-uniform scan (n_items ~ 30)
  const hg_pack_item im = items[it];
  const int local = (int)blockIdx.x - im.block_begin;
  const int Kpf = (im.Ci + 15) / 16 * 16, Npf = (im.Co + 127) / 128 * 128, Kpd = (im.Co + 15) / 16 * 16, Npd = (im.Ci + 127) / 128 * 128;
  const int gx = Npd / 32;
  const int bx = local % gx, by = local / gx;
  if (im.ksize == 3) pack_both_tile<9>(im.w, im.wt_fwd, im.wt_dgrad, im.Co, im.Ci, Kpf, Npf, Kpd, Npd, bx, by, tile_mem, im.wsq);
  else pack_both_tile<1>(im.w, im.wt_fwd, im.wt_dgrad, im.Co, im.Ci, Kpf, Npf, Kpd, Npd, bx, by, tile_mem, im.wsq);
}

// ------------------------------------------------------------------------------------------------
inline int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// log2 of the pixel tile (width, height, images) for MB compute-grid pixels per block over an Hc x Wc grid
inline void tile_shape(int MB, int Hc, int Wc, int min_t, int &lTW, int &lTH, int &lNI) {
  lTW = ceil_log2(Wc < min_t ? min_t : Wc);
  if (lTW > 5) lTW = 5;
  const int lMB = ceil_log2(MB);
  if (lTW > lMB - 1) lTW = lMB - 1;
  lTH = ceil_log2(Hc < min_t ? min_t : Hc);
  if (lTH > lMB - lTW) lTH = lMB - lTW;
  lNI = lMB - lTW - lTH;
}
// pixel tiles (grid x extent) of a launch with MB pixels per block
inline long long pixel_tiles(int MB, int B, int Hc, int Wc, int min_t) {
  int lTW, lTH, lNI;
  tile_shape(MB, Hc, Wc, min_t, lTW, lTH, lNI);
  const int TW = 1 << lTW, TH = 1 << lTH, NI = 1 << lNI;
  return (long long)((Wc + TW - 1) / TW) * ((Hc + TH - 1) / TH) * ((B + NI - 1) / NI);
}

// pixel-tile geometry for MB compute-grid pixels per block over an Hc x Wc grid; taps span [lo, hi] in y and x
Geom make_geom(int MB, int B, int Hc, int Wc, int IS, int lo_y, int hi_y, int lo_x, int hi_x, bool odd_chs,
               int min_t = 4) {
  Geom g;
  int lTW, lTH, lNI;
  tile_shape(MB, Hc, Wc, min_t, lTW, lTH, lNI);
  g.lTW = lTW; g.lTH = lTH; g.lNI = lNI;
  const int TW = 1 << lTW, TH = 1 << lTH, NI = 1 << g.lNI;
  g.TWp = (TW - 1) * IS + 1 + (hi_x - lo_x);
  g.IMS = ((TH - 1) * IS + 1 + (hi_y - lo_y)) * g.TWp;
  g.HALO = NI * g.IMS;
  g.CHS = odd_chs ? (g.HALO | 1) : g.HALO;
  g.inv_TWp = 1.0f / (float)g.TWp;
  g.inv_IMS = 1.0f / (float)g.IMS;
  g.inv_HALO = 1.0f / (float)g.HALO;
  g.tiles_x = (Wc + TW - 1) / TW;
  g.tiles_y = (Hc + TH - 1) / TH;
  g.groups = (B + NI - 1) / NI;
  g.lo_y = lo_y; g.lo_x = lo_x;
  return g;
}

// tap list of one launch: offsets (dy, dx) in input coordinates relative to pixel*IS, and the packed-weight tap
struct Taps {
  int n, ntx, dy[9], dx[9], w[9];   // ntx = taps per row of the (rows x ntx) tap grid
};

// tile shape + K split of one k_conv launch
enum ConvTile { TILE_16x256, TILE_32x256, TILE_64x256, TILE_128x128, TILE_128x128_SM, TILE_64x64 };
struct ConvPlan {
  ConvTile tile;
  int ksplit;
};

// How many blocks per CU should a launch of `nwg` equal blocks run with?  The MFMA kernels here are resident-block
// bound: a CU with c blocks in flight sustains e[c] of the matrix peak (measured, tools/occ_probe.py: 0.71 / 0.85 / 0.90
// for 1 / 2 / 3 blocks of the 128x128 tile), and a launch whose block count is not a multiple of (CUs x c) ends in a
// round at low occupancy that the dispatcher also balances badly (1024 blocks at c = 3: 104 TFLOP/s, at c = 2: 134).
constexpr size_t kLdsPerCu = 160 * 1024;
// per-process caches are keyed by the CURRENT device (a process that launches on a second GPU must not plan with the first
// one's CU count, nor skip the dynamic-LDS attribute there)
constexpr int kMaxDev = 16;
inline int cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  return dev;
}
inline int num_cus() {
  static int n[kMaxDev] = {0};
  const int dev = cur_dev();
  if (!n[dev]) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n[dev] = pr.multiProcessorCount;
    if (n[dev] <= 0) n[dev] = 256;
  }
  return n[dev];
}
// *t_out: modelled duration of the launch in units of (one block alone on a CU at the full matrix rate)
inline int pick_blocks_per_cu(long long nwg, int cmax, double *t_out = nullptr) {
  static const int forced = getenv("HG_CONV_OCC") ? atoi(getenv("HG_CONV_OCC")) : 0;   // experiments: fixed cap
  static const double e[9] = {0, 0.71, 0.85, 0.90, 0.92, 0.93, 0.93, 0.93, 0.93};
  if (cmax > 8) cmax = 8;
  const double cus = (double)num_cus();
  int best = cmax;
  double best_t = 1e300;
  for (int c = cmax; c >= 1; --c) {
    if (forced > 0 && c != (forced < cmax ? forced : cmax)) continue;
    const long long per_round = (long long)cus * c;
    const long long full = nwg / per_round, rem = nwg - full * per_round;
    double t = (double)full * c / e[c];
    if (rem > 0) {
      int cr = (int)((rem + (long long)cus - 1) / (long long)cus);
      if (full > 0) cr = 2 * cr < c ? 2 * cr : c;   // freed slots are refilled greedily: the tail lands unevenly
      t += cr / e[cr];
    }
    if (t < best_t * 0.995) { best_t = t; best = c; }
  }
  if (t_out) *t_out = best_t;
  return best;
}

// K split of a 128x128-tile launch with `nb` output tiles and `nch` K chunks: the split whose block count fills whole
// rounds of the chip (256 tiles alone run at 0.71 of the matrix rate, 3 x 256 at 0.90), if that pays for the slab
// traffic of k_splitk_reduce.  Times in seconds for `flops` and `out_bytes` of the whole launch.
inline int pick_ksplit_128(long long nb, int nch, double flops, double out_bytes, int max_split = 16) {
  int best = 1;
  double best_t = 1e300;
  for (int ks = 1; ks <= max_split && nch / ks >= 8; ++ks) {
    double tm;
    pick_blocks_per_cu(nb * ks, 3, &tm);
    // tm blocks-alone-times, each block does 1/ks of the K loop of a (flops / nb) tile, on one of num_cus() CUs
    double t = tm * (flops / nb / ks) / (157.3e12 / num_cus()) + 2e-6 * tm;   // + prologue / epilogue per block round
    if (ks > 1) t += 8e-6 + (ks + 1) * out_bytes / 3e12;
    if (t < best_t * 0.97) { best_t = t; best = ks; }
  }
  return best;
}

// Pick the largest tile that still gives >= ~1.5 blocks per CU.  Wide tiles need wide rows (the staging-register
// bound R16 in k_conv): 256-pixel tiles Wc > 8, 128-pixel tiles Wc > 4.  Launches that cannot fill the chip with
// output tiles (few pixels, many channels: the 2x2 ... 8x8 maps) split the reduction over K into slabs.
ConvPlan plan_conv(int B, int K, int N, int Hc, int Wc, int IS, int os, bool have_ws, bool big_split = true, int taps = 9) {
  const long long pix = (long long)B * Hc * Wc;
  auto blocks = [&](int nb, int mb) { return ((N + nb - 1) / nb) * ((pix + mb - 1) / mb); };
  const bool wide256 = Wc > 8 && Hc > 8, wide128 = Wc > 4 && Hc > 4;
  ConvPlan p;
  p.ksplit = 1;
  if (N <= 16 && wide256) { p.tile = TILE_16x256; return p; }
  if (N <= 32 && wide256) { p.tile = TILE_32x256; return p; }
  if (N <= 64 && wide256 && blocks(64, 256) >= 384) { p.tile = TILE_64x256; return p; }
  if (N > 64 && wide128) {
    const long long nb = blocks(128, 128);
    const int nch = (K + HG_CONV_KC - 1) / HG_CONV_KC;
    const bool may_split = HG_CONV_BIGTILE_SPLITK && big_split && have_ws && os == 1 && IS == 1;
    if (nb >= 256 || (may_split && nb >= 32 && nch >= 32)) {
      p.tile = TILE_128x128;
      if (may_split)
        p.ksplit = pick_ksplit_128(nb, nch, 2.0 * pix * K * N * taps, (double)pix * N * 4);
      return p;
    }
  }
#if HG_CONV_BIGTILE_SPLITK > 1
  // 4x4 maps (8 images per 128-pixel tile): the small-map instantiation of the 128x128 tile (larger halo bound)
  if (N > 64 && Wc == 4 && Hc == 4 && big_split && have_ws && os == 1 && IS == 1) {
    const long long nb = blocks(128, 128);
    const int nch = (K + HG_CONV_KC - 1) / HG_CONV_KC;
    if (nb >= 32 && nch >= 32) {
      int ks = (int)((512 + nb - 1) / nb);
      if (ks > nch / 8) ks = nch / 8;
      if (ks > 16) ks = 16;
      if (ks < 1) ks = 1;
      p.tile = TILE_128x128_SM; p.ksplit = ks; return p;
    }
  }
#endif
  p.tile = TILE_64x64;
  // pixel tiles of the 64x64 shape: images are grouped when the map is smaller than the tile
  const int tw = Wc <= 2 ? 2 : (Wc <= 4 ? 4 : (Wc <= 8 ? 8 : (Wc <= 16 ? 16 : 32)));
  int th = Hc <= 2 ? 2 : (Hc <= 4 ? 4 : (Hc <= 8 ? 8 : (Hc <= 16 ? 16 : 32)));
  if (th > 64 / tw) th = 64 / tw;
  const int ni = 64 / (tw * th);
  const long long nblk = (long long)((Wc + tw - 1) / tw) * ((Hc + th - 1) / th) * ((B + ni - 1) / ni) * ((N + 63) / 64);
  const int kc = IS == 2 ? 8 : 2 * HG_CONV_KC, nchunks = (K + kc - 1) / kc;
  // blocks to aim for: 1024 for the stride-1 launches (2x2 maps, 2048 channels: 103 -> 113 TFLOP/s; the kernel is small enough
  // for 4+ blocks per CU), 512 for the stride-2 forward and the four parity-class launches of the stride-2 data gradient
  // (measured slower with more, shorter blocks)
  const int tgt64 = (big_split && IS == 1) ? 1024 : 512;
  if (have_ws && os == 1 && nblk < 384 && nchunks >= 8) {
    int ks = (int)((tgt64 + nblk - 1) / nblk);
    if (ks > nchunks / 4) ks = nchunks / 4;
    if (ks > 32) ks = 32;
    if (ks > 1) p.ksplit = ks;
  }
  return p;
}

inline int launch_splitk_reduce(const ConvArgs &a, int ksplit, hipStream_t st);

// a launch tag: unique per call within the process, never 0
inline unsigned long long next_conv_tag() {
  static std::atomic<unsigned long long> ctr{0x9E3779B97F4A7C15ull ^ ((unsigned long long)(uintptr_t)&ctr << 17)};
  return ctr.fetch_add(2, std::memory_order_relaxed) | 1ull;
}

// output blocks of a plan over B x Hc x Wc compute pixels (before the K split)
inline long long plan_blocks(const ConvPlan &p, int B, int N, int Hc, int Wc) {
  const long long px = (long long)B * Hc * Wc;
  const int nbt = p.tile == TILE_16x256 ? 16 : p.tile == TILE_32x256 ? 32 : (p.tile == TILE_64x256 || p.tile == TILE_64x64) ? 64 : 128;
  const int mbt = (p.tile == TILE_128x128 || p.tile == TILE_128x128_SM) ? 128 : p.tile == TILE_64x64 ? 64 : 256;
  return ((N + nbt - 1) / nbt) * ((px + mbt - 1) / mbt);
}
// 3x3 stride-1 launches: the 2-channel K-chunk kernels instead of the 4-channel ones?  (see dispatch_conv)
inline bool short_k_chunks(const ConvPlan &p, int B, int N, int Hc, int Wc) {
  if (p.tile == TILE_32x256) return true;
  if (p.tile == TILE_64x256 || (p.tile == TILE_128x128 && p.ksplit == 1)) {
    const long long nwg = plan_blocks(p, B, N, Hc, Wc), cus = num_cus();
    return nwg % (4 * cus) == 0 && nwg % (3 * cus) != 0 && nwg <= 32 * cus;
  }
  return false;
}

// geometry / tap tables of one launch into `a`; returns the dynamic LDS bytes (0: not addressable, see below)
template <int WC, int WP, int TC, int TP, int TAPS, int KC, int IS, bool SM, int MT>
size_t prep_conv(ConvArgs &a, const Taps &tp, int ksplit) {
  constexpr int NB = WC * TC * MT, MB = WP * TP * MT;
  int lo_y = 0, hi_y = 0, lo_x = 0, hi_x = 0;
  for (int t = 0; t < TAPS; ++t) {
    lo_y = tp.dy[t] < lo_y ? tp.dy[t] : lo_y; hi_y = tp.dy[t] > hi_y ? tp.dy[t] : hi_y;
    lo_x = tp.dx[t] < lo_x ? tp.dx[t] : lo_x; hi_x = tp.dx[t] > hi_x ? tp.dx[t] : hi_x;
  }
  a.g = make_geom(MB, a.B, a.Hc, a.Wc, IS, lo_y, hi_y, lo_x, hi_x, false, SM ? 2 : 4);
  for (int t = 0; t < TAPS; ++t) a.toff[t] = (tp.dy[t] - lo_y) * a.g.TWp + (tp.dx[t] - lo_x);
  // the tap lists built below are (rows x ntx) grids, so the packed-weight tap index is affine in (t / ntx, t % ntx)
  a.ntx = tp.ntx;
  a.wrow0 = tp.w[0] * a.Kp;
  a.wrow_dx = tp.ntx > 1 ? (tp.w[1] - tp.w[0]) * a.Kp : 0;
  a.wrow_dy = TAPS > tp.ntx ? (tp.w[tp.ntx] - tp.w[0]) * a.Kp : 0;
  a.ksplit = ksplit;
  // the halo loads address one block's images with 32-bit byte offsets
  if ((long long)(1 << a.g.lNI) * a.K * a.Hi * a.Wi >= (1LL << 30)) return 0;
  // + dump rows for the spare staging lanes: 64 floats (halo) and, 16-byte aligned behind them, 64 float4 (weights)
  return ((size_t)TAPS * KC * NB + (size_t)KC * a.g.CHS + 64 + 8 + 256) * sizeof(float);
}

// blocks per CU for a launch of `nwg` blocks of `kern` (see pick_blocks_per_cu): fewer than the registers allow when that
// makes the block count a whole number of rounds; enforced by asking for more LDS than 1/(c+1) of a CU's.  `state`: the
// kernel's cached {register-limited blocks per CU, "large dynamic LDS allowed" flag}.
inline int fit_blocks_per_cu(const void *kern, int threads, long long nwg, size_t &lds, int state[2], const char *what) {
  if (!state[0]) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, lds) != hipSuccess || n < 1) n = 1;
    state[0] = n;
  }
  const int by_lds = (int)(kLdsPerCu / lds), cm = by_lds < state[0] ? (by_lds > 1 ? by_lds : 1) : state[0];
  const int c = pick_blocks_per_cu(nwg, cm);
  if (c < cm) {
    const size_t need = (size_t)kLdsPerCu / (c + 1) + 512;
    if (lds < need) lds = need;
  }
  static const bool dbg = getenv("HG_CONV_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "%s: %lld blocks, max %d (registers %d) -> %d per CU, lds %zu\n", what, nwg, cm, state[0], c, lds);
  if (lds > 48 * 1024 && !state[1]) {   // dynamic LDS above 48 KB has to be allowed once per kernel
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
    if (e != hipSuccess) return (int)e;
    state[1] = 1;
  }
  return 0;
}

// ws_bytes: size of the scratch behind a.slab (0: unknown -> no in-kernel combination)
template <int WC, int WP, int TC, int TP, int TAPS, int KC, int IS, bool SM = false, int MT = 32>
int launch_conv(ConvArgs a, const Taps &tp, int ksplit, bool reduce, hipStream_t st, size_t ws_bytes = 0) {
  constexpr int NB = WC * TC * MT, NT = WC * WP * 64;
  size_t lds = prep_conv<WC, WP, TC, TP, TAPS, KC, IS, SM, MT>(a, tp, ksplit);
  if (!lds) return HG_EUNSUPPORTED;
  bool inkernel_combine = false;
  (void)ws_bytes;
  const bool fe = a.iscale || a.oscale || a.noise_img || a.slope > 0.f;
  auto kern = fe ? k_conv<WC, WP, TC, TP, TAPS, KC, IS, SM, MT, true> : k_conv<WC, WP, TC, TP, TAPS, KC, IS, SM, MT, false>;
  static int state[2][2] = {{0, 0}, {0, 0}};
  const long long nwg = (long long)a.g.tiles_x * a.g.tiles_y * a.g.groups * ((a.N + NB - 1) / NB) * ksplit;
  char what[96];
  snprintf(what, sizeof what, "k_conv<%d,%d,%d,%d,taps %d,kc %d,is %d,mt %d,fe %d>", WC, WP, TC, TP, TAPS, KC, IS, MT, (int)fe);
  if (int rc = fit_blocks_per_cu((const void *)kern, NT, nwg, lds, state[fe], what)) return rc;
  unsigned gx = (unsigned)(a.g.tiles_x * a.g.tiles_y * a.g.groups);
  if (inkernel_combine) {
    // the splits of a tile (blockIdx.z) share an XCD when gridDim.x * gridDim.y is a multiple of 8 (linear id mod 8).  Pad x
    // only as far as that needs: padding 4 pixel tiles x 16 channel blocks to 8 x 16 left the real blocks on XCDs 0-3 (2x time)
    const unsigned ny = (unsigned)((a.N + NB - 1) / NB);
    unsigned m = 8;
    for (unsigned d = 2; d <= 8; d *= 2) if (ny % d == 0) m = 8 / d;
    gx = (gx + m - 1) / m * m;
  }
  dim3 grid(gx, (unsigned)((a.N + NB - 1) / NB), (unsigned)ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, a);
  HG_LAUNCH_CHECK();
  if (ksplit > 1 && reduce && !inkernel_combine) return launch_splitk_reduce(a, ksplit, st);
  return HG_OK;
}

// the four parity-class launches of a stride-2 data gradient as one (k_conv_parity4); the caller reduces the K-split slabs
template <int WC, int WP, int TC, int TP, int KC, bool SM = false, int MT = 32, bool WITH_FE = true>
int launch_conv_parity4(const ConvArgs (&base)[4], const Taps (&tp)[4], int ksplit, hipStream_t st) {
  constexpr int NB = WC * TC * MT, NT = WC * WP * 64;
  ConvArgs4 a4;
  size_t lds = 0;
  long long tiles_sum = 0;
  int tiles_max = 0;
  for (int c = 0; c < 4; ++c) {
    a4.c[c] = base[c];
    a4.tiles[c] = 0;
    if (base[c].Hc <= 0 || base[c].Wc <= 0) continue;   // empty class (1-pixel-wide image)
    size_t l = c == 0 ? prep_conv<WC, WP, TC, TP, 1, KC, 1, SM, MT>(a4.c[c], tp[c], ksplit)
               : c == 3 ? prep_conv<WC, WP, TC, TP, 4, KC, 1, SM, MT>(a4.c[c], tp[c], ksplit)
                        : prep_conv<WC, WP, TC, TP, 2, KC, 1, SM, MT>(a4.c[c], tp[c], ksplit);
    if (!l) return HG_EUNSUPPORTED;
    lds = l > lds ? l : lds;
    a4.tiles[c] = a4.c[c].g.tiles_x * a4.c[c].g.tiles_y * a4.c[c].g.groups;
    tiles_sum += a4.tiles[c];
    tiles_max = a4.tiles[c] > tiles_max ? a4.tiles[c] : tiles_max;
  }
  if (!tiles_max) return HG_OK;
  const ConvArgs &a = base[0];
  const bool fe = a.iscale || a.oscale || a.noise_img || a.slope > 0.f;
  if (fe && !WITH_FE) return HG_EUNSUPPORTED;
  auto kern = k_conv_parity4<WC, WP, TC, TP, KC, SM, MT, false>;
  if constexpr (WITH_FE) { if (fe) kern = k_conv_parity4<WC, WP, TC, TP, KC, SM, MT, true>; }
  static int state[2][2] = {{0, 0}, {0, 0}};
  const int ny = (a.N + NB - 1) / NB;
  if (int rc = fit_blocks_per_cu((const void *)kern, NT, tiles_sum * ny * ksplit, lds, state[fe], "k_conv_parity4")) return rc;
  // (only where the tiles alone cover the XCDs: 4 tiles x 16 channel blocks of a 4x4 map would use 4 of the 8 -- 94 -> 163 us)
  static const bool xcd_env = !(getenv("HG_PARITY_XCD") && atoi(getenv("HG_PARITY_XCD")) == 0);
  const bool xcd_map = xcd_env && tiles_max >= 64;
  a4.xcd_map = xcd_map;
  const unsigned gx = xcd_map ? (unsigned)((tiles_max + 7) / 8 * 32) : (unsigned)(4 * tiles_max);
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)ny, (unsigned)ksplit), dim3(NT), lds, st, a4);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

// K-split scratch: ksplit slabs in `out` layout, then (256-byte aligned) one 64-bit flag per (output tile, split) for the
// in-kernel combination.  Hc x Wc: the compute grid of the launch (== Ho x Wo for stride-1 / forward launches).
inline size_t conv_slab_bytes(const ConvPlan &p, int B, int N, int Ho, int Wo) {
  return p.ksplit > 1 ? (size_t)p.ksplit * B * N * Ho * Wo * sizeof(float) : 0;
}
inline size_t conv_ws_bytes(const ConvPlan &p, int B, int N, int Ho, int Wo) { return conv_slab_bytes(p, B, N, Ho, Wo); }


inline int launch_splitk_reduce(const ConvArgs &a, int ksplit, hipStream_t st) {
  const long long total = (long long)a.B * a.N * a.Ho * a.Wo;
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)nb), dim3(256), 0, st, a.slab, a.out, a.oscale, a.bias, a.noise_w,
                     a.noise_img, a.addend, a.noise_S, a.slope, total, a.Ho * a.Wo, a.Wo, a.N, ksplit);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

// force_ksplit > 0: the caller fixed the K split (and reduces the slabs itself); the 64x64 tile is used
template <int TAPS, int IS>
int dispatch_conv(ConvArgs a, const Taps &tp, void *ws, size_t ws_bytes, hipStream_t st, int force_ksplit = 0) {
  constexpr int KC = IS == 2 ? 4 : HG_CONV_KC;
  a.slab = (float *)ws;
  if (force_ksplit > 0) return launch_conv<2, 2, 1, 1, TAPS, 2 * KC, IS, true>(a, tp, force_ksplit, false, st);
  ConvPlan p = plan_conv(a.B, a.K, a.N, a.Hc, a.Wc, IS, a.os, ws != nullptr, true, TAPS);
  if (conv_slab_bytes(p, a.B, a.N, a.Ho, a.Wo) > ws_bytes) p.ksplit = 1;   // too little scratch: no K split
  // 3x3 stride-1 launches also exist with 2-channel K chunks: 16 fewer staging registers = one more block per CU (4
  // instead of 3).  Taken when that makes the launch whole rounds (1024 / 2048 blocks: 134 -> 139 TFLOP/s) and for the
  // 32-channel tile (+2..5 %); deep-K layers lose 3 % to the doubled barrier count and keep the 4-channel chunks.
  if constexpr (TAPS == 9 && IS == 1 && KC == 4) {
    // (the fused-extras instantiations of the two larger tiles fit 3 blocks per CU at either chunk size -- their epilogue
    // is the register peak -- so they keep the 4-channel chunks)
    const bool fe = a.iscale || a.oscale || a.noise_img || a.slope > 0.f;
    if (short_k_chunks(p, a.B, a.N, a.Hc, a.Wc) && (!fe || p.tile == TILE_32x256)) {
      if (p.tile == TILE_32x256) return launch_conv<1, 4, 1, 2, TAPS, 2, IS>(a, tp, 1, true, st);
      if (p.tile == TILE_64x256) return launch_conv<1, 4, 2, 2, TAPS, 2, IS>(a, tp, 1, true, st);
      return launch_conv<2, 2, 2, 2, TAPS, 2, IS>(a, tp, 1, true, st);
    }
  }
  switch (p.tile) {
    case TILE_16x256:   // 16 ch x 256 px on the 16x16x4 MFMA
      return launch_conv<1, 4, 1, 4, TAPS, 4, IS, false, 16>(a, tp, 1, true, st);
    case TILE_32x256: return launch_conv<1, 4, 1, 2, TAPS, KC, IS>(a, tp, 1, true, st);
    case TILE_64x256: return launch_conv<1, 4, 2, 2, TAPS, KC, IS>(a, tp, 1, true, st);
    case TILE_128x128: return launch_conv<2, 2, 2, 2, TAPS, KC, IS>(a, tp, p.ksplit, true, st, ws_bytes);
    case TILE_128x128_SM:
      if constexpr (IS == 1) return launch_conv<2, 2, 2, 2, TAPS, KC, IS, true>(a, tp, p.ksplit, true, st, ws_bytes);
      else return HG_EUNSUPPORTED;
    default: return launch_conv<2, 2, 1, 1, TAPS, 2 * KC, IS, true>(a, tp, p.ksplit, true, st, ws_bytes);
  }
}

struct WgradPlan {
  int PC, lTW, tiles_x, tiles_y, groups;
  int MT;          // MFMA tile: 16 when both channel counts are <= 16, else 32
  int WN, WK, WS;  // waves per block along n / k / pixel split
  int nchunks, splits, ktiles, ntiles, Kp32, Np32;
  size_t slab_bytes;
};

inline int out_size(int in, int stride) { return (in - 1) / stride + 1; }  // k = 3, pad 1 (or k = 1, pad 0, stride 1)

// pixels per chunk of k_wgrad.  128 where the per-chunk overhead (staging pass, barrier) has the least MFMA work to hide
// behind: the 16x16 tile (4x less work per pixel) and the pixel-split 32-channel tiles of the 3x3 kernel (each of the WS
// waves of a tile only multiplies 1/WS of a chunk) -- on rows at least 8 / 16 wide, where the halo still fits one pass.
constexpr int wgrad_chunk_pixels(int stride, int MT, int lTW, int WS, int taps) {
  return stride == 2 ? 32 : ((MT == 16 && lTW >= 3) || (HG_WGRAD_PC128 && MT == 32 && WS > 1 && taps == 9 && lTW >= 4)) ? 128 : 64;
}

WgradPlan make_wgrad_plan(int B, int K, int N, int Hi, int Wi, int ksize, int stride) {
  WgradPlan p;
  const int Ho = out_size(Hi, stride), Wo = out_size(Wi, stride);
  p.MT = (N <= 16 && K <= 16) ? 16 : 32;
  int lTW = ceil_log2(Wo < 2 ? 2 : Wo);
  if (p.MT == 16 && lTW < 2) lTW = 2;   // 4 pixels per 16x16x4 MFMA must share a row
  if (lTW > 5) lTW = 5;
  p.lTW = lTW;
  // pixels per chunk: the 16x16 tile does 4x less MFMA work per pixel, so it takes 128-pixel chunks where the halo of
  // 128 pixels still fits one staging pass (rows >= 8 wide)
  if (p.MT == 16) {
    p.WN = p.WK = 1;
  } else if (stride == 2) {  // only two shapes are instantiated for stride 2
    p.WN = p.WK = (N > 32 && K > 32) ? 2 : 1;
  } else {
    p.WN = N > 32 ? 2 : 1;
    p.WK = K > 32 ? 2 : 1;
  }
  p.WS = 4 / (p.WN * p.WK);
  p.PC = wgrad_chunk_pixels(stride, p.MT, lTW, p.WS, ksize * ksize);
  const int TW = 1 << lTW, TH = (p.PC / TW) < TW ? (p.PC / TW) : TW, NI = p.PC / (TW * TH);
  p.tiles_x = (Wo + TW - 1) / TW;
  p.tiles_y = (Ho + TH - 1) / TH;
  p.groups = (B + NI - 1) / NI;
  p.nchunks = p.tiles_x * p.tiles_y * p.groups;
  p.ktiles = (K + p.WK * p.MT - 1) / (p.WK * p.MT);
  p.ntiles = (N + p.WN * p.MT - 1) / (p.WN * p.MT);
  const int tiles = p.ktiles * p.ntiles;
  // blocks to aim for: the tap-split 3x3 tiles are 12-wave blocks of which one fits a CU -> one round of 256
#ifdef HG_WGRAD_TARGET
  const int target = HG_WGRAD_TARGET;
#else
  const int target = (ksize == 3 && p.MT == 32 && HG_WGRAD_TAPSPLIT) ? 256 : 512;
#endif
  int s = (target + tiles - 1) / tiles;
  if (s > p.nchunks) s = p.nchunks;
  if (s < 1) s = 1;
  p.splits = s;
  p.Kp32 = p.ktiles * p.WK * p.MT;
  p.Np32 = p.ntiles * p.WN * p.MT;
  p.slab_bytes = (size_t)s * ksize * ksize * p.Kp32 * p.Np32 * sizeof(float);   // one slab per block (WS waves combined in LDS)
  return p;
}

template <int WN, int WK, int WS, int TAPS, int LTW, int IS, int MT = 32, int TS = 1>
int launch_wgrad_k(const WgradArgs &a, const WgradPlan &p, hipStream_t st) {
  constexpr int PC = wgrad_chunk_pixels(IS, MT, LTW, WS, TAPS);
  using G = CGeom<PC, LTW, TAPS == 9 ? 1 : 0, IS>;
  size_t lds = ((size_t)WN * MT * (PC + 1) + (size_t)WK * MT * G::CHS) * sizeof(float);
  if ((WS == 1 || (TS > 1 && HG_WGRAD_TS_DBUF)) && 2 * lds <= 160 * 1024) lds *= 2;   // double buffered (NBUF in k_wgrad)
  if (a.gw != nullptr && WS == 1 && MT == 32 && TAPS == 9) {   // room for the store transpose of the single-slab case
    const size_t need = (size_t)WN * WK * 32 * WG_TP * sizeof(float);
    if (need > lds) lds = need;
  }
  if (WS > 1) {   // the in-block combination of the WS partial tiles
    const size_t need = (size_t)(WS - 1) * WN * WK * TS * (TAPS / TS) * (MT == 32 ? 16 : 4) * 64 * sizeof(float);
    if (need > lds) lds = need;
  }
  const bool sc = a.iscale != nullptr || a.gscale != nullptr;
  auto kern = sc ? k_wgrad<WN, WK, WS, TAPS, PC, LTW, IS, MT, TS, true> : k_wgrad<WN, WK, WS, TAPS, PC, LTW, IS, MT, TS, false>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.ktiles * p.ntiles), (unsigned)p.splits), dim3(WN * WK * WS * TS * 64), lds, st, a);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

template <int TAPS, int LTW, int IS>
int launch_wgrad_g(const WgradArgs &a, const WgradPlan &p, hipStream_t st) {
  if constexpr (LTW >= 2) {
    if (p.MT == 16) return launch_wgrad_k<1, 1, 4, TAPS, LTW, IS, 16>(a, p, st);
  }
  if (p.WN == 2 && p.WK == 2) {
#if HG_WGRAD_TAPSPLIT
    if constexpr (TAPS == 9) {
      return launch_wgrad_k<2, 2, 1, TAPS, LTW, IS, 32, 3>(a, p, st);   // kernel rows split over waves
    }
#endif
    return launch_wgrad_k<2, 2, 1, TAPS, LTW, IS>(a, p, st);
  }
  if constexpr (IS == 1) {
#if HG_WGRAD_TAPSPLIT > 1
    if constexpr (TAPS == 9) {
      if (a.gw == nullptr) {
        if (p.WN == 2) return launch_wgrad_k<2, 1, 2, TAPS, LTW, IS, 32, 3>(a, p, st);
        if (p.WK == 2) return launch_wgrad_k<1, 2, 2, TAPS, LTW, IS, 32, 3>(a, p, st);
        return launch_wgrad_k<1, 1, 4, TAPS, LTW, IS, 32, 3>(a, p, st);
      }
    }
#endif
    if (p.WN == 2) return launch_wgrad_k<2, 1, 2, TAPS, LTW, IS>(a, p, st);
    if (p.WK == 2) return launch_wgrad_k<1, 2, 2, TAPS, LTW, IS>(a, p, st);
  }
#if HG_WGRAD_TAPSPLIT > 1
  if constexpr (TAPS == 9 && IS == 2) {
    if (a.gw == nullptr) return launch_wgrad_k<1, 1, 4, TAPS, LTW, IS, 32, 3>(a, p, st);
  }
#endif
  return launch_wgrad_k<1, 1, 4, TAPS, LTW, IS>(a, p, st);
}

template <int TAPS, int IS>
int launch_wgrad(WgradArgs a, const WgradPlan &p, float *gw, hipStream_t st) {
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y;
  a.nchunks = p.nchunks; a.splits = p.splits; a.ktiles = p.ktiles; a.Kp32 = p.Kp32; a.Np32 = p.Np32;
  a.gw = (p.splits == 1) ? gw : nullptr;
  int rc;
  switch (p.lTW) {
    case 1: rc = launch_wgrad_g<TAPS, 1, IS>(a, p, st); break;
    case 2: rc = launch_wgrad_g<TAPS, 2, IS>(a, p, st); break;
    case 3: rc = launch_wgrad_g<TAPS, 3, IS>(a, p, st); break;
    case 4: rc = launch_wgrad_g<TAPS, 4, IS>(a, p, st); break;
    default: rc = launch_wgrad_g<TAPS, 5, IS>(a, p, st); break;
  }
  if (rc || a.gw) return rc;
  hipLaunchKernelGGL(k_wgrad_reduce<TAPS>, dim3((unsigned)((a.K + 31) / 32), (unsigned)TAPS, (unsigned)a.N), dim3(256), 0, st,
                     a.slab, gw, a.N, a.K, p.Np32, p.Kp32, p.splits);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

inline bool conv_args_ok(int B, int K, int N, int H, int W, int ksize, int stride) {
  if (B <= 0 || K <= 0 || N <= 0 || H <= 0 || W <= 0) return false;
  if (ksize != 1 && ksize != 3) return false;
  if (stride != 1 && !(stride == 2 && ksize == 3)) return false;
  return true;
}
inline bool fits_i32(int B, int K, int N, int H, int W) {
  const long long lim = 0x7fffffffLL;
  return (long long)B * K * H * W < lim && (long long)B * N * H * W < lim;
}

}  // namespace

extern "C" {

size_t hg_conv_packed_elems(int32_t Co, int32_t Ci, int32_t ksize, int32_t mode) {
  if (Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3)) return 0;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  return (size_t)ksize * ksize * round_up(K, 16) * round_up(N, 128);
}

int hg_conv_pack_weights(const float *w, float *wt, int32_t Co, int32_t Ci, int32_t ksize, int32_t mode, void *stream) {
  if (!w || !wt || Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3)) return HG_EINVAL;
  if (mode != HG_CONV_PACK_FWD && mode != HG_CONV_PACK_DGRAD) return HG_EINVAL;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  const int Kp = round_up(K, 16), Np = round_up(N, 128);
  // the grid covers the padded extent so that the padding is written (as zeros)
  const int CiP = mode == HG_CONV_PACK_FWD ? Kp : Np, CoP = mode == HG_CONV_PACK_FWD ? Np : Kp;
  const dim3 grid((unsigned)((CiP + 31) / 32), (unsigned)((CoP + 31) / 32));
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 3)
    hipLaunchKernelGGL(k_pack<9>, grid, dim3(256), 0, st, w, wt, Co, Ci, Kp, Np, mode);
  else
    hipLaunchKernelGGL(k_pack<1>, grid, dim3(256), 0, st, w, wt, Co, Ci, Kp, Np, mode);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

size_t hg_conv2d_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride,
                                 int32_t dgrad) {
  if (!conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return 0;
  if (dgrad) {
    if (stride != 1) {
      if (Hi == 1 || Wi == 1) return 0;
      const ConvPlan p = plan_conv(B, K, N, (Hi + 1) / 2, (Wi + 1) / 2, 1, 1, true, false);
      return p.tile == TILE_64x64 && p.ksplit > 1 ? (size_t)p.ksplit * B * N * Hi * Wi * sizeof(float) : 0;
    }
    return conv_ws_bytes(plan_conv(B, K, N, Hi, Wi, 1, 1, true, true, ksize * ksize), B, N, Hi, Wi);
  }
  const int Ho = out_size(Hi, stride), Wo = out_size(Wi, stride);
  return conv_ws_bytes(plan_conv(B, K, N, Ho, Wo, stride, 1, true, true, ksize * ksize), B, N, Ho, Wo);
}

int32_t hg_conv_pack_blocks(int32_t Co, int32_t Ci) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (round_up(Ci, 128) / 32) * (round_up(Co, 128) / 32);
}

int hg_conv_pack_weights_multi(const hg_pack_item *items_dev, int32_t n_items, int32_t total_blocks, void *stream) {
  if (!items_dev || n_items <= 0 || total_blocks <= 0) return HG_EINVAL;
  hipLaunchKernelGGL(k_pack_multi, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_conv2d_plan(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride, int32_t dgrad,
                   int32_t out[5]) {
  if (!out || !conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return HG_EINVAL;
  if (dgrad && stride != 1) return HG_EUNSUPPORTED;   // four parity-class launches: no single plan
  const int Hc = dgrad ? Hi : out_size(Hi, stride), Wc = dgrad ? Wi : out_size(Wi, stride);
  const ConvPlan p = plan_conv(B, K, N, Hc, Wc, dgrad ? 1 : stride, 1, true, true, ksize * ksize);
  const bool k2 = ksize == 3 && stride == 1 && HG_CONV_KC == 4 && short_k_chunks(p, B, N, Hc, Wc);
  const int base_kc = (dgrad ? 1 : stride) == 2 ? 4 : HG_CONV_KC;
  out[0] = (int32_t)p.tile;
  out[1] = p.ksplit;
  out[2] = p.tile == TILE_16x256 ? 4 : (p.tile == TILE_64x64 ? 2 * base_kc : (k2 ? 2 : base_kc));
  const long long nb = plan_blocks(p, B, N, Hc, Wc) * p.ksplit;
  out[3] = nb > 0x7fffffffLL ? 0x7fffffff : (int32_t)nb;
  out[4] = num_cus();
  return HG_OK;
}

int hg_conv_pack_weights_both(const float *w, float *wt_fwd, float *wt_dgrad, int32_t Co, int32_t Ci, int32_t ksize,
                              void *stream) {
  if (!w || !wt_fwd || !wt_dgrad || Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3)) return HG_EINVAL;
  const int Kpf = round_up(Ci, 16), Npf = round_up(Co, 128), Kpd = round_up(Co, 16), Npd = round_up(Ci, 128);
  // the grid covers both padded extents so that all padding is written
  const dim3 grid((unsigned)(round_up(Ci, 128) / 32), (unsigned)(round_up(Co, 128) / 32));
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 3)
    hipLaunchKernelGGL(k_pack_both<9>, grid, dim3(256), 0, st, w, wt_fwd, wt_dgrad, Co, Ci, Kpf, Npf, Kpd, Npd);
  else
    hipLaunchKernelGGL(k_pack_both<1>, grid, dim3(256), 0, st, w, wt_fwd, wt_dgrad, Co, Ci, Kpf, Npf, Kpd, Npd);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

static int conv2d_fwd_impl(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                           const float *bias, const float *noise_w, const float *noise_img, int32_t noise_S,
                           float lrelu_slope, int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                           int32_t stride, void *workspace, size_t workspace_bytes, void *stream,
                           const float *addend = nullptr) {
  if (!in || !wt || !out || !conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return HG_EINVAL;
  if (addend && (iscale || oscale || noise_img || lrelu_slope > 0.f)) return HG_EINVAL;   // plain (bias-only) epilogue only
  if ((noise_img != nullptr) != (noise_w != nullptr) || lrelu_slope < 0.f) return HG_EINVAL;
  if (noise_img && (noise_S < out_size(Hi, stride) || noise_S < out_size(Wi, stride))) return HG_EINVAL;
  if (!fits_i32(B, K, N, Hi, Wi)) return HG_EUNSUPPORTED;
  ConvArgs a;
  a.in = in; a.wt = wt; a.out = out; a.iscale = iscale; a.oscale = oscale; a.bias = bias; a.addend = addend;
  a.noise_w = noise_w; a.noise_img = noise_img; a.noise_S = noise_S; a.slope = lrelu_slope;
  a.B = B; a.K = K; a.N = N; a.Hi = Hi; a.Wi = Wi;
  a.Ho = a.Hc = out_size(Hi, stride); a.Wo = a.Wc = out_size(Wi, stride);
  a.os = 1; a.oy = a.ox = 0;
  a.Kp = round_up(K, 16); a.Np = round_up(N, 128);
  Taps tp;
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 1) {
    tp.n = 1; tp.ntx = 1; tp.dy[0] = tp.dx[0] = 0; tp.w[0] = 0;
    return dispatch_conv<1, 1>(a, tp, workspace, workspace_bytes, st);
  }
  tp.n = 9; tp.ntx = 3;
  for (int t = 0; t < 9; ++t) { tp.dy[t] = t / 3 - 1; tp.dx[t] = t % 3 - 1; tp.w[t] = t; }
  return stride == 1 ? dispatch_conv<9, 1>(a, tp, workspace, workspace_bytes, st)
                     : dispatch_conv<9, 2>(a, tp, workspace, workspace_bytes, st);
}

int hg_conv2d_fwd(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                  const float *bias, int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                  int32_t stride, void *workspace, size_t workspace_bytes, void *stream) {
  return conv2d_fwd_impl(in, wt, out, iscale, oscale, bias, nullptr, nullptr, 0, 0.f, B, K, N, Hi, Wi, ksize, stride,
                         workspace, workspace_bytes, stream);
}

int hg_conv2d_fwd_add(const float *in, const float *wt, float *out, const float *addend, const float *bias, int32_t B,
                      int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride, void *workspace,
                      size_t workspace_bytes, void *stream) {
  if (!addend) return HG_EINVAL;
  return conv2d_fwd_impl(in, wt, out, nullptr, nullptr, bias, nullptr, nullptr, 0, 0.f, B, K, N, Hi, Wi, ksize, stride,
                         workspace, workspace_bytes, stream, addend);
}

int hg_modconv2d_fwd(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                     const float *bias, const float *noise_w, const float *noise_img, int32_t noise_S,
                     float lrelu_slope, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, int32_t ksize,
                     void *workspace, size_t workspace_bytes, void *stream) {
  return conv2d_fwd_impl(in, wt, out, iscale, oscale, bias, noise_w, noise_img, noise_S, lrelu_slope, B, K, N, H, W, ksize,
                         1, workspace, workspace_bytes, stream);
}

int hg_conv2d_dgrad(const float *gout, const float *wt, float *gin, const float *iscale, const float *oscale,
                    int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride,
                    void *workspace, size_t workspace_bytes, void *stream) {
  if (!gout || !wt || !gin || !conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return HG_EINVAL;
  if (!fits_i32(B, K, N, Hi, Wi)) return HG_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  ConvArgs a;
  a.in = gout; a.wt = wt; a.out = gin; a.iscale = iscale; a.oscale = oscale; a.bias = nullptr; a.addend = nullptr;
  a.noise_w = a.noise_img = nullptr; a.noise_S = 0; a.slope = 0.f;
  a.B = B; a.K = K; a.N = N;
  a.Hi = out_size(Hi, stride); a.Wi = out_size(Wi, stride);  // the kernel's input is grad_out
  a.Ho = Hi; a.Wo = Wi;
  a.Kp = round_up(K, 16); a.Np = round_up(N, 128);
  Taps tp;
  if (stride == 1) {
    a.Hc = Hi; a.Wc = Wi; a.os = 1; a.oy = a.ox = 0;
    if (ksize == 1) {
      tp.n = 1; tp.ntx = 1; tp.dy[0] = tp.dx[0] = 0; tp.w[0] = 0;
      return dispatch_conv<1, 1>(a, tp, workspace, workspace_bytes, st);
    }
    tp.n = 9; tp.ntx = 3;
    for (int t = 0; t < 9; ++t) { tp.dy[t] = t / 3 - 1; tp.dx[t] = t % 3 - 1; tp.w[t] = t; }
    return dispatch_conv<9, 1>(a, tp, workspace, workspace_bytes, st);
  }
  // stride 2: gin[2y+pY, 2x+pX] = sum over the taps (dy,dx) with dy == pY+1, dx == pX+1 (mod 2) of
  //           gout[y + (pY+1-dy)/2, x + (pX+1-dx)/2] * W[.,.,dy,dx];  the dgrad packing stores W[dy,dx] at tap 8-(3dy+dx)
  a.os = 2;
  // K split decided once for the four parity launches (they fill disjoint pixels of the same slabs)
  int ksplit = 0;
  bool plan64 = false;
  {
    ConvPlan p = plan_conv(B, K, N, (Hi + 1) / 2, (Wi + 1) / 2, 1, 1, workspace != nullptr, false);
    plan64 = p.tile == TILE_64x64 && Hi > 1 && Wi > 1;
    // (a 1-pixel-wide image has empty parity classes, whose slab pixels would never be written: no split then)
    if (Hi > 1 && Wi > 1 && p.tile == TILE_64x64 && p.ksplit > 1 &&
        (size_t)p.ksplit * B * N * Hi * Wi * sizeof(float) <= workspace_bytes)
      ksplit = p.ksplit;
  }
  ConvArgs ca[4];
  Taps tps[4];
  for (int pY = 0; pY < 2; ++pY)
    for (int pX = 0; pX < 2; ++pX) {
      ConvArgs &c = ca[pY * 2 + pX];
      Taps &t = tps[pY * 2 + pX];
      c = a;
      c.Hc = (Hi - pY + 1) / 2; c.Wc = (Wi - pX + 1) / 2;
      c.oy = pY; c.ox = pX;
      t.n = 0; t.ntx = pX ? 2 : 1;
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          if (((pY + 1 - dy) & 1) == 0 && ((pX + 1 - dx) & 1) == 0) {
            t.dy[t.n] = (pY + 1 - dy) / 2; t.dx[t.n] = (pX + 1 - dx) / 2; t.w[t.n] = 8 - (3 * dy + dx);
            ++t.n;
          }
    }
  static const int merge4 = getenv("HG_DGRAD_S2_MERGE") ? atoi(getenv("HG_DGRAD_S2_MERGE")) : 2;
  if (merge4 >= 2 && !plan64 && Hi > 1 && Wi > 1) {
    // large maps: the four classes in one launch as well, for the cache-line pairing of k_conv_parity4's block order
    const ConvPlan p = plan_conv(B, K, N, Hi / 2, Wi / 2, 1, 2, false, false, 4);   // (the smallest class)
    int rc = HG_EUNSUPPORTED;
    switch (p.tile) {
      case TILE_16x256: rc = launch_conv_parity4<1, 4, 1, 4, 4, false, 16, false>(ca, tps, 1, st); break;
      case TILE_32x256: rc = launch_conv_parity4<1, 4, 1, 2, HG_CONV_KC, false, 32, false>(ca, tps, 1, st); break;
      case TILE_64x256: rc = launch_conv_parity4<1, 4, 2, 2, HG_CONV_KC, false, 32, false>(ca, tps, 1, st); break;
      case TILE_128x128: rc = launch_conv_parity4<2, 2, 2, 2, HG_CONV_KC, false, 32, false>(ca, tps, 1, st); break;
      default: break;
    }
    if (rc != HG_EUNSUPPORTED) return rc;
  }
  if (merge4 && plan64) {
    // small maps (the 64x64 tile): the four classes in one launch
    // (one launch has the blocks of all four classes: half the K split planned per class fills the chip as well, with
    // half the slab traffic -- measured best of 1 / 2 / 4)
    static const int ksdiv = getenv("HG_DGRAD_S2_KSDIV") ? atoi(getenv("HG_DGRAD_S2_KSDIV")) : 2;
    if (ksplit > 1) ksplit = ksplit / ksdiv > 1 ? ksplit / ksdiv : 1;
    for (int c = 0; c < 4; ++c) ca[c].slab = (float *)workspace;
    const int rc = launch_conv_parity4<2, 2, 1, 1, 2 * HG_CONV_KC, true>(ca, tps, ksplit > 1 ? ksplit : 1, st);
    if (rc) return rc;
  } else {
    for (int c = 0; c < 4; ++c) {
      if (ca[c].Hc <= 0 || ca[c].Wc <= 0) continue;
      int rc;
      if (tps[c].n == 1) rc = dispatch_conv<1, 1>(ca[c], tps[c], workspace, workspace_bytes, st, ksplit);
      else if (tps[c].n == 2) rc = dispatch_conv<2, 1>(ca[c], tps[c], workspace, workspace_bytes, st, ksplit);
      else rc = dispatch_conv<4, 1>(ca[c], tps[c], workspace, workspace_bytes, st, ksplit);
      if (rc) return rc;
    }
  }
  if (ksplit > 1) {
    a.slab = (float *)workspace;
    return launch_splitk_reduce(a, ksplit, st);
  }
  return HG_OK;
}

size_t hg_conv2d_wgrad_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                                       int32_t stride) {
  if (!conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return 0;
  return make_wgrad_plan(B, K, N, Hi, Wi, ksize, stride).slab_bytes;
}

int hg_conv2d_wgrad(const float *in, const float *gout, float *gw, const float *iscale, const float *gscale, int32_t B,
                    int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride, void *workspace,
                    size_t workspace_bytes, void *stream) {
  if (!in || !gout || !gw || !workspace || !conv_args_ok(B, K, N, Hi, Wi, ksize, stride)) return HG_EINVAL;
  if (!fits_i32(B, K, N, Hi, Wi)) return HG_EUNSUPPORTED;
  const WgradPlan p = make_wgrad_plan(B, K, N, Hi, Wi, ksize, stride);
  if (workspace_bytes < p.slab_bytes) return HG_EWORKSPACE;
  // k_wgrad addresses one image's channels with 32-bit byte offsets below kFar
  if ((long long)K * Hi * Wi >= (1LL << 29) || (long long)N * Hi * Wi >= (1LL << 29)) return HG_EUNSUPPORTED;
  WgradArgs a;
  a.in = in; a.gout = gout; a.slab = (float *)workspace; a.iscale = iscale; a.gscale = gscale;
  a.B = B; a.K = K; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = out_size(Hi, stride); a.Wo = out_size(Wi, stride);
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 1) return launch_wgrad<1, 1>(a, p, gw, st);
  return stride == 1 ? launch_wgrad<9, 1>(a, p, gw, st) : launch_wgrad<9, 2>(a, p, gw, st);
}

}  // extern "C"
