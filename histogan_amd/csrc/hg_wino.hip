// hg_wino.hip -- Winograd F(2x2, 3x3) convolutions on the fp32 MFMA (gfx950): include/hg_wino.h.
//
// The 3x3 stride-1 convolutions of the generator (Conv2DMod, histoGAN/histoGAN.py:431-439) and of the discriminator
// blocks (:510-515), forward and data gradient, are 60 % of a train step on the direct implicit GEMM (hg_conv.hip), which
// already runs at 0.85-0.9 of the fp32-MFMA peak on its best tiles: the only thing left to remove is multiplications.
//
//   Y = A^T [ U . V ] A,   U = G g G^T (4x4 per (n, k), packed once per optimizer step),  V = B^T d B (4x4 per input tile)
//
// One workgroup (8 waves) owns NB output channels x TB tiles (2x2 outputs each) x all 16 transform positions; wave w owns
// positions xi = 2w, 2w+1, i.e. two independent (NB x K) x (K x TB) GEMMs on v_mfma_f32_32x32x2_f32:
//   * A operand (U): every position's weights are used by exactly ONE wave, so they never pass through LDS -- the packing
//     kernel lays them out in lane order and a wave fetches its operands of a K chunk with two 16-byte loads per lane.
//   * B operand (V): each thread loads ONE 4x4 input patch (a (channel, tile) pair of the chunk) straight from global memory
//     with unconditional buffer loads (offset 0xFFFFFFFF = zero padding, no predicates), transforms it in registers (32
//     additions) and writes its 16 values into the position-major LDS buffer the waves read their B operands from.
//   * double-buffered V, ONE barrier per K chunk; loads of chunk c+2 are in flight while chunk c is multiplied.
//   * epilogue: the 16 position sums of an output tile live in 8 different waves; they meet in LDS (the dead V buffers),
//     each thread applies A^T . A to (channel, tile) pairs, the fused epilogue of hg_modconv2d_fwd, and stores 2x2 pixels.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_conv.h"
#include "../../include/hg_wino.h"

#ifndef HG_WINO_BPIPE
#define HG_WINO_BPIPE 1
#endif

namespace {

constexpr unsigned kOOB = 0xFFFFFFFFu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, unsigned bytes = kOOB) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}

struct WinoArgs {
  const float *in, *u;
  float *out;
  const float *iscale, *oscale, *bias, *addend, *noise_w, *noise_img;
  int noise_S;
  float slope;
  int B, K, N, H, W;
  int nblk, nch;           // channel blocks, K chunks (of the packed operand)
  int lTW, lTH, lNI;       // log2 of the block's tile grid: tiles per row, rows, images
  int tiles_w, tiles_h;    // 2x2 tiles of the map
  int bt_x, bt_y;          // block tiles per map row / column
  int ksplit;
  float *slab;
  int blocks, total_tiles;  // tiles per K split, tiles of the launch (= blocks * ksplit)
  int xcd;                  // 1: the channel blocks of a block tile run on ONE XCD (see tile_setup)
};

// accumulator row of register r (v_mfma_f32_32x32x2_f32: D[i][j], j = lane & 31, i = row(r, lane >> 5))
__device__ __forceinline__ int mrow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

// TC x TP MFMA tiles (32 channels x 32 tiles each) per transform position and wave; KC input channels per chunk.
// (TC, TP, KC) = (2, 2, 8): 64 channels x 64 tiles;  (1, 4, 4): 32 channels x 128 tiles (layers with 32 output channels).
template <int TC, int TP, int KC, bool FE>
__global__ __launch_bounds__(512) void k_wino(const WinoArgs a) {
  constexpr int NT = 512, NB = 32 * TC, TB = 32 * TP;
  static_assert(KC * TB == NT, "one (channel, tile) patch per thread and chunk");
  constexpr int NV = (KC / 2) * TC;      // A-operand floats per lane, position and chunk
  static_assert(NV == 2 || NV % 4 == 0, "operand loads are 8 or 16 bytes");
  constexpr int VSZ = 16 * KC * TB;      // floats of one V buffer (32 KB)
  extern __shared__ float smem[];        // [2][16][KC][TB]; the epilogue reuses it as [16][32][32]
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, lk = lane >> 5;
  const int H = a.H, W = a.W, K = a.K, N = a.N, HW = H * W;
  const int TWm = (1 << a.lTW) - 1, THm = (1 << a.lTH) - 1;
  const int t = tid % TB, kc = tid / TB;               // this thread's patch: channel kc of the chunk, tile t of the block
  const size_t in_elems = (size_t)a.B * K * HW;
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u);
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(FE && a.iscale ? a.iscale : a.u);
  const bool has_is = FE && a.iscale != nullptr;
  const int cps = (a.nch + a.ksplit - 1) / a.ksplit;   // K chunks per split

  // ---- state of the tile in work (tile_setup): where it is, and this thread's patch of it.
  // One 16-byte load per patch row (4 instead of 12 loads per patch: a chunk's 16 VMEM issue slots per wave were ~900 of its
  // ~5400 cycles -- ablation, profiles/r05_wino_ablation.txt).  The row starts at column 2 gtx - 1: dword-aligned only,
  // which buffer loads accept; the range check is per instruction, so rows outside the image carry the offset 0xFFFFFFFF
  // (hardware zero fill) while the left / right padding COLUMN of a border tile arrives as a neighbour's value and is zeroed
  // by a select; the one row that would begin an element BEFORE the descriptor's base (image 0 / channel 0 of the chunk, input
  // row 0, left border: offset -4 does not wrap into range) is loaded from column 0 and shifted by a select; the descriptor
  // ends with the tensor (the last row's load reaches one element past it).
  int nb = 0, bx = 0, by = 0, b0 = 0, zs = 0, c_begin = 0, c_end = 0;
  const float *inblk = a.in;
  unsigned vo[4], so = kOOB, uo[2];
  bool c0bad = false, c3bad = false, sh1 = false;
  // tile -> (K split, block tile, channel block).  With >= 8 channel blocks the channel block varies fastest: every XCD only
  // ever sees an eighth of U, and the workgroups that read the same input tiles are neighbours in dispatch order (their
  // second read hits the Infinity Cache).  With 2 or 4 channel blocks (a.xcd) the blocks of one tile are dispatched eight
  // apart instead -- the SAME XCD, the same moment: fabric reads of the 256 -> 128 @64^2 launch 721 -> 411 MB (FETCH_SIZE,
  // calibrated: profiles/r05_fetch_calibration.txt) against 134 MB of input + 2 MB of U; the launch time does not change
  // (the kernel is not bound by the fabric), the traffic does
  auto tile_setup = [&](int L) __attribute__((always_inline)) {
    int pt = L % a.blocks;
    zs = L / a.blocks;
    if (a.xcd) {     // work item L runs on XCD L % 8 (round-robin dispatch, grid a multiple of 8): tile = 8 * group + XCD
      const int q = pt & 7, r = pt >> 3;
      nb = r % a.nblk;
      pt = (r / a.nblk) * 8 + q;
    } else {
      nb = pt % a.nblk;
      pt /= a.nblk;
    }
    bx = pt % a.bt_x;
    pt /= a.bt_x;
    by = pt % a.bt_y;
    b0 = (pt / a.bt_y) << a.lNI;
    c_begin = zs * cps;
    c_end = c_begin + cps < a.nch ? c_begin + cps : a.nch;
    inblk = a.in + (size_t)b0 * K * HW;
    const int ttx = t & TWm, tty = (t >> a.lTW) & THm, timg = t >> (a.lTW + a.lTH);
    const int gtx = (bx << a.lTW) + ttx, gty = (by << a.lTH) + tty, b = b0 + timg;
    const bool ok = gtx < a.tiles_w && gty < a.tiles_h && b < a.B;
    const int x = 2 * gtx - 1, y0 = 2 * gty - 1;
    c0bad = x < 0;
    c3bad = x + 3 >= W;
    sh1 = ok && timg == 0 && kc == 0 && gty == 0 && gtx == 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = y0 + r;
      const bool rok = ok && (unsigned)y < (unsigned)H;
      const unsigned e = (unsigned)(((timg * K + kc) * H + y) * W + x);
      vo[r] = rok ? (r == 1 && sh1 ? 0u : e * 4u) : kOOB;
    }
    so = (FE && ok) ? (unsigned)(b * K + kc) * 4u : kOOB;
    // this wave's operand slices of U: position xi = 2 wave + x2
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
      uo[x2] = (unsigned)((((2 * wave + x2) * a.nblk + nb) * a.nch) * (NV * 64) + lane * (NV >= 4 ? 4 : 2)) * 4u;
  };

  f32x16 acc[2][TC][TP];
  float pd[2][16];      // [register set][patch element]: the patch of chunk c lives in set (c - c_begin) & 1
  float ps[2] = {1.f, 1.f};
  float ua[2][2][NV];   // [register set][position][operand]

  auto load_patch = [&](int c, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const size_t rem = in_elems - ((size_t)b0 * K * HW + (size_t)c * KC * HW);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(inblk + (size_t)c * KC * HW, rem < (1ull << 30) ? (unsigned)rem * 4u : kOOB);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 m = buf_load4(rx, vo[r], 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) pd[S][4 * r + e] = m[e];
    }
    if constexpr (FE) {
      if (has_is) ps[S] = buf_load(rs, so, c * KC * 4);
    }
  };
  auto load_u = [&](int c, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2) {
      if constexpr (NV >= 4) {
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
          const f32x4 v = buf_load4(ru, uo[x2] + (unsigned)q * 1024u, c * NV * 64 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) ua[S][x2][4 * q + e] = v[e];
        }
      } else {
        const f32x2 v = buf_load2(ru, uo[x2], c * NV * 64 * 4);
        ua[S][x2][0] = v[0];
        ua[S][x2][1] = v[1];
      }
    }
  };
  // V = B^T d B of patch set SET into buffer `buf`
  auto transform_store = [&](int buf, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    float *Vb = smem + buf * VSZ + kc * TB + t;
    float d[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = has_is ? pd[S][e] * ps[S] : pd[S][e];
    {
      const float m0 = d[4], m1 = d[5], m2 = d[6];
      d[5] = sh1 ? m0 : m1;
      d[6] = sh1 ? m1 : m2;
      d[7] = sh1 ? m2 : d[7];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[4 * r] = c0bad ? 0.f : d[4 * r];
      d[4 * r + 3] = c3bad ? 0.f : d[4 * r + 3];
    }
    float q[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[0 + c] = d[0 + c] - d[8 + c];
      q[4 + c] = d[4 + c] + d[8 + c];
      q[8 + c] = d[8 + c] - d[4 + c];
      q[12 + c] = d[4 + c] - d[12 + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Vb[(4 * r + 0) * KC * TB] = q[4 * r] - q[4 * r + 2];
      Vb[(4 * r + 1) * KC * TB] = q[4 * r + 1] + q[4 * r + 2];
      Vb[(4 * r + 2) * KC * TB] = q[4 * r + 2] - q[4 * r + 1];
      Vb[(4 * r + 3) * KC * TB] = q[4 * r + 1] - q[4 * r + 3];
    }
  };
  auto mfma_chunk = [&](int buf, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const float *Vc = smem + buf * VSZ + lk * TB + lm;
#if HG_WINO_BPIPE
    // B operands one MFMA step ahead (the two-slot pipeline of k_conv, hg_conv.hip): the LDS read of step s + 1 is issued in
    // front of the MFMAs of step s.  The compiler's own order reads a step's operands right in front of its MFMAs, into the A
    // registers the previous step released: a full LDS round trip per group of TC x TP MFMAs, exposed whenever the SIMD's
    // other wave is not ready -- under oldest-first arbitration that is the second half of every phase.
    constexpr int STEPS = KC;          // 2 positions x KC / 2 k steps
    float bv[2][TP];
    auto ldb = [&](int st, int slot) __attribute__((always_inline)) {
      const int x2 = st / (KC / 2), ks = st % (KC / 2);
#pragma unroll
      for (int j = 0; j < TP; ++j) bv[slot][j] = Vc[((2 * wave + x2) * KC + 2 * ks) * TB + 32 * j];
    };
    constexpr int DSN = (TP + 1) / 2;  // ds_read2_b32 per step
    ldb(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, DSN, 0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int x2 = st / (KC / 2), ks = st % (KC / 2);
      if (st + 1 < STEPS) ldb(st + 1, (st + 1) & 1);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
          acc[x2][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[S][x2][ks * TC + i], bv[st & 1][j], acc[x2][i][j], 0, 0, 0);
      if (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, DSN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TC * TP, 0);
    }
#else
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int ks = 0; ks < KC / 2; ++ks) {
        float bv[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) bv[j] = Vc[((2 * wave + x2) * KC + 2 * ks) * TB + 32 * j];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j)
            acc[x2][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[S][x2][ks * TC + i], bv[j], acc[x2][i][j], 0, 0, 0);
      }
#endif
  };
  auto clampc = [&](int c) __attribute__((always_inline)) { return c < c_end ? c : c_end - 1; };
#define HG_WINO_BARRIER()               \
  __builtin_amdgcn_sched_barrier(0);    \
  __syncthreads();                      \
  __builtin_amdgcn_sched_barrier(0)

  // PERSISTENT: one workgroup per CU walks over the launch's tiles, and the first chunk of the NEXT tile (its patch and its
  // weights) is requested before the epilogue of the current one -- a tile otherwise starts with a full global round trip
  // in front of its first MFMA and ends with ~5 us of LDS exchange + stores during which nothing is in flight.
  int L = blockIdx.x;
  if (L < a.total_tiles) {
    tile_setup(L);
    if (c_begin < c_end) {
      load_patch(c_begin, S0{});
      load_u(c_begin, S0{});
    }
  }
  for (; L < a.total_tiles; L += gridDim.x) {
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[x2][i][j][r] = 0.f;

    // K chunks of this tile, two phases per trip (V buffer / register sets 0, then 1).  Phase c: request the patch of chunk
    // c + 2 FIRST (into the set the previous phase's transform freed), multiply chunk c while transforming the patch of
    // chunk c + 1 (requested a whole phase ago: the compiler interleaves these additions with the MFMAs without exposing the
    // load latency), then request the weights of chunk c + 2 into the set the MFMAs released; ONE barrier.  Every load of
    // the loop body is UNCONDITIONAL (past the end the chunk index is clamped: a redundant load nobody consumes): with loads
    // behind branches the compiler assumes the shortest path and waits for (almost) every outstanding load before the first
    // MFMA of a chunk.  sched_barrier at the phase boundaries: the scheduler otherwise hoists the first additions of the
    // NEXT transform across the barrier, right behind the loads they consume, or sinks the phase's loads behind its MFMAs.
    // (Measured and rejected, profiles/r05_wino_ablation.txt, the code is in the history at commit f419444: the loads spread between the MFMA
    // steps with sched_group_barrier, +10 %; the two waves of a SIMD in opposite halves of a chunk with two barriers
    // ("ping-pong"), +5 %; MFMAs and transform as separate scheduling regions, +-0.)
    const int nc = c_end - c_begin;
    if (nc > 0) {
      load_patch(clampc(c_begin + 1), S1{});
      load_u(clampc(c_begin + 1), S1{});
      transform_store(0, S0{});
      HG_WINO_BARRIER();
      for (int c = c_begin; c + 1 < c_end; c += 2) {
        load_patch(clampc(c + 2), S0{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(0, S0{});
        transform_store(1, S1{});
        load_u(clampc(c + 2), S0{});
        HG_WINO_BARRIER();
        load_patch(clampc(c + 3), S1{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(1, S1{});
        transform_store(0, S0{});
        load_u(clampc(c + 3), S1{});
        HG_WINO_BARRIER();
      }
      if (nc & 1) mfma_chunk(0, S0{});
    }
    __builtin_amdgcn_sched_barrier(0);

    // the finished tile's coordinates for the epilogue; then the next tile's set-up and first requests
    const int e_n0 = nb * NB, e_bx = bx, e_by = by, e_b0 = b0, e_zs = zs;
    if (L + (int)gridDim.x < a.total_tiles) {
      tile_setup(L + (int)gridDim.x);
      if (c_begin < c_end) {
        load_patch(c_begin, S0{});
        load_u(c_begin, S0{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: one (channel tile, tile tile) pair at a time through LDS
  const bool fin = a.ksplit == 1;
  float *ob = fin ? a.out : a.slab + (size_t)e_zs * a.B * N * HW;
  const int erow = tid >> 5, ecol = tid & 31;
  // Epilogue operands, requested BEFORE the first pass so that their latency runs under the LDS exchange (read where they
  // are used, each pass waited for a global round trip of its own): per-channel bias / noise weight of this thread's 2 TC
  // channels, demodulation scale per (channel, tile), the noise image rows of its TP tiles.
  int etb[TP], ety[TP], etx[TP];
  bool eok[TP];
  float ebias[TC][2], enw[TC][2], eosc[TC][2][TP];
  f32x2 enz[TP][2];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int tl = 32 * j + ecol;
    const int ttx = tl & TWm, tty = (tl >> a.lTW) & THm, timg = tl >> (a.lTW + a.lTH);
    etx[j] = (e_bx << a.lTW) + ttx; ety[j] = (e_by << a.lTH) + tty; etb[j] = e_b0 + timg;
    eok[j] = etx[j] < a.tiles_w && ety[j] < a.tiles_h && etb[j] < a.B;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      enz[j][p2] = f32x2{0.f, 0.f};
      if (fin && a.noise_img && eok[j])
        enz[j][p2] = *reinterpret_cast<const f32x2 *>(a.noise_img + ((size_t)etb[j] * a.noise_S + 2 * ety[j] + p2) * a.noise_S + 2 * etx[j]);
    }
  }
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = e_n0 + 32 * i + erow + 16 * h, nc_ = n < N ? n : N - 1;
      ebias[i][h] = (fin && a.bias) ? a.bias[nc_] : 0.f;
      enw[i][h] = (fin && a.noise_img) ? a.noise_w[nc_] : 0.f;
#pragma unroll
      for (int j = 0; j < TP; ++j) eosc[i][h][j] = (fin && a.oscale && eok[j]) ? a.oscale[etb[j] * N + nc_] : 1.f;
    }
#pragma unroll
  for (int i = 0; i < TC; ++i) {
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      // the residual addend of this pass (plain epilogue only), requested ahead of the exchange as well
      f32x2 ead[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          ead[h][p2] = f32x2{0.f, 0.f};
          const int n = e_n0 + 32 * i + erow + 16 * h;
          if (fin && a.addend && eok[j] && n < N)
            ead[h][p2] = *reinterpret_cast<const f32x2 *>(a.addend + (((size_t)etb[j] * N + n) * H + 2 * ety[j] + p2) * W + 2 * etx[j]);
        }
      __syncthreads();   // (first pass: every wave is done with the V buffers this staging area aliases)
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[((2 * wave + x2) * 32 + mrow(r, lk)) * 32 + lm] = acc[x2][i][j][r];
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = erow + 16 * h;
        const int n = e_n0 + 32 * i + row;
        float m[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = smem[(e * 32 + row) * 32 + ecol];
        if (!eok[j] || n >= N) continue;
        float s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s0[c] = m[c] + m[4 + c] + m[8 + c];
          s1[c] = m[4 + c] - m[8 + c] - m[12 + c];
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2];
        y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2];
        y[1][1] = s1[1] - s1[2] - s1[3];
        const size_t o = (((size_t)etb[j] * N + n) * H + 2 * ety[j]) * W + 2 * etx[j];
        if (fin) {
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              float v = fmaf(y[p2][q], eosc[i][h][j], fmaf(enw[i][h], enz[j][p2][q], ebias[i][h])) + ead[h][p2][q];
              if (a.slope > 0.f) v = v > 0.f ? v : a.slope * v;
              y[p2][q] = v;
            }
        }
        *reinterpret_cast<f32x2 *>(ob + o) = f32x2{y[0][0], y[0][1]};
        *reinterpret_cast<f32x2 *>(ob + o + W) = f32x2{y[1][0], y[1][1]};
      }
    }
  }
    __syncthreads();   // the next tile's first transform writes the V buffer this staging area aliases
  }
#undef HG_WINO_BARRIER
}

// out = epilogue(sum_z slab[z]) in fixed order (as k_splitk_reduce of hg_conv.hip)
__global__ __launch_bounds__(256) void k_wino_reduce(const float *__restrict__ slab, float *__restrict__ out,
                                                     const float *__restrict__ oscale, const float *__restrict__ bias,
                                                     const float *__restrict__ noise_w, const float *__restrict__ noise_img,
                                                     const float *__restrict__ addend, int noise_S, float slope,
                                                     long long total4, int HW, int W, int N, int ksplit) {
  const size_t total = (size_t)total4 * 4;
  for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long long)gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ksplit; ++z) v += *reinterpret_cast<const f32x4 *>(slab + (size_t)z * total + (size_t)i4 * 4);
    const long long i = i4 * 4;
    const long long bn = i / HW;
    const int n = (int)(bn % N);
    const float bs = bias ? bias[n] : 0.f, osc = oscale ? oscale[bn] : 1.f, nw = noise_img ? noise_w[n] : 0.f;
    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
    if (noise_img) {
      const int p = (int)(i - bn * HW), y = p / W, x = p - y * W;   // W % 4 == 0 is not guaranteed: element-wise
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xe = x + e, ye = y + xe / W;
        nz[e] = noise_img[((size_t)(bn / N) * noise_S + ye) * noise_S + xe % W];
      }
    }
    f32x4 ad = {0.f, 0.f, 0.f, 0.f};
    if (addend) ad = *reinterpret_cast<const f32x4 *>(addend + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r = fmaf(v[e], osc, fmaf(nw, nz[e], bs)) + ad[e];
      if (slope > 0.f) r = r > 0.f ? r : slope * r;
      v[e] = r;
    }
    *reinterpret_cast<f32x4 *>(out + i) = v;
  }
}

// U = G g G^T of every (n, k) pair in the lane order k_wino<TC, *, KC> loads: one thread per operand slot of chunk `ch`,
// channel block `nb`, sixteen coalesced stores.  mode DGRAD: k = Co, n = Ci, taps flipped (== positions 0 <-> 3 swapped
// in both directions, since G J = P G for the 3x3 flip J and the row swap P).
template <int TC, int KC>
__device__ __forceinline__ void wino_pack_slot(const float *__restrict__ w, float *__restrict__ u, int Co, int Ci, int nblk,
                                               int nch, int mode, int ch, int nb, int tid /* < NV * 64 */,
                                               float *__restrict__ wsq = nullptr) {
  constexpr int NV = (KC / 2) * TC, NB = 32 * TC;
  int e, lane;
  if constexpr (NV >= 4) {
    e = (tid >> 8) * 4 + (tid & 3);
    lane = (tid >> 2) & 63;
  } else {
    e = tid & 1;
    lane = tid >> 1;
  }
  const int ks = e / TC, i = e % TC, lm = lane & 31, lk = lane >> 5;
  const int k = ch * KC + 2 * ks + lk, n = nb * NB + 32 * i + lm;
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  float g[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) g[q] = 0.f;
  if (k < K && n < N) {
    const float *p = w + (mode == HG_CONV_PACK_FWD ? ((size_t)n * Ci + k) : ((size_t)k * Ci + n)) * 9;
#pragma unroll
    for (int q = 0; q < 9; ++q) g[q] = p[q];
    if (wsq != nullptr && mode == HG_CONV_PACK_FWD) {   // every (co, ci) pair has exactly one slot per operand: written once
      float q2 = 0.f;
#pragma unroll
      for (int q = 0; q < 9; ++q) q2 = fmaf(g[q], g[q], q2);   // (the tap order of pack_both_tile in hg_conv.hip)
      wsq[(size_t)n * Ci + k] = q2;
    }
  }
  // rows: G g (4 x 3), then columns: (G g) G^T (4 x 4)
  float gg[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    gg[0][c] = g[c];
    gg[1][c] = 0.5f * (g[c] + g[3 + c] + g[6 + c]);
    gg[2][c] = 0.5f * (g[c] - g[3 + c] + g[6 + c]);
    gg[3][c] = g[6 + c];
  }
  float uu[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uu[r][0] = gg[r][0];
    uu[r][1] = 0.5f * (gg[r][0] + gg[r][1] + gg[r][2]);
    uu[r][2] = 0.5f * (gg[r][0] - gg[r][1] + gg[r][2]);
    uu[r][3] = gg[r][2];
  }
  const size_t xstride = (size_t)nblk * nch * (NV * 64);
  float *dst = u + ((size_t)nb * nch + ch) * (NV * 64) + tid;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int rr = mode == HG_CONV_PACK_FWD ? r : (r == 0 ? 3 : (r == 3 ? 0 : r));
      const int cc = mode == HG_CONV_PACK_FWD ? c : (c == 0 ? 3 : (c == 3 ? 0 : c));
      dst[(size_t)(4 * r + c) * xstride] = uu[rr][cc];
    }
}

// geometry of the packed operand of one (weight, mode): variant (0: 64-channel blocks, 8-channel chunks; 1: 32 / 4),
// channel blocks, chunks, and the 512-thread pack blocks that write it (variant 1: four chunks per block)
struct PackGeom {
  int variant, nblk, nch, blocks;
};
__host__ __device__ inline PackGeom pack_geom(int Co, int Ci, int mode, int force_variant) {
  const int K = mode == HG_CONV_PACK_FWD ? Ci : Co, N = mode == HG_CONV_PACK_FWD ? Co : Ci;
  PackGeom g;
  g.variant = force_variant >= 0 ? force_variant : (N <= 32 ? 1 : 0);
  const int NB = g.variant ? 32 : 64, KC = g.variant ? 4 : 8;
  g.nblk = (N + NB - 1) / NB;
  g.nch = K / KC;
  g.blocks = (K % KC) ? 0 : (g.variant ? ((g.nch + 3) / 4) * g.nblk : g.nch * g.nblk);
  return g;
}
// pack block `local` (512 threads) of one (weight, mode)
__device__ __forceinline__ void wino_pack_block(const float *w, float *u, int Co, int Ci, int mode, const PackGeom &g, int local,
                                                float *wsq = nullptr) {
  if (g.variant == 0) {
    wino_pack_slot<2, 8>(w, u, Co, Ci, g.nblk, g.nch, mode, local % g.nch, local / g.nch, threadIdx.x, wsq);
  } else {
    const int cg = (g.nch + 3) / 4;
    const int ch = (local % cg) * 4 + (threadIdx.x >> 7);
    if (ch < g.nch) wino_pack_slot<1, 4>(w, u, Co, Ci, g.nblk, g.nch, mode, ch, local / cg, threadIdx.x & 127, wsq);
  }
}
__global__ __launch_bounds__(512) void k_wino_pack(const float *__restrict__ w, float *__restrict__ u, int Co, int Ci, int mode,
                                                   int force_variant) {
  const PackGeom g = pack_geom(Co, Ci, mode, force_variant);
  wino_pack_block(w, u, Co, Ci, mode, g, blockIdx.x);
}
// every 3x3 weight of a model, both modes, in ONE launch (the table of hg_wino_pack_weights_multi)
__global__ __launch_bounds__(512) void k_wino_pack_multi(const hg_wino_pack_item *__restrict__ items, int n_items,
                                                         int force_variant) {
  int it = 0;
  while (it + 1 < n_items && (int)blockIdx.x >= items[it + 1].block_begin) ++it;   // wave-uniform scan
  const hg_wino_pack_item im = items[it];
  int local = (int)blockIdx.x - im.block_begin;
  const PackGeom gf = pack_geom(im.Co, im.Ci, HG_CONV_PACK_FWD, force_variant);
  const int bf = im.u_fwd ? gf.blocks : 0;
  if (local < bf) {
    wino_pack_block(im.w, im.u_fwd, im.Co, im.Ci, HG_CONV_PACK_FWD, gf, local, im.wsq);
    return;
  }
  local -= bf;
  const PackGeom gd = pack_geom(im.Co, im.Ci, HG_CONV_PACK_DGRAD, force_variant);
  if (im.u_dgrad && local < gd.blocks) wino_pack_block(im.w, im.u_dgrad, im.Co, im.Ci, HG_CONV_PACK_DGRAD, gd, local);
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW = G^T [ sum_tiles (A dY A^T) . (B^T d B) ] G
// Per transform position the sum over tiles is an (N x tiles) x (tiles x K) GEMM whose reduction index is the tile.  One
// workgroup owns 64 n x 64 k x 16 positions (wave w: positions 2w, 2w+1, 2x2 MFMA tiles each) and walks over chunks of 8
// tiles: each thread transforms ONE input patch (channel k, tile t) into V and ONE 2x2 output-gradient tile (channel n,
// tile t) into A dY A^T, both position-major in LDS ([position][tile][channel], pitch 68: conflict-free stores with lanes
// along the tiles, conflict-free operand reads with lanes along the channels).  The chunks of a block are strided over
// the splits; the partial sums go to slabs [split][position][n][k] and k_wino_wgrad_reduce sums them in fixed order and
// applies G^T . G.
struct WinoWgArgs {
  const float *in, *gout;
  float *slab;
  int B, K, N, H, W;
  int ktiles, splits, nchunks;
  int lPW, lPH, lPI;       // log2 of the chunk pattern: tiles per row, rows, images (PW * PH * PI == 8)
  int lcg, lrg;            // log2 of the column / row groups of a map (tiles_w >> lPW, tiles_h >> lPH)
  int Np, Kp;              // slab extents (multiples of 64)
  int xtiles, total_blocks;  // (n, k) tiles, tiles x splits
  int xcd;                   // 1: all (n, k) tiles of a pixel split run on ONE XCD
};

constexpr int WG_P = 68;                   // LDS row pitch (floats) of a [position][tile] row of 64 channels
constexpr int WG_OP = 16 * 8 * WG_P;       // floats of one operand buffer

__global__ __launch_bounds__(512) void k_wino_wgrad(const WinoWgArgs a) {
  extern __shared__ float smem[];          // [2 buffers][dM, V][16][8][WG_P]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, lk = lane >> 5;
  const int H = a.H, W = a.W, K = a.K, N = a.N, HW = H * W;
  for (int L = blockIdx.x; L < a.total_blocks; L += gridDim.x) {    // persistent, as k_wino
  // Work item L runs on XCD L % 8.  a.xcd: split = 8 * group + XCD and the (n, k) tiles of a split follow each other eight
  // apart -- every block that reads a pixel range sits behind the same L2, so `in` and `gout` cross the fabric once instead
  // of once per n tile / k tile (the old order, (n, k) tile fastest, gave each XCD ONE channel tile and all the pixels:
  // fabric reads 1917 MB at 256 -> 128 @64^2 against 201 MB of operands; now 201.5 MB).  Same slabs, same sums: bit-identical.
  int bxy, sp;
  if (a.xcd) {
    const int q = L & 7, r = L >> 3;
    bxy = r % a.xtiles;
    sp = (r / a.xtiles) * 8 + q;
  } else {
    bxy = L % a.xtiles;
    sp = L / a.xtiles;
  }
  const int k0 = (bxy % a.ktiles) * 64, n0 = (bxy / a.ktiles) * 64;

  // ---- transform role: tile t of the chunk pattern, channel ch of the block's 64 (input channel k0 + ch / gradient
  //      channel n0 + ch)
  const int t = lane & 7, ch = wave * 8 + (lane >> 3);
  const int PWm = (1 << a.lPW) - 1, PHm = (1 << a.lPH) - 1;
  const int dtx = t & PWm, dty = (t >> a.lPW) & PHm, dimg = t >> (a.lPW + a.lPH);
  const int nrg = 1 << a.lrg, ncg = 1 << a.lcg;
  // Zero padding = the byte offset 0xFFFFFFFF (the descriptor's range check answers 0.0).  Which elements of a patch are
  // padding depends on the chunk (scalar: first / last row group, first / last column group) AND on the lane (its tile's
  // row / column inside the chunk pattern): all-ones / zero MASKS, OR-ed onto the offsets -- no predicates, no branches
  // (written as `cond ? kOOB : offset` the compiler builds exec-masked branches around every load).
  const unsigned m_row0 = dty == 0 ? kOOB : 0u, m_rowl = dty == PHm ? kOOB : 0u;
  const unsigned m_tx0 = dtx == 0 ? kOOB : 0u, m_txl = dtx == PWm ? kOOB : 0u;
  const unsigned m_kbad = k0 + ch < K ? 0u : kOOB, m_nbad = n0 + ch < N ? 0u : kOOB;
  // patch element (r, c) relative to the chunk origin, against a base one row and one column BEFORE the image group
#if HG_WINO_ROWLOAD
  // one dword-aligned 16-byte load per patch row (as k_wino): the padding columns of border tiles arrive as neighbours'
  // values and are cleared with the chunk's column masks when the patch is transformed; the one row that would begin
  // one element before the tensor (image 0, channel 0, row 0, left border) is loaded from column 0 and shifted
  unsigned vo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) vo[r] = (unsigned)(((dimg * K + k0 + ch) * H + 2 * dty + r) * W + 2 * dtx) * 4u;
  const unsigned m_first = (dimg == 0 && k0 + ch == 0 && dty == 0 && dtx == 0) ? kOOB : 0u;
  const size_t in_elems = (size_t)a.B * K * HW;
  unsigned cmask[2][3];      // [register set][left column, right column, shifted row 1]
#else
  unsigned vo[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned e = (unsigned)(((dimg * K + k0 + ch) * H + 2 * dty + r) * W + 2 * dtx);
    vo[r][0] = e * 4u;
    vo[r][1] = (e + 1u) * 4u;
    vo[r][2] = (e + 3u) * 4u;
  }
#endif
  const unsigned go = (unsigned)(((dimg * N + n0 + ch) * H + 2 * dty) * W + 2 * dtx) * 4u;

  f32x16 acc[2][2][2];
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x2][i][j][r] = 0.f;

  float pd[2][16], gd[2][4];   // [register set][element]: chunk iteration `it` lives in set it & 1
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  auto load_chunk = [&](int c, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    // chunk -> (image group, row group, column group): scalar
    const int cg = c & (ncg - 1), rg = (c >> a.lcg) & (nrg - 1), ig = c >> (a.lcg + a.lrg);
    const int b0 = ig << a.lPI;
    const unsigned top = rg == 0 ? kOOB : 0u, bot = rg == nrg - 1 ? kOOB : 0u;          // scalar masks
    const unsigned left = cg == 0 ? kOOB : 0u, right = cg == ncg - 1 ? kOOB : 0u;
    const unsigned m_img = (unsigned)((a.B - 1 - b0 - dimg) >> 31);                       // all ones when b0 + dimg >= B
    const unsigned soff = (unsigned)(((rg << a.lPH) * 2) * W + (cg << a.lPW) * 2) * 4u;
    const __amdgpu_buffer_rsrc_t rg_ = make_rsrc(a.gout + (size_t)b0 * N * HW);
    const unsigned base = m_kbad | m_img;
    const unsigned c0 = left & m_tx0, c3 = right & m_txl;
    unsigned rm[4];
    rm[0] = base | (top & m_row0);
    rm[1] = rm[2] = base;
    rm[3] = base | (bot & m_rowl);
#if HG_WINO_ROWLOAD
    // (the descriptor ends with the tensor: the last row's 16-byte load reaches one element past it)
    const size_t rem = in_elems - (size_t)b0 * K * HW + (size_t)(W + 1);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.in + (size_t)b0 * K * HW - (W + 1), rem < (1ull << 30) ? (unsigned)rem * 4u : kOOB);
    const unsigned shm = (b0 == 0 && rg == 0 && cg == 0) ? m_first : 0u;
    cmask[S][0] = c0; cmask[S][1] = c3; cmask[S][2] = shm;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 m = buf_load4(rx, (vo[r] + (r == 1 ? (shm & 4u) : 0u)) | rm[r], (int)soff);
#pragma unroll
      for (int e = 0; e < 4; ++e) pd[S][4 * r + e] = m[e];
    }
#else
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.in + (size_t)b0 * K * HW - (W + 1));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pd[S][4 * r] = buf_load(rx, vo[r][0] | rm[r] | c0, (int)soff);
      const f32x2 m = buf_load2(rx, vo[r][1] | rm[r], (int)soff);
      pd[S][4 * r + 1] = m[0];
      pd[S][4 * r + 2] = m[1];
      pd[S][4 * r + 3] = buf_load(rx, vo[r][2] | rm[r] | c3, (int)soff);
    }
#endif
    const unsigned gm = m_nbad | m_img;
    const f32x2 g0 = buf_load2(rg_, go | gm, (int)soff);
    const f32x2 g1 = buf_load2(rg_, (go + (unsigned)W * 4u) | gm, (int)soff);
    gd[S][0] = g0[0]; gd[S][1] = g0[1]; gd[S][2] = g1[0]; gd[S][3] = g1[1];
  };
  auto transform_store = [&](int buf, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    float *Mb = smem + buf * 2 * WG_OP + t * WG_P + ch;   // dM [position][tile][n]
    float *Vb = Mb + WG_OP;                               // V  [position][tile][k]
    // V = B^T d B
    float d[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = pd[S][e];
#if HG_WINO_ROWLOAD
    {
      const bool sh = cmask[S][2] != 0u;
      const float m0 = d[4], m1 = d[5], m2 = d[6];
      d[5] = sh ? m0 : m1;
      d[6] = sh ? m1 : m2;
      d[7] = sh ? m2 : d[7];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[4 * r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d[4 * r]) & ~cmask[S][0]);
        d[4 * r + 3] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d[4 * r + 3]) & ~cmask[S][1]);
      }
    }
#endif
    float q[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[0 + c] = d[0 + c] - d[8 + c];
      q[4 + c] = d[4 + c] + d[8 + c];
      q[8 + c] = d[8 + c] - d[4 + c];
      q[12 + c] = d[4 + c] - d[12 + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Vb[(4 * r + 0) * 8 * WG_P] = q[4 * r] - q[4 * r + 2];
      Vb[(4 * r + 1) * 8 * WG_P] = q[4 * r + 1] + q[4 * r + 2];
      Vb[(4 * r + 2) * 8 * WG_P] = q[4 * r + 2] - q[4 * r + 1];
      Vb[(4 * r + 3) * 8 * WG_P] = q[4 * r + 1] - q[4 * r + 3];
    }
    // dM = A dY A^T,  A = [[1, 0], [1, 1], [1, -1], [0, -1]]
    const float *g_ = gd[S];
    const float r0[2] = {g_[0], g_[1]}, r1[2] = {g_[0] + g_[2], g_[1] + g_[3]}, r2[2] = {g_[0] - g_[2], g_[1] - g_[3]},
                r3[2] = {-g_[2], -g_[3]};
    const float *rows[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Mb[(4 * r + 0) * 8 * WG_P] = rows[r][0];
      Mb[(4 * r + 1) * 8 * WG_P] = rows[r][0] + rows[r][1];
      Mb[(4 * r + 2) * 8 * WG_P] = rows[r][0] - rows[r][1];
      Mb[(4 * r + 3) * 8 * WG_P] = -rows[r][1];
    }
  };
  auto mfma_chunk = [&](int buf) __attribute__((always_inline)) {
    const float *Mc = smem + buf * 2 * WG_OP + lk * WG_P + lm;
    const float *Vc = Mc + WG_OP;
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int row = ((2 * wave + x2) * 8 + 2 * s_) * WG_P;
        float av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = Mc[row + 32 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = Vc[row + 32 * j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[x2][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[x2][i][j], 0, 0, 0);
      }
  };

  // this block's chunks: a contiguous range (consecutive chunks continue along the tile row: the other half of the cache
  // lines just fetched); two per trip, loads unconditional with a clamped index, as k_wino
  const int cps = (a.nchunks + a.splits - 1) / a.splits;
  const int c0 = sp * cps;
  const int nc = c0 < a.nchunks ? (c0 + cps <= a.nchunks ? cps : a.nchunks - c0) : 0;
  auto chunk_of = [&](int it) __attribute__((always_inline)) { return c0 + (it < nc ? it : nc - 1); };
  if (nc > 0) {
    // phase `it`: request chunk it + 2 first, multiply chunk it, transform chunk it + 1 (requested a phase ago), barrier
    load_chunk(chunk_of(0), S0{});
    load_chunk(chunk_of(1), S1{});
    transform_store(0, S0{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it + 1 < nc; it += 2) {
      load_chunk(chunk_of(it + 2), S0{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(0);
      transform_store(1, S1{});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(chunk_of(it + 3), S1{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(1);
      transform_store(0, S0{});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nc & 1) mfma_chunk(0);
  }

  // slab[split][position][n][k]: D[i = n][j = k], 32 consecutive lanes write 32 consecutive k
  float *sb = a.slab + (size_t)sp * 16 * a.Np * a.Kp;
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sb[((size_t)(2 * wave + x2) * a.Np + n0 + 32 * i + mrow(r, lk)) * a.Kp + k0 + 32 * j + lm] = acc[x2][i][j][r];
  __syncthreads();   // (the next tile's first transform writes the LDS buffers the slowest wave may still read)
  }
}

// gw[n][k][3][3] = G^T (sum_s slab[s][.][n][k]) G.  Block = (n, 4 * KQ consecutive k), KQ = 256 / SG: thread (kq = 4 k's,
// sg = one of SG split groups) sums its splits (16-byte loads, sixteen independent streams), applies the (linear) transform
// to its partial sum, and the SG groups combine in LDS in fixed order (deterministic); 36 * KQ contiguous floats per block
// are stored.  SG = 32 / 8 / 1 by the number of splits (many splits: few (n, k); few splits: many).
template <int SG>
__global__ __launch_bounds__(256) void k_wino_wgrad_reduce(const float *__restrict__ slab, float *__restrict__ gw, int N, int K,
                                                           int Np, int Kp, int splits) {
  constexpr int KQ = 256 / SG;
  __shared__ float part[SG][KQ][37];
  const int tid = threadIdx.x, kq = tid % KQ, sg = tid / KQ;
  const int n = blockIdx.y, kb = blockIdx.x * (4 * KQ), k4 = kb + kq * 4;
  const size_t pstride = (size_t)Np * Kp, sstride = 16 * pstride;
  const float *p = slab + (size_t)n * Kp + k4;
  f32x4 u[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) u[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (k4 < Kp) {
    for (int s_ = sg; s_ < splits; s_ += SG) {
#pragma unroll
      for (int e = 0; e < 16; ++e) u[e] += *reinterpret_cast<const f32x4 *>(p + (size_t)s_ * sstride + e * pstride);
    }
  }
  // rows: G^T u (3 x 4), G^T = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]; then the same along the columns
  f32x4 tq[3][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 hs = 0.5f * (u[4 + c] + u[8 + c]), hd = 0.5f * (u[4 + c] - u[8 + c]);
    tq[0][c] = u[c] + hs;
    tq[1][c] = hd;
    tq[2][c] = hs + u[12 + c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const f32x4 hs = 0.5f * (tq[r][1] + tq[r][2]), hd = 0.5f * (tq[r][1] - tq[r][2]);
    const f32x4 g0 = tq[r][0] + hs, g1 = hd, g2 = hs + tq[r][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[sg][kq][e * 9 + 3 * r + 0] = g0[e];
      part[sg][kq][e * 9 + 3 * r + 1] = g1[e];
      part[sg][kq][e * 9 + 3 * r + 2] = g2[e];
    }
  }
  __syncthreads();
  // output q of the block's 4 KQ k x 9: k_local = q / 9 -> (kq = k_local >> 2, e = k_local & 3), tap = q % 9
  for (int q = tid; q < 36 * KQ; q += 256) {
    const int kl = q / 9, rc = q - kl * 9;
    if (kb + kl >= K) break;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < SG; ++g) v += part[g][kl >> 2][(kl & 3) * 9 + rc];
    gw[((size_t)n * K + kb) * 9 + q] = v;
  }
}

inline int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
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

// the variant serving N output channels: 0 = 64 ch x 64 tiles (KC 8), 1 = 32 ch x 128 tiles (KC 4)
inline int force_variant() {
  static const int force = getenv("HG_WINO_VARIANT") ? atoi(getenv("HG_WINO_VARIANT")) : -1;
  return force;
}
inline int variant_of(int N) { return force_variant() >= 0 ? force_variant() : (N <= 32 ? 1 : 0); }
struct WinoPlan {
  int variant, NB, TB, KC, nblk, nch;
  int lTW, lTH, lNI, tiles_w, tiles_h, bt_x, bt_y, groups;
  int ksplit;
  long long blocks;   // before the K split
};
inline bool make_plan(int B, int K, int N, int H, int W, WinoPlan &p) {
  if (B <= 0 || K <= 0 || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return false;
  p.variant = variant_of(N);
  p.NB = p.variant ? 32 : 64;
  p.TB = p.variant ? 128 : 64;
  p.KC = p.variant ? 4 : 8;
  if (K % p.KC) return false;
  p.nblk = (N + p.NB - 1) / p.NB;
  p.nch = K / p.KC;
  p.tiles_w = W / 2;
  p.tiles_h = H / 2;
  const int lTB = ceil_log2(p.TB);
  p.lTW = ceil_log2(p.tiles_w);
  if (p.lTW > 4) p.lTW = 4;
  p.lTH = ceil_log2(p.tiles_h);
  if (p.lTH > lTB - p.lTW) p.lTH = lTB - p.lTW;
  p.lNI = lTB - p.lTW - p.lTH;
  p.bt_x = (p.tiles_w + (1 << p.lTW) - 1) >> p.lTW;
  p.bt_y = (p.tiles_h + (1 << p.lTH) - 1) >> p.lTH;
  p.groups = (B + (1 << p.lNI) - 1) >> p.lNI;
  p.blocks = (long long)p.nblk * p.bt_x * p.bt_y * p.groups;
  // 32-bit byte offsets: one block's images, the packed operand
  if ((long long)(1 << p.lNI) * K * H * W >= (1LL << 30)) return false;
  if (16LL * p.nblk * p.nch * ((p.KC / 2) * (p.NB / 32) * 64) >= (1LL << 30)) return false;
  if ((long long)B * N * H * W >= (1LL << 31) || (long long)B * K * H * W >= (1LL << 31)) return false;
  // K split: one block per CU; rounds of the chip in units of one chunk, + ~6 chunks of prologue / epilogue per block
  static const int force_ks = getenv("HG_WINO_KSPLIT") ? atoi(getenv("HG_WINO_KSPLIT")) : 0;
  const int cus = num_cus();
  int best = 1;
  double best_t = 1e300;
  for (int ks = 1; ks <= 16; ++ks) {
    if (ks > 1 && p.nch / ks < 8) break;
    const long long rounds = (p.blocks * ks + cus - 1) / cus;
    double t = (double)rounds * ((p.nch + ks - 1) / ks + 6);
    if (ks > 1) t += 4.0 + 0.02 * ks * rounds;   // slab traffic + the reduce launch
    if (t < best_t * 0.97) { best_t = t; best = ks; }
  }
  p.ksplit = force_ks > 0 ? (force_ks < p.nch ? force_ks : p.nch) : best;
  return true;
}

struct WgPlan {
  int lPW, lPH, lPI, lcg, lrg, nchunks, ktiles, ntiles, splits;
  size_t slab_bytes;
};
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline bool make_wg_plan(int B, int K, int N, int H, int W, WgPlan &p) {
  if (B <= 0 || K <= 0 || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return false;
  const int tw = W / 2, th = H / 2;
  if (!is_pow2(tw) || !is_pow2(th)) return false;
  const int ltw = ceil_log2(tw), lth = ceil_log2(th);
  p.lPW = ltw < 3 ? ltw : 3;
  p.lPH = lth < 3 - p.lPW ? lth : 3 - p.lPW;
  p.lPI = 3 - p.lPW - p.lPH;
  p.lcg = ltw - p.lPW;
  p.lrg = lth - p.lPH;
  const int igroups = (B + (1 << p.lPI) - 1) >> p.lPI;
  const long long nch = (long long)igroups << (p.lcg + p.lrg);
  if (nch > (1 << 30)) return false;
  p.nchunks = (int)nch;
  p.ktiles = (K + 63) / 64;
  p.ntiles = (N + 63) / 64;
  // 32-bit byte offsets inside one image group (+ one map of scalar offset)
  if ((long long)((1 << p.lPI) * (K > N ? K : N) + 1) * H * W >= (1LL << 30)) return false;
  const int tiles = p.ktiles * p.ntiles, cus = num_cus();
  static const int force = getenv("HG_WINO_WG_SPLITS") ? atoi(getenv("HG_WINO_WG_SPLITS")) : 0;
  int s = tiles >= cus ? 1 : (cus + tiles - 1) / tiles;
  if (force > 0) s = force;
  if (s > p.nchunks) s = p.nchunks;
  if (s < 1) s = 1;
  p.splits = s;
  p.slab_bytes = (size_t)s * 16 * (p.ntiles * 64) * (p.ktiles * 64) * sizeof(float);
  return true;
}

template <int TC, int TP, int KC>
int launch_wino(const WinoArgs &a, const WinoPlan &p, bool fe, hipStream_t st) {
  auto kern = fe ? k_wino<TC, TP, KC, true> : k_wino<TC, TP, KC, false>;
  constexpr size_t lds = 2 * 16 * KC * 32 * TP * sizeof(float);
  static bool attr[kMaxDev][2] = {};
  const int dev = cur_dev();
  if (!attr[dev][fe]) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr[dev][fe] = true;
  }
  const long long total = p.blocks * a.ksplit;
  static const int persist = getenv("HG_WINO_PERSIST") ? atoi(getenv("HG_WINO_PERSIST")) : 1;
  const long long grid = persist && total > num_cus() ? num_cus() : total;   // (512 threads, ~215 registers: one per CU)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // namespace

extern "C" {

int hg_wino_supported(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W) {
  WinoPlan p;
  if (!make_plan(B, K, N, H, W, p)) return 0;
  // measured against the direct kernel at the C3 shapes (tools/wino_probe.py, profiles/r05_wino_probe_v5.txt): both variants
  // win from 32 input channels on -- the 64-channel variant 1.27x at 32 -> 64 and 1.5-2.2x from 64 up, the 32-channel variant
  // (twice the transform work per MFMA) 1.15-1.23x at 32 -> 32 and 1.38x at 64 -> 32 @256^2; 16 -> 32 loses (0.8-0.93x)
  static const int min_k0 = getenv("HG_WINO_MIN_K") ? atoi(getenv("HG_WINO_MIN_K")) : 32;
  static const int min_k1 = getenv("HG_WINO_MIN_K1") ? atoi(getenv("HG_WINO_MIN_K1")) : 32;
  if (K < (p.variant ? min_k1 : min_k0)) return 0;
  return p.blocks * p.ksplit >= num_cus() / 2;
}

size_t hg_wino_packed_elems(int32_t Co, int32_t Ci, int32_t mode) {
  if (Co <= 0 || Ci <= 0 || (mode != HG_CONV_PACK_FWD && mode != HG_CONV_PACK_DGRAD)) return 0;
  const PackGeom g = pack_geom(Co, Ci, mode, force_variant());
  if (!g.blocks) return 0;
  return (size_t)16 * g.nblk * g.nch * (g.variant ? 128 : 512);
}

int hg_wino_pack_weights(const float *w, float *u, int32_t Co, int32_t Ci, int32_t mode, void *stream) {
  if (!w || !u || !hg_wino_packed_elems(Co, Ci, mode)) return HG_EINVAL;
  const PackGeom g = pack_geom(Co, Ci, mode, force_variant());
  hipLaunchKernelGGL(k_wino_pack, dim3((unsigned)g.blocks), dim3(512), 0, (hipStream_t)stream, w, u, Co, Ci, mode, force_variant());
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int32_t hg_wino_pack_blocks(int32_t Co, int32_t Ci, int32_t want_fwd, int32_t want_dgrad) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (want_fwd ? pack_geom(Co, Ci, HG_CONV_PACK_FWD, force_variant()).blocks : 0) +
         (want_dgrad ? pack_geom(Co, Ci, HG_CONV_PACK_DGRAD, force_variant()).blocks : 0);
}

int hg_wino_pack_weights_multi(const hg_wino_pack_item *items_dev, int32_t n_items, int32_t total_blocks, void *stream) {
  if (!items_dev || n_items <= 0 || total_blocks <= 0) return HG_EINVAL;
  hipLaunchKernelGGL(k_wino_pack_multi, dim3((unsigned)total_blocks), dim3(512), 0, (hipStream_t)stream, items_dev, n_items,
                     force_variant());
  HG_LAUNCH_CHECK();
  return HG_OK;
}

size_t hg_wino_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W) {
  WinoPlan p;
  if (!make_plan(B, K, N, H, W, p) || p.ksplit == 1) return 0;
  return (size_t)p.ksplit * B * N * H * W * sizeof(float);
}

int hg_wino_conv2d(const float *in, const float *u, float *out, const float *iscale, const float *oscale,
                   const float *bias, const float *noise_w, const float *noise_img, int32_t noise_S, float lrelu_slope,
                   const float *addend, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, void *workspace,
                   size_t workspace_bytes, void *stream) {
  if (!in || !u || !out) return HG_EINVAL;
  if ((noise_img != nullptr) != (noise_w != nullptr) || lrelu_slope < 0.f) return HG_EINVAL;
  if (noise_img && (noise_S < H || noise_S < W || (noise_S & 1))) return HG_EINVAL;
  if (addend && (iscale || oscale || noise_img || lrelu_slope > 0.f)) return HG_EINVAL;
  WinoPlan p;
  if (!make_plan(B, K, N, H, W, p)) return HG_EUNSUPPORTED;
  if (p.ksplit > 1 && (!workspace || workspace_bytes < (size_t)p.ksplit * B * N * H * W * sizeof(float))) p.ksplit = 1;
  WinoArgs a;
  a.in = in; a.u = u; a.out = out; a.iscale = iscale; a.oscale = oscale; a.bias = bias; a.addend = addend;
  a.noise_w = noise_w; a.noise_img = noise_img; a.noise_S = noise_S; a.slope = lrelu_slope;
  a.B = B; a.K = K; a.N = N; a.H = H; a.W = W;
  a.nblk = p.nblk; a.nch = p.nch; a.lTW = p.lTW; a.lTH = p.lTH; a.lNI = p.lNI;
  a.tiles_w = p.tiles_w; a.tiles_h = p.tiles_h; a.bt_x = p.bt_x; a.bt_y = p.bt_y;
  a.ksplit = p.ksplit; a.slab = (float *)workspace;
  a.blocks = (int)p.blocks; a.total_tiles = (int)(p.blocks * p.ksplit);
  static const int xcd = getenv("HG_WINO_XCD") ? atoi(getenv("HG_WINO_XCD")) : 1;
  a.xcd = xcd && p.nblk > 1 && p.nblk < 8 && (p.blocks / p.nblk) % 8 == 0;
  hipStream_t st = (hipStream_t)stream;
  const bool fe = iscale != nullptr;
  int rc = p.variant == 0 ? launch_wino<2, 2, 8>(a, p, fe, st) : launch_wino<1, 4, 4>(a, p, fe, st);
  if (rc) return rc;
  if (p.ksplit > 1) {
    const long long total = (long long)B * N * H * W;   // H, W even -> a multiple of 4
    long long nb = (total / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_wino_reduce, dim3((unsigned)nb), dim3(256), 0, st, a.slab, out, oscale, bias, noise_w, noise_img,
                       addend, noise_S, lrelu_slope, total / 4, H * W, W, N, p.ksplit);
    HG_LAUNCH_CHECK();
  }
  return HG_OK;
}

size_t hg_wino_wgrad_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W) {
  WgPlan p;
  if (!make_wg_plan(B, K, N, H, W, p)) return 0;
  return p.slab_bytes;
}

int hg_wino_wgrad_supported(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W) {
  WgPlan p;
  if (!make_wg_plan(B, K, N, H, W, p)) return 0;
  // 64 x 64 channel tiles: a layer with fewer channels on a side multiplies padding (and the transforms, done once per
  // 64 x 64 tile, stop amortising: 0.75x at 32 -> 64); 2x2 maps are slab traffic rather than arithmetic (0.84-0.95x);
  // measured 1.25-1.9x elsewhere (tools/wino_probe.py, profiles/r05_wino_probe_v5.txt)
  static const int min_c = getenv("HG_WINO_WG_MIN_C") ? atoi(getenv("HG_WINO_WG_MIN_C")) : 64;
  static const int min_s = getenv("HG_WINO_WG_MIN_S") ? atoi(getenv("HG_WINO_WG_MIN_S")) : 4;
  if (K < min_c || N < min_c || H < min_s || W < min_s) return 0;
  return p.nchunks / p.splits >= 4;
}

int hg_wino_wgrad(const float *in, const float *gout, float *gw, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W,
                  void *workspace, size_t workspace_bytes, void *stream) {
  if (!in || !gout || !gw || !workspace) return HG_EINVAL;
  WgPlan p;
  if (!make_wg_plan(B, K, N, H, W, p)) return HG_EUNSUPPORTED;
  if (workspace_bytes < p.slab_bytes) return HG_EWORKSPACE;
  WinoWgArgs a;
  a.in = in; a.gout = gout; a.slab = (float *)workspace;
  a.B = B; a.K = K; a.N = N; a.H = H; a.W = W;
  a.ktiles = p.ktiles; a.splits = p.splits; a.nchunks = p.nchunks;
  a.lPW = p.lPW; a.lPH = p.lPH; a.lPI = p.lPI; a.lcg = p.lcg; a.lrg = p.lrg;
  a.Np = p.ntiles * 64; a.Kp = p.ktiles * 64;
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t lds = 4 * (size_t)WG_OP * sizeof(float);
  static bool attr[kMaxDev] = {};
  const int dev = cur_dev();
  if (!attr[dev]) {
    hipError_t e = hipFuncSetAttribute((const void *)k_wino_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr[dev] = true;
  }
  a.xtiles = p.ktiles * p.ntiles; a.total_blocks = a.xtiles * p.splits;
  static const int xcd = getenv("HG_WINO_XCD") ? atoi(getenv("HG_WINO_XCD")) : 1;
  a.xcd = xcd && p.splits % 8 == 0;
  static const int persist = getenv("HG_WINO_PERSIST") ? atoi(getenv("HG_WINO_PERSIST")) : 1;
  const int grid = persist && a.total_blocks > num_cus() ? num_cus() : a.total_blocks;
  hipLaunchKernelGGL(k_wino_wgrad, dim3((unsigned)grid), dim3(512), lds, st, a);
  HG_LAUNCH_CHECK();
  if (p.splits >= 16)
    hipLaunchKernelGGL(k_wino_wgrad_reduce<32>, dim3((unsigned)((a.Kp + 31) / 32), (unsigned)N), dim3(256), 0, st, a.slab, gw, N, K,
                       a.Np, a.Kp, p.splits);
  else if (p.splits >= 4)
    hipLaunchKernelGGL(k_wino_wgrad_reduce<8>, dim3((unsigned)((a.Kp + 127) / 128), (unsigned)N), dim3(256), 0, st, a.slab, gw, N, K,
                       a.Np, a.Kp, p.splits);
  else
    hipLaunchKernelGGL(k_wino_wgrad_reduce<1>, dim3((unsigned)((a.Kp + 1023) / 1024), (unsigned)N), dim3(256), 0, st, a.slab, gw, N,
                       K, a.Np, a.Kp, p.splits);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
