// hg_hist.hip -- RGB-uv colour histogram, forward + backward, hand-written for gfx950.
//
// Replaces the ~80-launch-per-image aten chain of the reference
// (histogram_classes/RGBuvHistBlock.py:75-228) and its autograd replay by
//   forward : k_hist_fwd  (MFMA split-K over pixels)  -> k_hist_finish (slab sum + normalisation)
//   backward: k_hist_bwd (MFMA, mirrored-bin merge: symmetric boundary, h <= 64) or k_hist_bwd_planes (MFMA, one plane
//             at a time: any boundary, h <= 128, one-plane projections) [-> resize adjoint]; k_hist_bwd_generic beyond
//   method = thresholding: k_thr_fwd_lean -> k_hist_finish / k_thr_bwd_lean (true scatter-add; HBM / VALU bound);
//   method = RBF with a narrow kernel: k_hist_rbf_fwd / k_hist_rbf_bwd (truncated scatter / gather)
//
// Maths:  with L_c = log(x_c + 1e-6), a = L_R-L_G, b = L_R-L_B, c = L_G-L_B,
// Iy = sqrt(R^2+G^2+B^2+1e-6), k(.) the soft-bin kernel and bins b_i = linspace(lo,hi,h):
//   plane0[i][j] = sum_n Iy k(a-b_i)  k(b-b_j)        (u,v) = ( a,  b)   RGBuvHistBlock.py:112-148
//   plane1[i][j] = sum_n Iy k(-a-b_i) k(c-b_j)        (u,v) = (-a,  c)   RGBuvHistBlock.py:150-187
//   plane2[i][j] = sum_n Iy k(-b-b_i) k(-c-b_j)       (u,v) = (-b, -c)   RGBuvHistBlock.py:190-222
// Each plane is a rank-N contraction  (Iy*Ku)^T @ Kv  (h x N x h)  -> v_mfma_f32_32x32x2_f32 with
// BOTH operands generated on the fly from 16 B of per-pixel state (a,b,c,Iy).  Because both MFMA
// operands use the same lane->(bin = lane&31, pixel = lane>>5) map, one evaluated kernel value can
// feed the A side of one plane and the B side of another.  With the mirrored bin table
// b'_i = -b_(h-1-i) we have k(-a-b_i) = k(a-b'_(h-1-i)), so planes 1 and 2 are accumulated with
// un-negated variables into index-flipped tiles and flipped back when the slab is written; for the
// default symmetric boundary b' == b and only 3 kernel vectors per pixel are evaluated instead of 6.
//
// Precision: u is fp32 exactly as in the reference (fp32 log - fp32 log); the reference then
// evaluates |u-b_i| in fp64.  Here t=(u-b_i)/sigma is formed as fma(u, 1/sigma, c_hi)+c_lo with
// (c_hi,c_lo) a double-single split of -b_i/sigma, which keeps t to ~1e-7 relative without fp64;
// thresholding compares in fp64 so its 0/1 weights are bit-identical to the reference's.
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (no reduced-precision path on gfx950).
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include <cstdlib>

#define HG_VERSION_NUM 102   // 102: hg_hist_params.struct_size (ABI guard), hg_rgbuv_hist_uses_proj_cache

#ifndef HG_FWD_SCHED_GROUPS
#define HG_FWD_SCHED_GROUPS 1
#endif
#ifndef HG_FWD_VALU_PER_MFMA
#define HG_FWD_VALU_PER_MFMA 4
#endif
#ifndef HG_FWD_PACKED
#define HG_FWD_PACKED 1   // packed-fp32 (v_pk_*) operand generation in k_hist_fwd
#endif
#ifndef HG_FWD_SHARE_ASM
#define HG_FWD_SHARE_ASM 1   // shared-reciprocal operand generation of k_hist_fwd as one block of packed instructions
#endif
#ifndef HG_FWD_MFMA_GROUP
#define HG_FWD_MFMA_GROUP 12  // k_hist_fwd at configs[1]: groups of 1: 505 us, 3: 498, 6: 473, 12: 465
#endif
#ifndef HG_BWD_SCHED_GROUPS
#define HG_BWD_SCHED_GROUPS 1
#endif
#ifndef HG_BWD_MFMA_GROUP
#define HG_BWD_MFMA_GROUP 1   // k_hist_bwd K loop: MFMAs per group between the VALU work of the next step (1: round 2's 1 : 2 interleave)
#endif
#ifndef HG_BWD_SCHED_BARRIER
#define HG_BWD_SCHED_BARRIER 1
#endif
#ifndef HG_BWD_WAVES
#define HG_BWD_WAVES 2  // min waves/SIMD the backward kernel is register-budgeted for
#endif
#if HG_BWD_WAVES > 2
#error "HG_BWD_WAVES > 2: the shared-reciprocal assembly of k_hist_bwd clobbers v[240:255] (needs the 256-VGPR budget)"
#endif

#ifndef HG_HIST_PROBE
#define HG_HIST_PROBE 0      // 1 (tagged experiment builds only): per-phase shader-cycle counters in k_hist_fwd / k_hist_bwd
#endif

namespace {

#if HG_HIST_PROBE
// [kernel 0 fwd / 1 bwd][0 waves, 1 total, 2 prologue, 3 projection / pixel state, 4 K loops, 5 epilogue, 6 longest wave, 7 spare]
__device__ unsigned long long hg_probe[2][8];
#define HG_PROBE_T(var) const long long var = __builtin_readcyclecounter()
#else
#define HG_PROBE_T(var)
#endif

constexpr float kEps = 1e-6f;  // RGBuvHistBlock.py:26
// k_hist_fwd's per-wave staging row: 64 pixels + 4 entries of slack -- the K loop prefetches the (a, b, c, weight) tuple two
// steps ahead without clamping the index (the operands made from entries past the batch are never multiplied)
constexpr int kFwdStage = 68;

struct DevParams {
  int B, C, H, W;
  long long sb, sc, sh, sw;
  int Hs, Ws, mode;
  const int *rows, *cols;
  int h, P, method, intensity, green;
  int proj;                 // HG_PROJ_*: 0 RGB-uv (3 planes), 1 rg-chroma, 2 direct (Lab): one plane, run as `green`
  int pre_relu;             // the caller's F.relu in front of the block (histoGAN.py:955) folded into the clamp mask
  float4 *cache;            // optional [B][npix][2] float4: (a, b, c, Iy), (r, g, b, -) written by the forward, read by the backward
  int npix;
  double lo, hi, step;      // bins: i*step+lo, last == hi  (np.linspace)
  double inv_sigma_d;       // (double)(float)(1/sigma) -- pairs with inv_sigma
  float inv_sigma;
  double half_eps;          // thresholding: eps/2, eps=(|lo|+|hi|)/h  (RGBuvHistBlock.py:70-71,124)
  float rscale_h, rscale_w; // bilinear: (float)H/Hs, (float)W/Ws
  // backward: t = (u-lo)/sigma - i*(step/sigma), step/sigma split so that i*ds_hi is exact
  double inv_sigma_x;       // exact 1/sigma in double
  float ds_hi, ds_lo;
  float dk_scale;           // -2/sigma
};

__device__ __forceinline__ double bin_center(const DevParams &P, int i) {
  return (i == P.h - 1) ? P.hi : (double)i * P.step + P.lo;
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// Gradient mask of the clamp (RGBuvHistBlock.py:76: passes where 0 <= x <= 1).  With pre_relu the F.relu the train
// step applies first (histoGAN/histoGAN.py:955) is part of the block: clamp(relu(x)) == clamp(x) in the forward, and
// the relu's mask (x > 0) only removes the point x == 0 from the clamp's.
__device__ __forceinline__ bool grad_mask(const DevParams &P, float raw) {
  return (P.pre_relu ? raw > 0.f : raw >= 0.f) && raw <= 1.f;
}

// Stage 0 (clamp + resize), RGBuvHistBlock.py:76-99.  n indexes the Hs x Ws sampled grid.
__device__ __forceinline__ void sample_rgb(const DevParams &P, const float *xb, int n, float &r,
                                           float &g, float &b) {
  const int ys = n / P.Ws, xs = n - ys * P.Ws;
  if (P.mode == HG_RESIZE_BILINEAR) {
    // aten upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
    // (mul, then sub, each rounded -- as aten's area_pixel_compute_source_index<float>; an fma here
    //  moves lambda by up to 1 ulp(src) ~ 1.5e-5 and the interpolated value by ~5e-6)
    float sy = fmaxf(__fsub_rn(__fmul_rn(P.rscale_h, (float)ys + 0.5f), 0.5f), 0.f);
    float sx = fmaxf(__fsub_rn(__fmul_rn(P.rscale_w, (float)xs + 0.5f), 0.5f), 0.f);
    const int y0 = min((int)sy, P.H - 1), x0 = min((int)sx, P.W - 1);
    const float ly = clamp01(sy - (float)y0), lx = clamp01(sx - (float)x0);
    const int y1 = y0 + (y0 < P.H - 1 ? 1 : 0), x1 = x0 + (x0 < P.W - 1 ? 1 : 0);
    const long long o00 = y0 * P.sh + x0 * P.sw, o01 = y0 * P.sh + x1 * P.sw;
    const long long o10 = y1 * P.sh + x0 * P.sw, o11 = y1 * P.sh + x1 * P.sw;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float *pc = xb + c * P.sc;
      const float p00 = clamp01(pc[o00]), p01 = clamp01(pc[o01]);
      const float p10 = clamp01(pc[o10]), p11 = clamp01(pc[o11]);
      // 4 combined weights, fma chain: the form that reproduces aten's CPU kernel bit-for-bit on
      // ~90 % of pixels (the separable forms match ~60 %; differences are 1 ulp of the pixel value)
      const float wx0 = 1.f - lx, wy0 = 1.f - ly;
      v[c] = fmaf(__fmul_rn(ly, lx), p11, fmaf(__fmul_rn(ly, wx0), p10,
                  fmaf(__fmul_rn(wy0, lx), p01, __fmul_rn(__fmul_rn(wy0, wx0), p00))));
    }
    r = v[0]; g = v[1]; b = v[2];
  } else {
    int yy = ys, xx = xs;
    if (P.mode == HG_RESIZE_SAMPLING) { yy = P.rows[ys]; xx = P.cols[xs]; }
    const long long o = yy * P.sh + xx * P.sw;
    r = clamp01(xb[o]); g = clamp01(xb[o + P.sc]); b = clamp01(xb[o + 2 * P.sc]);
  }
}

// Stage 1 (projection), RGBuvHistBlock.py:104-115: three logs and three chroma differences.
__device__ __forceinline__ void project(const DevParams &P, float r, float g, float b, float &a,
                                        float &bb, float &c, float &iy) {
  if (P.proj == HG_PROJ_RGCHROMA) {
    // rgChromaHistBlock.py:100-110: (u, v) = (R, G) / (R+G+B + eps), weight Iy.  The single plane runs through the
    // `green` path, which bins (-a, c): a = -u, c = v.
    const float s = __fadd_rn(__fadd_rn(__fadd_rn(r, g), b), kEps);
    a = -__fdiv_rn(r, s); c = __fdiv_rn(g, s); bb = 0.f;
    iy = P.intensity ? __fsqrt_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(g, g)), __fmul_rn(b, b)), kEps)) : 1.f;
    return;
  }
  if (P.proj == HG_PROJ_DIRECT) {
    // LabHistBlock.py:102-109: (u, v) = channels (1, 2), weight = channel 0 (L) when intensity_scale
    a = -g; c = b; bb = 0.f;
    iy = P.intensity ? r : 1.f;
    return;
  }
  // The reference's CPU logf is correctly rounded for >99.9% of inputs (probe); an fp64 log rounded
  // to fp32 reproduces it, where a 1-ulp device logf would perturb u by ~1e-7 and, through the
  // ~50x sensitivity of k at |u-b| = sigma, move single weights by ~5e-6.  3 fp64 logs per pixel
  // are noise next to 384+ MFMA cycles per pixel.
  const float lr = (float)log((double)__fadd_rn(r, kEps));
  const float lg = (float)log((double)__fadd_rn(g, kEps));
  const float lb = (float)log((double)__fadd_rn(b, kEps));
  a = __fsub_rn(lr, lg); bb = __fsub_rn(lr, lb); c = __fsub_rn(lg, lb);
  // pow(I,2) summed left to right in fp32, no fma contraction (RGBuvHistBlock.py:105-108)
  iy = P.intensity ? __fsqrt_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(g, g)), __fmul_rn(b, b)), kEps)) : 1.f;
}

// Per-pixel state of the backward kernels: from the forward's cache (two 16-byte loads) or recomputed (up to 12 taps of
// the bilinear resize + three fp64 logarithms: ~370 VALU instructions per pixel that the MFMA-bound kernels pay for in
// matrix-pipe time -- fp32 MFMA and VALU instructions do not overlap on gfx950, tools/ubench/mfma_valu_overlap.hip).
__device__ __forceinline__ void pixel_state(const DevParams &P, const float *xb, int b, int n, bool valid, float &r,
                                            float &g, float &bl, float &a, float &bb, float &c, float &iy) {
  if (P.cache) {
    const float4 *cp = P.cache + ((long long)b * P.npix + (valid ? n : 0)) * 2;
    const float4 u = cp[0], v = cp[1];
    a = u.x; bb = u.y; c = u.z; iy = u.w; r = v.x; g = v.y; bl = v.z;
    if (!valid) { r = g = bl = 0.f; project(P, r, g, bl, a, bb, c, iy); }
    return;
  }
  r = g = bl = 0.f;
  if (valid) sample_rgb(P, xb, n, r, g, bl);
  project(P, r, g, bl, a, bb, c, iy);
}

struct BinC { float chi, clo; double bd; };

// constants of bin i for the direct table (mirror=false) or the mirrored table b'_i = -b_(h-1-i)
__device__ __forceinline__ BinC make_binc(const DevParams &P, int i, bool mirror) {
  double bd = mirror ? -bin_center(P, P.h - 1 - i) : bin_center(P, i);
  const double c = -bd * P.inv_sigma_d;
  BinC r; r.chi = (float)c; r.clo = (float)(c - (double)r.chi); r.bd = bd;
  return r;
}

template <int METHOD>
__device__ __forceinline__ float kern_eval(const DevParams &P, float u, const BinC &bc) {
  if constexpr (METHOD == HG_METHOD_THRESHOLDING) {
    return (fabs((double)u - bc.bd) <= P.half_eps) ? 1.f : 0.f;
  } else {
    const float t = fmaf(u, P.inv_sigma, bc.chi) + bc.clo;
    if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) return __builtin_amdgcn_rcpf(fmaf(t, t, 1.f));
    else return expf(-(t * t));
  }
}

// MFMA operands of one K step (2 pixels): A side carries the Iy weight, B side is the bare kernel.
template <int T>
struct Ops { float A0[T], A1[T], A2[T], B0[T], B1[T], B2[T]; };

// backward: operands of one K step (2 bins): 6*T A values (Ghat from LDS) + the 3 kernel values
template <int T>
struct BOps { float A[T][6]; float ka, kb, kc; };

__device__ __forceinline__ void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// Forward.  grid = (S splits, nbd*nbd output blocks, B images), 256 threads = 4 independent waves.
// Each wave owns a contiguous run of `chunk` pixels and accumulates a (3 x BLK x BLK) partial
// histogram block (BLK = 32*T) in 3*T*T MFMA accumulator tiles; the 4 waves are then summed through
// LDS in fixed order and written as one slab  slabs[b][s][p][h][h]  (real bin order, flips undone).
template <int T, int METHOD, bool SYM, bool DIAG, bool GREEN, bool SHARE = false>
__global__ __launch_bounds__(256, 2) void k_hist_fwd(const DevParams P, const float *__restrict__ x,
                                                     float *__restrict__ slabs, double *__restrict__ slab_tot,
                                                     const int chunk) {
  constexpr int BLK = 32 * T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *stage = reinterpret_cast<float4 *>(smem);            // [4 waves][kFwdStage pixels]
  float *red = reinterpret_cast<float *>(smem + 4 * kFwdStage * 16);  // [3][BLK][BLK]

  // wave-uniform values in scalar registers: loop bounds and addresses derived from them cost no VALU issue slots
  // (which the MFMAs of this kernel pay for)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = lane >> 5, l31 = lane & 31;
  const int nbd = (P.h + BLK - 1) / BLK;
  const int bi = blockIdx.y / nbd, bj = blockIdx.y - bi * nbd;
  const int b = blockIdx.z, s = blockIdx.x, S = gridDim.x;
  const float *xb = x + (long long)b * P.sb;

  HG_PROBE_T(pt0);
#if HG_HIST_PROBE
  long long p_proj = 0, p_loop = 0;
#endif
  // per-lane bin constants: A side = rows (i) of this block, B side = columns (j)
  BinC cA[T], cAm[T], cB[T], cBm[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = BLK * bi + 32 * t + l31, j = BLK * bj + 32 * t + l31;
    cA[t] = make_binc(P, i, false);
    cB[t] = make_binc(P, j, false);
    cAm[t] = make_binc(P, i, !SYM);
    cBm[t] = make_binc(P, j, !SYM);
  }

  f32x16 acc[3][T][T];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = 0; tj < T; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][ti][tj][r] = 0.f;

  const long long start = (long long)(s * 4 + wave) * chunk;
  const int end = (int)min((long long)P.npix, start + chunk);
  constexpr bool green = GREEN;

  // Packed-fp32 operand generation (inverse-quadratic, two tiles per side): the two tiles' evaluations of one variable are
  // one v_pk_fma / v_pk_add / v_pk_fma (+ two v_rcp) instead of two of each -- the same IEEE operations lane for lane,
  // so bit-identical values; 17 VALU instructions per K step instead of 28.  (VALU work is what the fp32 MFMA cannot
  // hide on gfx950.)
  [[maybe_unused]] f32x2 chiA, cloA, chiAm, cloAm, chiB, cloB, chiBm, cloBm;
  if constexpr (T == 2) {
    chiA = f32x2{cA[0].chi, cA[1].chi}; cloA = f32x2{cA[0].clo, cA[1].clo};
    chiAm = f32x2{cAm[0].chi, cAm[1].chi}; cloAm = f32x2{cAm[0].clo, cAm[1].clo};
    chiB = f32x2{cB[0].chi, cB[1].chi}; cloB = f32x2{cB[0].clo, cB[1].clo};
    chiBm = f32x2{cBm[0].chi, cBm[1].chi}; cloBm = f32x2{cBm[0].clo, cBm[1].clo};
  }
  auto iq2 = [&](float u, const f32x2 &chi, const f32x2 &clo) __attribute__((always_inline)) -> f32x2 {
    const f32x2 uu = {u, u}, is = {P.inv_sigma, P.inv_sigma}, one = {1.f, 1.f};
    const f32x2 t = __builtin_elementwise_fma(uu, is, chi) + clo;
    const f32x2 den = __builtin_elementwise_fma(t, t, one);
    return f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  };
  // denominators 1 + t^2 of the two tiles of one variable (no reciprocal yet)
  auto den2 = [&](float u, const f32x2 &chi, const f32x2 &clo) __attribute__((always_inline)) -> f32x2 {
    const f32x2 uu = {u, u}, is = {P.inv_sigma, P.inv_sigma}, one = {1.f, 1.f};
    const f32x2 t = __builtin_elementwise_fma(uu, is, chi) + clo;
    return __builtin_elementwise_fma(t, t, one);
  };
  auto make_ops = [&](const f32x4 &q, Ops<T> &o) {
    if constexpr (SHARE) {
      // Shared reciprocals (round 4; PMC: profiles/r04_hist_pmc_stalls.txt).  fp32 MFMA and VALU instructions share
      // the issue path on gfx950, and v_rcp_f32 runs at a quarter of the VALU rate: the six reciprocals of a K step were
      // 96 of its ~1 035 cycles.  1/a and 1/b from ONE reciprocal: r = 1/(a b), 1/a = b r, 1/b = a r -- here for the four
      // denominators of (ka, kb) at once (r = 1/(a0 b0 a1 b1)) and the two of kc: 2 v_rcp + 7 packed multiplies instead
      // of 6 v_rcp + 2.  Each value picks up ~3 roundings instead of 1 (<= 2e-7 relative; the parity bars are 1e-5).
      // The launcher takes this instantiation only when (1 + t_max^2)^4 stays far below the fp32 range.
      static_assert(T == 2 && METHOD == HG_METHOD_INVERSE_QUADRATIC && SYM && DIAG && !GREEN, "shared-reciprocal path");
#if HG_FWD_SHARE_ASM
      // Written as ONE block of packed instructions: clang 22 "unpacks" v_pk_* instructions that follow an MFMA into two
      // scalar ones (a peephole meant for the 16-bit MFMAs, whose shadow hides VALU work) -- on the fp32 MFMA, which
      // shares the fp32 lanes with the VALU, that doubles their cost.  20 VALU instructions + 2 v_rcp_f32 per K step;
      // v[244:255] are the block's temporaries.  s_nop 0: trans result -> next VALU read; s_nop 1: VALU write -> MFMA read.
      const f32x2 qxy = __builtin_shufflevector(q, q, 0, 1), qzw = __builtin_shufflevector(q, q, 2, 3);
      const f32x2 is2 = {P.inv_sigma, P.inv_sigma};
      f32x2 a0, a2, b0, b1;
      asm volatile(
          "v_pk_fma_f32 v[244:245], %4, %6, %7 op_sel_hi:[0,1,1]\n\t"                    // t(a) = a / sigma + c_hi
          "v_pk_fma_f32 v[246:247], %4, %6, %9 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"     // t(b)
          "v_pk_fma_f32 v[248:249], %5, %6, %11 op_sel_hi:[0,1,1]\n\t"                   // t(c)
          "v_pk_add_f32 v[244:245], v[244:245], %8\n\t"                                   // + c_lo
          "v_pk_add_f32 v[246:247], v[246:247], %10\n\t"
          "v_pk_add_f32 v[248:249], v[248:249], %12\n\t"
          "v_pk_fma_f32 v[244:245], v[244:245], v[244:245], 1.0 op_sel_hi:[1,1,0]\n\t"   // da = 1 + t^2
          "v_pk_fma_f32 v[246:247], v[246:247], v[246:247], 1.0 op_sel_hi:[1,1,0]\n\t"   // db
          "v_pk_fma_f32 v[248:249], v[248:249], v[248:249], 1.0 op_sel_hi:[1,1,0]\n\t"   // dc
          "v_pk_mul_f32 v[250:251], v[244:245], v[246:247]\n\t"                           // (a0 b0, a1 b1)
          "v_mul_f32 v253, v248, v249\n\t"                                                // c0 c1
          "v_mul_f32 v252, v250, v251\n\t"                                                // a0 b0 a1 b1
          "v_rcp_f32 v253, v253\n\t"
          "v_rcp_f32 v252, v252\n\t"
          "s_nop 0\n\t"
          "v_pk_mul_f32 %3, v[248:249], v[252:253] op_sel:[1,1] op_sel_hi:[0,1]\n\t"      // kc = (c1, c0) / (c0 c1)
          "v_pk_mul_f32 v[250:251], v[250:251], v[252:253] op_sel:[1,0] op_sel_hi:[0,0]\n\t"  // (1/(a0 b0), 1/(a1 b1))
          "v_pk_mul_f32 v[254:255], v[250:251], %5 op_sel:[0,1] op_sel_hi:[1,1]\n\t"      // ... times the weight
          "v_pk_mul_f32 %2, v[244:245], v[250:251]\n\t"                                   // kb = da / (da db)
          "v_pk_mul_f32 %0, v[246:247], v[254:255]\n\t"                                   // w ka
          "v_pk_mul_f32 %1, v[244:245], v[254:255]\n\t"                                   // w kb
          "s_nop 1"
          : "=&v"(a0), "=&v"(a2), "=&v"(b0), "=&v"(b1)
          : "v"(qxy), "v"(qzw), "s"(is2), "v"(chiA), "v"(cloA), "v"(chiAm), "v"(cloAm), "v"(chiB), "v"(cloB)
          : "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
#else
      const f32x2 da = den2(q.x, chiA, cloA), db = den2(q.y, chiAm, cloAm), dc = den2(q.z, chiB, cloB);
      const f32x2 pab = da * db;                                    // (a0 b0, a1 b1)
      const float r4 = __builtin_amdgcn_rcpf(pab.x * pab.y);
      const f32x2 rab = f32x2{pab.y, pab.x} * f32x2{r4, r4};        // (1/(a0 b0), 1/(a1 b1))
      const f32x2 wr = rab * f32x2{q.w, q.w};
      const f32x2 a0 = db * wr, a2 = da * wr;                       // w ka, w kb
      const f32x2 b0 = da * rab;                                    // kb
      const float r2 = __builtin_amdgcn_rcpf(dc.x * dc.y);
      const f32x2 b1 = f32x2{dc.y, dc.x} * f32x2{r2, r2};           // kc
#endif
      o.A0[0] = a0.x; o.A0[1] = a0.y; o.A1[0] = a0.x; o.A1[1] = a0.y; o.A2[0] = a2.x; o.A2[1] = a2.y;
      o.B0[0] = b0.x; o.B0[1] = b0.y; o.B1[0] = b1.x; o.B1[1] = b1.y; o.B2[0] = b1.x; o.B2[1] = b1.y;
      return;
    }
    if constexpr (T == 2 && METHOD == HG_METHOD_INVERSE_QUADRATIC && HG_FWD_PACKED) {
      const f32x2 w2 = {q.w, q.w};
      const f32x2 ka = iq2(q.x, chiA, cloA), kbA = iq2(q.y, chiAm, cloAm);
      const f32x2 a0 = w2 * ka, a2 = w2 * kbA;
      const f32x2 a1 = SYM ? a0 : w2 * iq2(q.x, chiAm, cloAm);
      const f32x2 b0 = (SYM && DIAG) ? kbA : iq2(q.y, chiB, cloB);
      const f32x2 b1 = iq2(q.z, chiB, cloB);
      const f32x2 b2 = SYM ? b1 : iq2(q.z, chiBm, cloBm);
      o.A0[0] = a0.x; o.A0[1] = a0.y; o.A1[0] = a1.x; o.A1[1] = a1.y; o.A2[0] = a2.x; o.A2[1] = a2.y;
      o.B0[0] = b0.x; o.B0[1] = b0.y; o.B1[0] = b1.x; o.B1[1] = b1.y; o.B2[0] = b2.x; o.B2[1] = b2.y;
      return;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float ka = kern_eval<METHOD>(P, q.x, cA[t]);
      const float kbA = kern_eval<METHOD>(P, q.y, cAm[t]);
      o.A0[t] = q.w * ka;
      o.A1[t] = SYM ? o.A0[t] : q.w * kern_eval<METHOD>(P, q.x, cAm[t]);
      o.A2[t] = q.w * kbA;
      o.B0[t] = (SYM && DIAG) ? kbA : kern_eval<METHOD>(P, q.y, cB[t]);
      o.B1[t] = kern_eval<METHOD>(P, q.z, cB[t]);
      o.B2[t] = SYM ? o.B1[t] : kern_eval<METHOD>(P, q.z, cBm[t]);
    }
  };

  // (Round 3 moved the projection -- clamp / resize / three fp64 logarithms, ~370 VALU instructions per pixel, a third of
  // this kernel's VALU work -- into a pre-pass kernel of its own: k_hist_fwd got 2 % faster (455 -> 446 us at configs[1]),
  // the pre-pass cost 23 us.  The other wave of the SIMD evidently hides most of it here; it stays in the kernel.)
  float r_ = 0.f, g_ = 0.f, b_ = 0.f;
  if (start + lane < end) sample_rgb(P, xb, (int)start + lane, r_, g_, b_);
  HG_PROBE_T(pt1);
  for (int base = (int)start; base < end; base += 64) {
    HG_PROBE_T(pb0);
    float a, bb, c, iy;
    project(P, r_, g_, b_, a, bb, c, iy);
    const bool valid = base + lane < end;
    stage[wave * kFwdStage + lane] = valid ? make_float4(a, bb, c, iy) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.cache && valid && blockIdx.y == 0) {          // one writer per pixel (the bin-block replicas skip it)
      float4 *cp = P.cache + ((long long)b * P.npix + base + lane) * 2;
      cp[0] = make_float4(a, bb, c, iy);
      cp[1] = make_float4(r_, g_, b_, 0.f);
    }
    // prefetch the next 64 pixels while this batch is in the MFMA loop
    if (base + 64 + lane < end) sample_rgb(P, xb, base + 64 + lane, r_, g_, b_);
    lds_wave_sync();
    // K steps of this batch, rounded up to an even count: entries past the batch hold zero weights (written above), so
    // the padding step adds exact zeros and the loop body needs no branch between its two steps
    const int steps = (((min(64, end - base) + 1) >> 1) + 1) & ~1;
    HG_PROBE_T(pb1);
    // Software-pipelined K loop: the operands of step m+1 (VALU: 3..6 kernel vectors) are generated next to the
    // 3*T*T MFMAs of step m.
    // one K step: the MFMAs of `use` while the operands of the following step are generated into `gen` from qn; the
    // float4 two steps ahead is fetched from LDS meanwhile (its latency is covered by a whole step of MFMAs even when
    // the two waves of a SIMD run phase-locked)
    const f32x4 *srow = reinterpret_cast<const f32x4 *>(stage) + wave * kFwdStage + half;
    auto kstep = [&](const Ops<T> &use, Ops<T> &gen, const f32x4 &qn, f32x4 &qf, int mf) __attribute__((always_inline)) {
      qf = srow[2 * mf];                                   // {a, b, c, weight}; broadcast per half-wave
      if constexpr (!(SHARE && HG_FWD_SHARE_ASM)) make_ops(qn, gen);
#pragma unroll
      for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = 0; tj < T; ++tj) {
          if (!green) acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.A0[ti], use.B0[tj], acc[0][ti][tj], 0, 0, 0);
          acc[1][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.A1[ti], use.B1[tj], acc[1][ti][tj], 0, 0, 0);
          if (!green) acc[2][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.A2[ti], use.B2[tj], acc[2][ti][tj], 0, 0, 0);
        }
      if constexpr (SHARE && HG_FWD_SHARE_ASM) {
        // the LDS read, the 12 MFMAs as one group, then the operand block of the next step: nothing crosses the fences
        __builtin_amdgcn_sched_barrier(0);
        make_ops(qn, gen);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(qf));
        return;
      }
#if HG_FWD_SCHED_GROUPS
      // MFMAs in groups of HG_FWD_MFMA_GROUP with the operand generation of the next step between the groups.  On
      // gfx950 the fp32 MFMA does not hide VALU work (same issue path: tools/ubench/mfma_valu_overlap.hip -- 3 VALU
      // instructions per MFMA cost 0.79 of peak interleaved 1:3 and 0.86 as 6 MFMAs : 18 VALU), so what is left to
      // schedule is the number of MFMA <-> VALU switches.
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      constexpr int NM = (green ? 1 : 3) * T * T, GRP = HG_FWD_MFMA_GROUP < NM ? HG_FWD_MFMA_GROUP : NM;
#pragma unroll
      for (int i = 0; i < NM; i += GRP) {
        __builtin_amdgcn_sched_group_barrier(0x008, GRP, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, HG_FWD_VALU_PER_MFMA * GRP, 0);
      }
#endif
      // pin: qf is complete here (after a step's worth of MFMA issue) -- as ONE 128-bit tuple, so that the packed
      // operand generation can address its halves in place (op_sel) instead of through copies
      asm volatile("" : "+v"(qf));
    };
    // two steps per iteration with the operand sets swapping roles: no register copies between steps
    Ops<T> opA, opB;
    make_ops(srow[0], opA);
    f32x4 qa = srow[2], qb;
    for (int m = 0; m < steps; m += 2) {
      kstep(opA, opB, qa, qb, m + 2);
      kstep(opB, opA, qb, qa, m + 3);
    }
    lds_wave_sync();
#if HG_HIST_PROBE
    { HG_PROBE_T(pb2); p_proj += pb1 - pb0; p_loop += pb2 - pb1; }
#endif
  }
  HG_PROBE_T(pt2);

  // fixed-order sum of the 4 waves' tiles through LDS (deterministic)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ti = 0; ti < T; ++ti)
#pragma unroll
          for (int tj = 0; tj < T; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
              const int i = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * half;
              const int j = 32 * tj + l31;
              float *d = &red[(p * BLK + i) * BLK + j];
              if (w == 0) *d = acc[p][ti][tj][r]; else *d += acc[p][ti][tj][r];
            }
    }
    __syncthreads();
  }

  // slab write: slabs[((b*S+s)*Pn + po)*h*h + oi*h + oj], undoing the index flips of planes 1, 2
  const int h = P.h, Pn = P.P;
  float *slab = slabs + ((long long)(b * S + s) * Pn) * h * h;
  float tsum = 0.f;
  for (int e = threadIdx.x; e < 3 * BLK * BLK; e += 256) {
    const int p = e / (BLK * BLK), rem = e - p * BLK * BLK;
    const int il = rem / BLK, jl = rem - il * BLK;
    const int I = BLK * bi + il, J = BLK * bj + jl;
    if (I >= h || J >= h) continue;
    if (green && p != 1) continue;
    const int oi = (p == 0) ? I : h - 1 - I;
    const int oj = (p == 2) ? h - 1 - J : J;
    const int po = green ? 0 : p;
    const float v = red[e];
    slab[((long long)po * h + oi) * h + oj] = v;
    tsum += v;
  }
  // this workgroup's share of the image total (fixed summation order): k_hist_finish needs no pass to find the normaliser
  __syncthreads();
  tsum = hg_block_sum_256(tsum, reinterpret_cast<float *>(smem));
  if (threadIdx.x == 0) slab_tot[((long long)b * S + s) * gridDim.y + blockIdx.y] = (double)tsum;
#if HG_HIST_PROBE
  if (lane == 0) {
    HG_PROBE_T(pt3);
    atomicAdd(&hg_probe[0][0], 1ull); atomicAdd(&hg_probe[0][1], (unsigned long long)(pt3 - pt0));
    atomicAdd(&hg_probe[0][2], (unsigned long long)(pt1 - pt0)); atomicAdd(&hg_probe[0][3], (unsigned long long)p_proj);
    atomicAdd(&hg_probe[0][4], (unsigned long long)p_loop); atomicAdd(&hg_probe[0][5], (unsigned long long)(pt3 - pt2));
    atomicMax(&hg_probe[0][6], (unsigned long long)(pt3 - pt0));
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Backward.  With S' = sum(raw)+1e-6 and out = raw/S':  dL/draw = Ghat = (G - <G,out>) / S'.
// Every workgroup of k_hist_bwd rebuilds Ghat for its image in LDS (planes 1/2 index-flipped and
// zero-padded to BLK x BLK, the accumulation layout of the forward): 2 x 48 KB of L2 reads per
// workgroup instead of a separate low-occupancy prep launch.

// Chain rule from (dL/da, dL/db, dL/dc, dL/dIy) to the pixel and the store:
//   dL_R = da+db, dL_G = -da+dc, dL_B = -db-dc,  dx_c = dL_c/(x_c+1e-6) + dIy x_c/Iy,
// masked by the clamp of RGBuvHistBlock.py:76 when written straight to grad_x.
__device__ __forceinline__ void store_rgb_grad(const DevParams &P, const float *xb, int b, int n, float dr, float dg,
                                               float dbl, float *gdst) {
  if (P.mode == HG_RESIZE_NONE) {
    const int ys = n / P.Ws, xs = n - ys * P.Ws;
    const long long xo = ys * P.sh + xs * P.sw;
    const float xr = xb[xo], xg = xb[xo + P.sc], xbv = xb[xo + 2 * P.sc];
    float *gb = gdst + ((long long)b * P.C) * P.npix + n;
    gb[0] = grad_mask(P, xr) ? dr : 0.f;
    gb[P.npix] = grad_mask(P, xg) ? dg : 0.f;
    gb[2LL * P.npix] = grad_mask(P, xbv) ? dbl : 0.f;
  } else {
    // gradient w.r.t. the resized (already clamped) image: gxs[b][3][Hs*Ws]
    float *gb = gdst + ((long long)b * 3) * P.npix + n;
    gb[0] = dr; gb[P.npix] = dg; gb[2LL * P.npix] = dbl;
  }
}

__device__ __forceinline__ void store_pixel_grad(const DevParams &P, const float *xb, int b, int n, float r_,
                                                 float g_, float b_, float iy, float da, float db, float dc,
                                                 float dIy, float *gdst) {
  const float dLr = da + db, dLg = dc - da, dLb = -db - dc;
  float dr = dLr / (r_ + kEps), dg = dLg / (g_ + kEps), dbl = dLb / (b_ + kEps);
  if (P.intensity) {
    const float w = dIy / iy;
    dr = fmaf(w, r_, dr); dg = fmaf(w, g_, dg); dbl = fmaf(w, b_, dbl);
  }
  store_rgb_grad(P, xb, b, n, dr, dg, dbl, gdst);
}

// Chain rule of the one-plane projections: dU = dL/du, dV = dL/dv, dW = dL/dweight
__device__ __forceinline__ void store_pixel_grad_proj(const DevParams &P, const float *xb, int b, int n, float r_,
                                                      float g_, float b_, float iy, float dU, float dV, float dW,
                                                      float *gdst) {
  float dr, dg, dbl;
  if (P.proj == HG_PROJ_RGCHROMA) {
    // u = r/S, v = g/S, S = r+g+b+eps:  du/dr = 1/S - r/S^2, du/dg = du/db = -r/S^2 (same for v with g)
    const float S = ((r_ + g_) + b_) + kEps, inv = 1.f / S;
    const float common = -(dU * r_ + dV * g_) * inv * inv;
    dr = fmaf(dU, inv, common); dg = fmaf(dV, inv, common); dbl = common;
    if (P.intensity) {
      const float w = dW / iy;
      dr = fmaf(w, r_, dr); dg = fmaf(w, g_, dg); dbl = fmaf(w, b_, dbl);
    }
  } else {  // HG_PROJ_DIRECT: weight = channel 0, (u, v) = channels (1, 2)
    dr = P.intensity ? dW : 0.f; dg = dU; dbl = dV;
  }
  store_rgb_grad(P, xb, b, n, dr, dg, dbl, gdst);
}

// bin permutation shared by the MFMA K index and the accumulator row index (see k_hist_bwd)
__device__ __forceinline__ constexpr int beta0(int s) { return (s & 3) + 8 * ((s >> 2) & 3) + 32 * (s >> 4); }

// Main backward kernel (symmetric boundary, h <= 32*T).  grid = (S, B); 4 waves, each takes rounds
// of 32 pixels: lane l handles pixel (l&31) and the 16*T bins  beta0(s) + 4*(l>>5), s < 16*T.
//   Wa[i] = sum_j G0[i][j] kb_j + G1f[i][j] kc_j      (rows indexed by a-bins)
//   Wb[i] = sum_j G0[j][i] ka_j + G2f[i][j] kc_j      (rows indexed by b-bins)
//   Wc[i] = sum_j G1f[j][i] ka_j + G2f[j][i] kb_j     (rows indexed by c-bins)
// as D[bin][pixel] MFMA tiles (A = Ghat from LDS, B = kernel values generated in registers), then
//   dL/da = Iy * sum_i k'(a-b_i) Wa[i]   (same for b, c),   dL/dIy = 1/2 sum_i (ka Wa + kb Wb + kc Wc)[i]
//   dL_R = da+db, dL_G = -da+dc, dL_B = -db-dc,   dx_c = dL_c/(x_c+1e-6) + dIy x_c/Iy   
template <int T, int METHOD, bool GREEN, bool SHARE = false>
__global__ __launch_bounds__(256, HG_BWD_WAVES) void k_hist_bwd(const DevParams P, const float *__restrict__ x,
                                                                const float *__restrict__ gout,
                                                                const float *__restrict__ hist,
                                                                const float *__restrict__ sums,
                                                                float *__restrict__ gdst, const int rounds_per_wave) {
  constexpr int BLK = 32 * T, NS = 16 * T, LD = BLK + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *G = reinterpret_cast<float *>(smem);  // [3][BLK][LD]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, q = lane & 31;
  const int b = blockIdx.y, s_ = blockIdx.x;
  const float *xb = x + (long long)b * P.sb;
  constexpr bool green = GREEN;
  HG_PROBE_T(pt0);
#if HG_HIST_PROBE
  long long p_state = 0, p_loop = 0, p_epi = 0;
#endif

  {
    const int h = P.h, n = P.P * h * h;
    const float *g = gout + (long long)b * n, *o = hist + (long long)b * n;
    float d = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) d = fmaf(g[e], o[e], d);
    d = hg_block_sum_256(d, G);  // <G, out>; G's first 16 B are scratch until the fill below
    const float inv = 1.f / sums[b];
    for (int e = threadIdx.x; e < 3 * BLK * BLK; e += 256) {
      const int p = e / (BLK * BLK), rem = e - p * BLK * BLK;
      const int I = rem / BLK, J = rem - I * BLK;
      float v = 0.f;
      if (I < h && J < h && (!green || p == 1)) {
        const int oi = (p == 0) ? I : h - 1 - I;
        const int oj = (p == 2) ? h - 1 - J : J;
        const int po = green ? 0 : p;
        v = (g[((long long)po * h + oi) * h + oj] - d) * inv;
      }
      G[(p * BLK + I) * LD + J] = v;
    }
  }
  __syncthreads();
  const float *G0 = G, *G1 = G + BLK * LD, *G2 = G + 2 * BLK * LD;
  HG_PROBE_T(pt1);

  const long long wstart = ((long long)(s_ * 4 + wave) * rounds_per_wave) * 32;
  for (int rd = 0; rd < rounds_per_wave; ++rd) {
    const long long n0 = wstart + (long long)rd * 32;
    if (n0 >= P.npix) break;
    const int n = (int)n0 + q;
    const bool valid = n < P.npix;
    HG_PROBE_T(pr0);
    float r_, g_, b_, a, bb, c, iy;
    pixel_state(P, xb, b, n, valid, r_, g_, b_, a, bb, c, iy);

    // t_s = (u - lo - 4*half*step)/sigma - beta0(s)*step/sigma, double-single
    float th[3], tl[3];
    double ud[3] = {(double)a, (double)bb, (double)c};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const double tb = (ud[v] - P.lo - (double)(4 * half) * P.step) * P.inv_sigma_x;
      th[v] = (float)tb; tl[v] = (float)(tb - (double)th[v]);
    }
    auto eval = [&](int v, int s, float &t) -> float {
      const int beta = beta0(s) + 4 * half;
      if constexpr (METHOD == HG_METHOD_THRESHOLDING) {
        t = 0.f;
        return (fabs(ud[v] - bin_center(P, beta)) <= P.half_eps) ? 1.f : 0.f;
      } else {
        const float kf = -(float)beta0(s);
        t = fmaf(kf, P.ds_hi, th[v]) + fmaf(kf, P.ds_lo, tl[v]);
        if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) return __builtin_amdgcn_rcpf(fmaf(t, t, 1.f));
        else return expf(-(t * t));
      }
    };

    f32x16 W[3][T];
    // (no zero fill: the first K step's MFMAs take an inline-constant 0 as their C operand -- 48*T v_mov per round less)

    // Kernel values are generated just in time (3 evaluations per 6*T MFMAs) and NOT kept: the
    // epilogue re-evaluates them bit-identically, which keeps the kernel at W (48*T regs) + temporaries
    // so that several waves per SIMD can overlap one wave's VALU phases with another's MFMAs.
    // (Round 3 also ran this loop in groups of four steps -- within a group beta0 advances by one, so the twelve LDS operand
    // addresses and the bin offset become immediates: 24 -> 16.75 VALU instructions per 12 MFMAs in the ISA -- and measured
    // 0.9515 vs 0.955 ms at configs[1]: the K loop is not VALU-issue-bound.  The one-step form stays.)
    // MFMA loop: a REAL loop over the 16*T K-steps (2 bins each).  Fully unrolled, the compiler hoists
    // the (round-invariant) LDS operand reads and sinks the chain-free MFMA intrinsics below them,
    // which costs >200 spilled VGPRs; one 6*T-MFMA body (384*T cycles) per iteration needs no unroll.
    auto make_bops = [&](int s, BOps<T> &o) {
      const int b0 = (s & 3) + 8 * ((s >> 2) & 3) + 32 * (s >> 4);
      const int beta = b0 + 4 * half;
#pragma unroll
      for (int rt = 0; rt < T; ++rt) {
        const int row = 32 * rt + q;
        if (!green) {
          o.A[rt][0] = G0[row * LD + beta];
          o.A[rt][1] = G0[beta * LD + row];
          o.A[rt][2] = G2[beta * LD + row];
          o.A[rt][3] = G2[row * LD + beta];
        }
        o.A[rt][4] = G1[row * LD + beta];
        o.A[rt][5] = G1[beta * LD + row];
      }
      if constexpr (METHOD == HG_METHOD_THRESHOLDING) {
        const double bc = bin_center(P, beta);
        o.ka = (fabs(ud[0] - bc) <= P.half_eps) ? 1.f : 0.f;
        o.kb = (fabs(ud[1] - bc) <= P.half_eps) ? 1.f : 0.f;
        o.kc = (fabs(ud[2] - bc) <= P.half_eps) ? 1.f : 0.f;
      } else {
        const float kf = -(float)b0;
        // (a, b) as one packed-fp32 pair (v_pk_fma / v_pk_add: the same IEEE operations, half the instructions)
        const f32x2 kf2 = {kf, kf};
        const f32x2 tab = __builtin_elementwise_fma(kf2, f32x2{P.ds_hi, P.ds_hi}, f32x2{th[0], th[1]}) +
                          __builtin_elementwise_fma(kf2, f32x2{P.ds_lo, P.ds_lo}, f32x2{tl[0], tl[1]});
        const float ta = tab.x, tb = tab.y;
        const float tc = fmaf(kf, P.ds_hi, th[2]) + fmaf(kf, P.ds_lo, tl[2]);
        if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
          const f32x2 dab = __builtin_elementwise_fma(tab, tab, f32x2{1.f, 1.f});
          o.ka = __builtin_amdgcn_rcpf(dab.x);
          o.kb = __builtin_amdgcn_rcpf(dab.y);
          o.kc = __builtin_amdgcn_rcpf(fmaf(tc, tc, 1.f));
        } else {
          o.ka = expf(-(ta * ta)); o.kb = expf(-(tb * tb)); o.kc = expf(-(tc * tc));
        }
      }
    };
    HG_PROBE_T(pr1);
    if constexpr (SHARE) {
      // ---- round 4: two K steps per iteration, shared reciprocals, one block of packed instructions ----------------------
      // fp32 MFMA and VALU issue exclude each other on gfx950 and v_rcp_f32 runs at a quarter of the VALU rate
      // (profiles/r04_hist_pmc_stalls_raw.txt, r04_ubench_hist_fwd_loop.txt): per K step the 3 kernel evaluations were 13 VALU
      // instructions + 3 reciprocals (100 issue cycles next to 768 of MFMAs).  Two steps together are six denominators
      // (a1, b1, a2, b2, c1, c2): 1/(a1 a2 b1 b2) gives four reciprocals, 1/(c1 c2) two -- 21 packed VALU instructions + 2
      // v_rcp_f32 per TWO steps (58 cycles per step).  Written as inline assembly: clang 22 unpacks v_pk_* behind MFMAs.
      // Step s and s + 1 (s even) use the bins beta0(s) and beta0(s) + 1.  s_nop 0: trans -> VALU; s_nop 1: VALU -> MFMA.
      static_assert(METHOD == HG_METHOD_INVERSE_QUADRATIC && !GREEN, "shared-reciprocal backward");
      struct B2 { float A[2][T][6]; f32x2 k1, k2, kc; };
      const f32x2 thab = {th[0], th[1]}, tlab = {tl[0], tl[1]}, thc2 = {th[2], th[2]}, tlc2 = {tl[2], tl[2]};
      const f32x2 dsh2 = {P.ds_hi, P.ds_hi}, dsl2 = {P.ds_lo, P.ds_lo};
      auto lds2 = [&](int s, B2 &o) __attribute__((always_inline)) {
        const int b0 = (s & 3) + 8 * ((s >> 2) & 3) + 32 * (s >> 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int beta = b0 + j + 4 * half;
#pragma unroll
          for (int rt = 0; rt < T; ++rt) {
            const int row = 32 * rt + q;
            o.A[j][rt][0] = G0[row * LD + beta];
            o.A[j][rt][1] = G0[beta * LD + row];
            o.A[j][rt][2] = G2[beta * LD + row];
            o.A[j][rt][3] = G2[row * LD + beta];
            o.A[j][rt][4] = G1[row * LD + beta];
            o.A[j][rt][5] = G1[beta * LD + row];
          }
        }
      };
      auto gen2 = [&](int s, B2 &o) __attribute__((always_inline)) {
        const int b0 = (s & 3) + 8 * ((s >> 2) & 3) + 32 * (s >> 4);
        const float kf = -(float)b0;
        const f32x2 kfp = {kf, kf - 1.f};
        f32x2 k1, k2, kc;
        asm volatile(
            "v_pk_fma_f32 v[240:241], %3, %8, %4 op_sel_hi:[0,1,1]\n\t"                    // step 1 (a, b): kf ds_hi + t_hi
            "v_pk_fma_f32 v[242:243], %3, %8, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"     // step 2 (a, b)
            "v_pk_fma_f32 v[244:245], %3, %8, %6\n\t"                                       // (c step 1, c step 2)
            "v_pk_fma_f32 v[246:247], %3, %9, %5 op_sel_hi:[0,1,1]\n\t"                    // kf ds_lo + t_lo
            "v_pk_fma_f32 v[248:249], %3, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "v_pk_fma_f32 v[250:251], %3, %9, %7\n\t"
            "v_pk_add_f32 v[240:241], v[240:241], v[246:247]\n\t"                           // t
            "v_pk_add_f32 v[242:243], v[242:243], v[248:249]\n\t"
            "v_pk_add_f32 v[244:245], v[244:245], v[250:251]\n\t"
            "v_pk_fma_f32 v[240:241], v[240:241], v[240:241], 1.0 op_sel_hi:[1,1,0]\n\t"   // (a1, b1) = 1 + t^2
            "v_pk_fma_f32 v[242:243], v[242:243], v[242:243], 1.0 op_sel_hi:[1,1,0]\n\t"   // (a2, b2)
            "v_pk_fma_f32 v[244:245], v[244:245], v[244:245], 1.0 op_sel_hi:[1,1,0]\n\t"   // (c1, c2)
            "v_pk_mul_f32 v[246:247], v[240:241], v[242:243]\n\t"                           // (a1 a2, b1 b2)
            "v_mul_f32 v249, v244, v245\n\t"                                                // c1 c2
            "v_mul_f32 v248, v246, v247\n\t"                                                // a1 a2 b1 b2
            "v_rcp_f32 v249, v249\n\t"
            "v_rcp_f32 v248, v248\n\t"
            "s_nop 0\n\t"
            "v_pk_mul_f32 %2, v[244:245], v[248:249] op_sel:[1,1] op_sel_hi:[0,1]\n\t"      // (1/c1, 1/c2)
            "v_pk_mul_f32 v[246:247], v[246:247], v[248:249] op_sel:[1,0] op_sel_hi:[0,0]\n\t"  // (1/(a1 a2), 1/(b1 b2))
            "v_pk_mul_f32 %0, v[242:243], v[246:247]\n\t"                                   // (1/a1, 1/b1)
            "v_pk_mul_f32 %1, v[240:241], v[246:247]\n\t"                                   // (1/a2, 1/b2)
            "s_nop 1"
            : "=&v"(k1), "=&v"(k2), "=&v"(kc)
            : "v"(kfp), "v"(thab), "v"(tlab), "v"(thc2), "v"(tlc2), "s"(dsh2), "s"(dsl2)
            : "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251");
        o.k1 = k1; o.k2 = k2; o.kc = kc;
      };
      auto mfma12 = [&](const float (&A)[T][6], float ka, float kb, float kcv, bool first) __attribute__((always_inline)) {
        const f32x16 Z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < T; ++rt) {
          if (first) {                         // C = 0 as an inline constant: no zero fill of the 96 accumulators
            W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][0], kb, Z, 0, 0, 0);
            W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][1], ka, Z, 0, 0, 0);
            W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][2], kb, Z, 0, 0, 0);
          } else {
            W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][0], kb, W[0][rt], 0, 0, 0);
            W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][1], ka, W[1][rt], 0, 0, 0);
            W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][2], kb, W[2][rt], 0, 0, 0);
          }
          W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][3], kcv, W[1][rt], 0, 0, 0);
          W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][4], kcv, W[0][rt], 0, 0, 0);
          W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rt][5], ka, W[2][rt], 0, 0, 0);
        }
      };
      B2 cur2, nxt2;
      lds2(0, cur2);
      gen2(0, cur2);
      {                                        // steps 0 and 1, peeled (C = 0 in step 0)
        lds2(2, nxt2);
        mfma12(cur2.A[0], cur2.k1.x, cur2.k1.y, cur2.kc.x, true);
        mfma12(cur2.A[1], cur2.k2.x, cur2.k2.y, cur2.kc.y, false);
        __builtin_amdgcn_sched_barrier(0);
        gen2(2, nxt2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rt = 0; rt < T; ++rt)
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(nxt2.A[j][rt][i]));
        cur2 = nxt2;
      }
#pragma unroll 1
      for (int s = 2; s < NS; s += 2) {
        const int sn = min(s + 2, NS - 2);     // (the last iteration prepares operands nobody uses)
        lds2(sn, nxt2);
        mfma12(cur2.A[0], cur2.k1.x, cur2.k1.y, cur2.kc.x, false);
        mfma12(cur2.A[1], cur2.k2.x, cur2.k2.y, cur2.kc.y, false);
        // the LDS reads of the next pair of steps, the 12 * T MFMAs of this pair as one group, then the operand block
        __builtin_amdgcn_sched_barrier(0);
        gen2(sn, nxt2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rt = 0; rt < T; ++rt)
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(nxt2.A[j][rt][i]));
        cur2 = nxt2;
      }
    } else {
    BOps<T> cur;
    make_bops(0, cur);
    {                                        // K step 0, peeled: C = 0
      BOps<T> nxt;
      make_bops(1, nxt);
      const f32x16 Z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rt = 0; rt < T; ++rt) {
        if (!green) {
          W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][0], cur.kb, Z, 0, 0, 0);
          W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][1], cur.ka, Z, 0, 0, 0);
          W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][2], cur.kb, Z, 0, 0, 0);
          W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][3], cur.kc, W[1][rt], 0, 0, 0);
          W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][4], cur.kc, W[0][rt], 0, 0, 0);
          W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][5], cur.ka, W[2][rt], 0, 0, 0);
        } else {
          W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][4], cur.kc, Z, 0, 0, 0);
          W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][5], cur.ka, Z, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) W[1][rt][r] = 0.f;
        }
      }
#pragma unroll
      for (int rt = 0; rt < T; ++rt)
#pragma unroll
        for (int i = green ? 4 : 0; i < 6; ++i) asm volatile("" : "+v"(nxt.A[rt][i]));
      cur = nxt;
    }
#pragma unroll 1
    for (int s = 1; s < NS; ++s) {
      BOps<T> nxt;
      make_bops(min(s + 1, NS - 1), nxt);  // LDS reads + 3 kernel evaluations for the next K step
#pragma unroll
      for (int rt = 0; rt < T; ++rt) {
        if (!green) {
          W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][0], cur.kb, W[0][rt], 0, 0, 0);
          W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][1], cur.ka, W[1][rt], 0, 0, 0);
          W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][2], cur.kb, W[2][rt], 0, 0, 0);
          W[1][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][3], cur.kc, W[1][rt], 0, 0, 0);
        }
        W[0][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][4], cur.kc, W[0][rt], 0, 0, 0);
        W[2][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.A[rt][5], cur.ka, W[2][rt], 0, 0, 0);
      }
#if HG_BWD_SCHED_GROUPS
      // all LDS operand reads of the next step first, then the MFMAs in groups of HG_BWD_MFMA_GROUP with the next step's
      // address arithmetic / kernel evaluations between the groups (fp32 MFMA and VALU issue exclude each other on gfx950:
      // what the schedule controls is the number of MFMA <-> VALU switches)
      __builtin_amdgcn_sched_group_barrier(0x100, 6 * T, 0);
      {
        constexpr int NM = (green ? 2 : 6) * T, GRP = HG_BWD_MFMA_GROUP < NM ? HG_BWD_MFMA_GROUP : NM;
#pragma unroll
        for (int i = 0; i < NM; i += GRP) {
          __builtin_amdgcn_sched_group_barrier(0x008, GRP, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2 * GRP, 0);
        }
      }
#endif
      // pin: the next step's operands are complete here, after a step's worth of MFMA issue; without
      // it the compiler rotates the loop and every MFMA waits on an LDS read issued just before it
#pragma unroll
      for (int rt = 0; rt < T; ++rt)
#pragma unroll
        for (int i = green ? 4 : 0; i < 6; ++i) asm volatile("" : "+v"(nxt.A[rt][i]));
      cur = nxt;
    }
    }   // !SHARE

    HG_PROBE_T(pr2);
    // epilogue: this lane holds W[v][t][r] for bin beta0(16t+r)+4*half of pixel q.
    // The empty asm makes the eval inputs opaque so the compiler re-evaluates (4 VALU ops each)
    // instead of keeping 2 x 48*T values alive across the MFMA loop (CSE would cost ~190 VGPRs).
#pragma unroll
    for (int v = 0; v < 3; ++v) asm volatile("" : "+v"(th[v]), "+v"(tl[v]), "+v"(ud[v]));
    float gsum[3] = {0.f, 0.f, 0.f}, isum = 0.f;
    if constexpr (METHOD != HG_METHOD_THRESHOLDING) {
      // two consecutive bins per packed-fp32 operation: their accumulators are adjacent registers of one MFMA tile, the
      // bin offsets compile-time pairs -- 8 v_pk_* + 2 transcendentals per two bins instead of 18 instructions
      f32x2 gs2[3], is2 = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        gs2[v] = f32x2{0.f, 0.f};
        const f32x2 th2 = {th[v], th[v]}, tl2 = {tl[v], tl[v]};
        if constexpr (SHARE) {
          // four bins per reciprocal: (d0, d1), (d2, d3) -> r = 1 / (d0 d2 d1 d3) (the K loop's trick; no MFMA nearby, so
          // the compiler keeps the packed instructions)
#pragma unroll
          for (int s = 0; s < NS; s += 4) {
            f32x2 t2[2], d2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x2 kf2 = {-(float)beta0(s + 2 * j), -(float)beta0(s + 2 * j + 1)};
              t2[j] = __builtin_elementwise_fma(kf2, f32x2{P.ds_hi, P.ds_hi}, th2) +
                      __builtin_elementwise_fma(kf2, f32x2{P.ds_lo, P.ds_lo}, tl2);
              d2[j] = __builtin_elementwise_fma(t2[j], t2[j], f32x2{1.f, 1.f});
            }
            const f32x2 pq = d2[0] * d2[1];
            const float r4 = __builtin_amdgcn_rcpf(pq.x * pq.y);
            const f32x2 rq = f32x2{pq.y, pq.x} * f32x2{r4, r4};
            const f32x2 kq[2] = {d2[1] * rq, d2[0] * rq};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int sj = s + 2 * j;
              const f32x2 w2 = {W[v][sj >> 4][sj & 15], W[v][sj >> 4][(sj & 15) + 1]};
              const f32x2 kw2 = kq[j] * w2;
              is2 += kw2;
              gs2[v] = __builtin_elementwise_fma(t2[j] * kq[j], kw2, gs2[v]);
            }
#if HG_BWD_SCHED_BARRIER
            if ((s & 7) == 4) __builtin_amdgcn_sched_barrier(0);
#endif
          }
        } else
#pragma unroll
        for (int s = 0; s < NS; s += 2) {
          const f32x2 kf2 = {-(float)beta0(s), -(float)beta0(s + 1)};
          const f32x2 t2 = __builtin_elementwise_fma(kf2, f32x2{P.ds_hi, P.ds_hi}, th2) +
                           __builtin_elementwise_fma(kf2, f32x2{P.ds_lo, P.ds_lo}, tl2);
          f32x2 k2;
          if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
            const f32x2 d2 = __builtin_elementwise_fma(t2, t2, f32x2{1.f, 1.f});
            k2 = f32x2{__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
          } else {
            k2 = f32x2{expf(-(t2.x * t2.x)), expf(-(t2.y * t2.y))};
          }
          const f32x2 w2 = {W[v][s >> 4][s & 15], W[v][s >> 4][(s & 15) + 1]};
          const f32x2 kw2 = k2 * w2;
          is2 += kw2;
          if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) gs2[v] = __builtin_elementwise_fma(t2 * k2, kw2, gs2[v]);
          else gs2[v] = __builtin_elementwise_fma(t2, kw2, gs2[v]);
#if HG_BWD_SCHED_BARRIER
          if ((s & 7) == 6) __builtin_amdgcn_sched_barrier(0);  // keep the re-evaluations from being hoisted en bloc
#endif
        }
        gsum[v] = gs2[v].x + gs2[v].y;
      }
      isum = is2.x + is2.y;
    } else {
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          float t;
          const float kv = eval(v, s, t);
          isum += kv * W[v][s >> 4][s & 15];
        }
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) gsum[v] += __shfl_xor(gsum[v], 32, 64);
    isum += __shfl_xor(isum, 32, 64);

    const float da = iy * P.dk_scale * gsum[0], db = iy * P.dk_scale * gsum[1], dc = iy * P.dk_scale * gsum[2];
    const float dIy = P.intensity ? 0.5f * isum : 0.f;
    if (valid && half == 0) store_pixel_grad(P, xb, b, n, r_, g_, b_, iy, da, db, dc, dIy, gdst);
    if (valid && half == 1 && P.mode == HG_RESIZE_NONE) {
      for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + n] = 0.f;
    }
#if HG_HIST_PROBE
    { HG_PROBE_T(pr3); p_state += pr1 - pr0; p_loop += pr2 - pr1; p_epi += pr3 - pr2; }
#endif
  }
#if HG_HIST_PROBE
  if (lane == 0) {
    HG_PROBE_T(pt3);
    atomicAdd(&hg_probe[1][0], 1ull); atomicAdd(&hg_probe[1][1], (unsigned long long)(pt3 - pt0));
    atomicAdd(&hg_probe[1][2], (unsigned long long)(pt1 - pt0)); atomicAdd(&hg_probe[1][3], (unsigned long long)p_state);
    atomicAdd(&hg_probe[1][4], (unsigned long long)p_loop); atomicAdd(&hg_probe[1][5], (unsigned long long)p_epi);
    atomicMax(&hg_probe[1][6], (unsigned long long)(pt3 - pt0));
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Plane-at-a-time MFMA backward: any hist_boundary, h <= 32*RT <= 128, the one-plane projections (rg-chroma, Lab)
// and green_only.  The mirrored-table merge of k_hist_bwd needs lo == -hi and all of Ghat (3 x h x (h+1) floats) in
// LDS; here ONE plane's Ghat is resident (h = 128: 66 KB, two workgroups per CU) and the workgroup walks its pixels
// once per plane, the reference's plane structure taken literally (RGBuvHistBlock.py:112-148, 150-187, 190-222):
//   Wu[i] = sum_j Ghat_p[i][j] k(v-b_j)      Wv[j] = sum_i Ghat_p[i][j] k(u-b_i)          (u, v) of plane p
//   dL/du = Iy sum_i k'(u-b_i) Wu[i]          dL/dv = Iy sum_j k'(v-b_j) Wv[j]             dL/dIy += sum_i k(u-b_i) Wu[i]
// as D[bin][pixel] tiles exactly like k_hist_bwd (same K-index permutation beta0, same double-single t, kernel
// values re-evaluated in the epilogue): 2*RT MFMAs per K step against 2 kernel evaluations.  (da, db, dc, dIy) of a
// pixel are carried from plane to plane through `part` ([B][4][npix] floats, written and re-read by the same lane);
// the last plane applies the chain rule and stores.  Replaces the one-pixel-per-lane fp64 k_hist_bwd_generic for every
// smooth-kernel case up to h = 128 (configs[4]: h = 128).
template <int RT>
struct POps { float Au[RT], Av[RT]; float ku, kv; };

template <int RT, int METHOD>
__global__ __launch_bounds__(256, 2) void k_hist_bwd_planes(const DevParams P, const float *__restrict__ x,
                                                            const float *__restrict__ gout,
                                                            const float *__restrict__ hist,
                                                            const float *__restrict__ sums, float *__restrict__ part,
                                                            float *__restrict__ gdst, const int rounds_per_wave) {
  constexpr int BLK = 32 * RT, NS = 16 * RT, LD = BLK + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *G = reinterpret_cast<float *>(smem);  // [BLK][LD]: Ghat of the current plane, zero-padded

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, q = lane & 31;
  const int b = blockIdx.y, s_ = blockIdx.x;
  const float *xb = x + (long long)b * P.sb;
  const int h = P.h, nplanes = P.P;
  const float *g = gout + (long long)b * nplanes * h * h, *o = hist + (long long)b * nplanes * h * h;
  float dot = 0.f;
  for (int e = threadIdx.x; e < nplanes * h * h; e += 256) dot = fmaf(g[e], o[e], dot);
  dot = hg_block_sum_256(dot, G);  // <G, out>
  const float inv = 1.f / sums[b];
  float *pb = part + (long long)b * 4 * P.npix;
  const long long wstart = ((long long)(s_ * 4 + wave) * rounds_per_wave) * 32;

  for (int pi = 0; pi < nplanes; ++pi) {
    const int p = P.green ? 1 : pi;                   // geometric plane: (u, v) = (a, b) | (-a, c) | (-b, -c)
    __syncthreads();                                  // the previous plane's operand reads are done
    for (int e = threadIdx.x; e < BLK * BLK; e += 256) {
      const int I = e / BLK, J = e - I * BLK;
      G[I * LD + J] = (I < h && J < h) ? (g[((long long)pi * h + I) * h + J] - dot) * inv : 0.f;
    }
    __syncthreads();

    for (int rd = 0; rd < rounds_per_wave; ++rd) {
      const long long n0 = wstart + (long long)rd * 32;
      if (n0 >= P.npix) break;
      const int n = (int)n0 + q;
      const bool valid = n < P.npix;
      float r_, g_, b_, a, bb, c, iy;
      pixel_state(P, xb, b, n, valid, r_, g_, b_, a, bb, c, iy);
      const float u = (p == 0) ? a : (p == 1 ? -a : -bb), v = (p == 0) ? bb : (p == 1 ? c : -c);

      // t_s = (u - lo - 4*half*step)/sigma - beta0(s)*step/sigma, double-single (beta0(s) < 128: beta0*ds_hi exact)
      float th[2], tl[2];
      {
        const double tu = ((double)u - P.lo - (double)(4 * half) * P.step) * P.inv_sigma_x;
        const double tv = ((double)v - P.lo - (double)(4 * half) * P.step) * P.inv_sigma_x;
        th[0] = (float)tu; tl[0] = (float)(tu - (double)th[0]);
        th[1] = (float)tv; tl[1] = (float)(tv - (double)th[1]);
      }
      auto kern = [&](float t) -> float {
        if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) return __builtin_amdgcn_rcpf(fmaf(t, t, 1.f));
        else return expf(-(t * t));
      };

      f32x16 Wu[RT], Wv[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Wu[t][r] = 0.f; Wv[t][r] = 0.f; }

      auto make_ops = [&](int s, POps<RT> &op) {
        const int b0 = beta0(s);
        const int beta = b0 + 4 * half;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = 32 * rt + q;
          op.Au[rt] = G[row * LD + beta];
          op.Av[rt] = G[beta * LD + row];
        }
        const float kf = -(float)b0;
        const f32x2 kf2 = {kf, kf};                      // (u, v) as one packed-fp32 pair
        const f32x2 tuv = __builtin_elementwise_fma(kf2, f32x2{P.ds_hi, P.ds_hi}, f32x2{th[0], th[1]}) +
                          __builtin_elementwise_fma(kf2, f32x2{P.ds_lo, P.ds_lo}, f32x2{tl[0], tl[1]});
        if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
          const f32x2 d2 = __builtin_elementwise_fma(tuv, tuv, f32x2{1.f, 1.f});
          op.ku = __builtin_amdgcn_rcpf(d2.x); op.kv = __builtin_amdgcn_rcpf(d2.y);
        } else {
          op.ku = kern(tuv.x); op.kv = kern(tuv.y);
        }
      };
      POps<RT> cur;
      make_ops(0, cur);
#pragma unroll 1
      for (int s = 0; s < NS; ++s) {
        POps<RT> nxt;
        make_ops(min(s + 1, NS - 1), nxt);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          Wu[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.Au[rt], cur.kv, Wu[rt], 0, 0, 0);
          Wv[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.Av[rt], cur.ku, Wv[rt], 0, 0, 0);
        }
#if HG_BWD_SCHED_GROUPS
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RT, 0);
#pragma unroll
        for (int i = 0; i < 2 * RT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#endif
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(nxt.Au[rt]), "+v"(nxt.Av[rt]));
        cur = nxt;
      }

      // epilogue: this lane holds W*[t][r] for bin beta0(16t+r)+4*half of pixel q; kernel values re-evaluated
      asm volatile("" : "+v"(th[0]), "+v"(tl[0]), "+v"(th[1]), "+v"(tl[1]));
      float gu, gv, isum;
      {
        // two consecutive bins per packed-fp32 operation (adjacent accumulator registers), as in k_hist_bwd
        f32x2 gu2 = {0.f, 0.f}, gv2 = {0.f, 0.f}, is2 = {0.f, 0.f};
        auto kern2 = [&](const f32x2 &t) __attribute__((always_inline)) -> f32x2 {
          if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
            const f32x2 d = __builtin_elementwise_fma(t, t, f32x2{1.f, 1.f});
            return f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
          } else {
            return f32x2{expf(-(t.x * t.x)), expf(-(t.y * t.y))};
          }
        };
#pragma unroll
        for (int s = 0; s < NS; s += 2) {
          const f32x2 kf2 = {-(float)beta0(s), -(float)beta0(s + 1)};
          const f32x2 dsh = {P.ds_hi, P.ds_hi}, dsl = {P.ds_lo, P.ds_lo};
          const f32x2 tu = __builtin_elementwise_fma(kf2, dsh, f32x2{th[0], th[0]}) + __builtin_elementwise_fma(kf2, dsl, f32x2{tl[0], tl[0]});
          const f32x2 tv = __builtin_elementwise_fma(kf2, dsh, f32x2{th[1], th[1]}) + __builtin_elementwise_fma(kf2, dsl, f32x2{tl[1], tl[1]});
          const f32x2 ku = kern2(tu), kv = kern2(tv);
          const f32x2 kwu = ku * f32x2{Wu[s >> 4][s & 15], Wu[s >> 4][(s & 15) + 1]};
          const f32x2 kwv = kv * f32x2{Wv[s >> 4][s & 15], Wv[s >> 4][(s & 15) + 1]};
          is2 += kwu;
          if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
            gu2 = __builtin_elementwise_fma(tu * ku, kwu, gu2); gv2 = __builtin_elementwise_fma(tv * kv, kwv, gv2);
          } else {
            gu2 = __builtin_elementwise_fma(tu, kwu, gu2); gv2 = __builtin_elementwise_fma(tv, kwv, gv2);
          }
#if HG_BWD_SCHED_BARRIER
          if ((s & 7) == 6) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        gu = gu2.x + gu2.y; gv = gv2.x + gv2.y; isum = is2.x + is2.y;
      }
      gu += __shfl_xor(gu, 32, 64);
      gv += __shfl_xor(gv, 32, 64);
      isum += __shfl_xor(isum, 32, 64);
      const float du = iy * P.dk_scale * gu, dv = iy * P.dk_scale * gv;   // dL/du, dL/dv of this plane

      if (valid && half == 0) {
        float da = 0.f, db = 0.f, dc = 0.f, dIy = 0.f;
        if (pi > 0) { da = pb[n]; db = pb[P.npix + n]; dc = pb[2LL * P.npix + n]; dIy = pb[3LL * P.npix + n]; }
        if (p == 0) { da += du; db += dv; }
        else if (p == 1) { da -= du; dc += dv; }
        else { db -= du; dc -= dv; }
        dIy += isum;
        if (pi + 1 < nplanes) {
          pb[n] = da; pb[P.npix + n] = db; pb[2LL * P.npix + n] = dc; pb[3LL * P.npix + n] = dIy;
        } else if (P.proj != HG_PROJ_RGBUV) {
          // one plane binned as (u, v) = (-a, c): dL/du = -da, dL/dv = dc
          store_pixel_grad_proj(P, xb, b, n, r_, g_, b_, iy, -da, dc, P.intensity ? dIy : 0.f, gdst);
        } else {
          store_pixel_grad(P, xb, b, n, r_, g_, b_, iy, da, db, dc, P.intensity ? dIy : 0.f, gdst);
        }
      }
      if (valid && half == 1 && pi + 1 == nplanes && P.mode == HG_RESIZE_NONE) {
        for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + n] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Generic backward (any h, any hist_boundary): the reference's plane structure taken literally,
// one pixel per lane, fp64 kernel evaluation, fp32 mat-vecs against Ghat held in global/L2 (every
// lane reads the same Ghat element -> one broadcast load).  Used where the MFMA kernel does not
// apply (h > 64 or an asymmetric boundary); ~10x slower per pixel but exact in structure.
__global__ __launch_bounds__(1024) void k_hist_ghat(const float *__restrict__ gout, const float *__restrict__ hist,
                                                    const float *__restrict__ sums, float *__restrict__ gh, int n) {
  __shared__ float sm16[16];
  const int b = blockIdx.x;
  const float *g = gout + (long long)b * n, *o = hist + (long long)b * n;
  float d = 0.f;
  for (int e = threadIdx.x; e < n; e += 1024) d = fmaf(g[e], o[e], d);
  d = hg_wave_sum(d);
  if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = d;
  __syncthreads();
  d = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) d += sm16[w];      // fixed order: deterministic
  const float inv = 1.f / sums[b];
  for (int e = threadIdx.x; e < n; e += 1024) gh[(long long)b * n + e] = (g[e] - d) * inv;
}

template <int METHOD>
__device__ __forceinline__ void kern_eval_d(const DevParams &P, float u, int i, float &k, float &dk) {
  const double d = (double)u - bin_center(P, i);
  if constexpr (METHOD == HG_METHOD_THRESHOLDING) {
    k = (fabs(d) <= P.half_eps) ? 1.f : 0.f; dk = 0.f;
  } else {
    const double t = d * P.inv_sigma_x;
    if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
      const double kd = 1.0 / (1.0 + t * t);
      k = (float)kd; dk = (float)(-2.0 * t * kd * kd * P.inv_sigma_x);
    } else {
      const double kd = exp(-t * t);
      k = (float)kd; dk = (float)(-2.0 * t * kd * P.inv_sigma_x);
    }
  }
}

template <int METHOD>
__global__ __launch_bounds__(64) void k_hist_bwd_generic(const DevParams P, const float *__restrict__ x,
                                                         const float *__restrict__ gh, float *__restrict__ gdst) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *kbuf = reinterpret_cast<float *>(smem);  // [h][64]
  const int lane = threadIdx.x, b = blockIdx.y, h = P.h;
  const int n = blockIdx.x * 64 + lane;
  const bool valid = n < P.npix;
  const float *xb = x + (long long)b * P.sb;
  float r_ = 0.f, g_ = 0.f, b_ = 0.f;
  if (valid) sample_rgb(P, xb, n, r_, g_, b_);
  float a, bb, c, iy;
  project(P, r_, g_, b_, a, bb, c, iy);
  // plane p: (u, v) = (su*U, sv*V) with U,V in {a,b,c}  (RGBuvHistBlock.py:112-115,150-153,190-193)
  const float uvals[3] = {a, -a, -bb}, vvals[3] = {bb, c, -c};
  float gu[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f}, isum = 0.f;
  for (int p = 0; p < 3; ++p) {
    if (P.green && p != 1) continue;
    const float *G = gh + ((long long)b * P.P + (P.green ? 0 : p)) * h * h;
    const float u = uvals[p], v = vvals[p];
    // pass 1: T_i = sum_j G[i][j] kv_j ;  g_u += k'u_i T_i ;  dIy += ku_i T_i
    for (int j = 0; j < h; ++j) { float k, dk; kern_eval_d<METHOD>(P, v, j, k, dk); kbuf[j * 64 + lane] = k; }
    for (int i = 0; i < h; ++i) {
      float T = 0.f;
      const float *Gi = G + (long long)i * h;
      for (int j = 0; j < h; ++j) T = fmaf(Gi[j], kbuf[j * 64 + lane], T);
      float k, dk; kern_eval_d<METHOD>(P, u, i, k, dk);
      gu[p] = fmaf(dk, T, gu[p]);
      isum = fmaf(k, T, isum);
    }
    // pass 2: Sx_j = sum_i G[i][j] ku_i ;  g_v += k'v_j Sx_j
    for (int i = 0; i < h; ++i) { float k, dk; kern_eval_d<METHOD>(P, u, i, k, dk); kbuf[i * 64 + lane] = k; }
    for (int j = 0; j < h; ++j) {
      float Sx = 0.f;
      for (int i = 0; i < h; ++i) Sx = fmaf(G[(long long)i * h + j], kbuf[i * 64 + lane], Sx);
      float k, dk; kern_eval_d<METHOD>(P, v, j, k, dk);
      gv[p] = fmaf(dk, Sx, gv[p]);
    }
  }
  // u0 = a, v0 = b, u1 = -a, v1 = c, u2 = -b, v2 = -c
  const float da = iy * (gu[0] - gu[1]), db = iy * (gv[0] - gu[2]), dc = iy * (gv[1] - gv[2]);
  const float dIy = P.intensity ? isum : 0.f;
  if (valid) {
    if (P.proj != HG_PROJ_RGBUV)   // one plane, binned as (u, v) = (-a, c): gu[1] = dL/du, gv[1] = dL/dv (before the weight)
      store_pixel_grad_proj(P, xb, b, n, r_, g_, b_, iy, iy * gu[1], iy * gv[1], dIy, gdst);
    else
      store_pixel_grad(P, xb, b, n, r_, g_, b_, iy, da, db, dc, dIy, gdst);
    if (P.mode == HG_RESIZE_NONE)
      for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + n] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// Thresholding (RGBuvHistBlock.py:116-124): the kernel is a 0/1 window of width eps around each bin centre, so a
// pixel touches (normally) ONE bin per plane -- a true scatter-add histogram, HBM-bound, not the dense h x h
// product of the smooth kernels.  One workgroup takes a pixel range of one image and, plane after plane, bins it
// into an h x h LDS grid with 64-bit FIXED-POINT atomics (Iy * 2^32): integer sums are order-independent, so the
// result is deterministic (and more accurate than an fp32 running sum); the grid is flushed as a float slab in the
// layout of k_hist_fwd (k_hist_finish sums and normalises).  Pixels are re-projected per plane
// (their second and third read hit L2).
constexpr double kThrScale = 4294967296.0;   // 2^32

// candidate bin range [lo_i, hi_i] for |u - b_i| <= eps/2 (one spare bin either side; hits are decided by thr_hit);
// inv_step = 1/step, w = half_eps/step (0 when step == 0: single bin)
__device__ __forceinline__ void thr_range(const DevParams &P, float u, double inv_step, double w, int &lo_i, int &hi_i) {
  if (P.h == 1 || !(P.step > 0.0)) { lo_i = 0; hi_i = P.h - 1; return; }
  const double t = ((double)u - P.lo) * inv_step;
  const double a = floor(t - w) - 1.0, b = ceil(t + w) + 1.0;
  lo_i = a < 0.0 ? 0 : (a > (double)(P.h - 1) ? P.h : (int)a);
  hi_i = b > (double)(P.h - 1) ? P.h - 1 : (b < 0.0 ? -1 : (int)b);
}

__device__ __forceinline__ bool thr_hit(const DevParams &P, float u, int i) {
  return fabs((double)u - bin_center(P, i)) <= P.half_eps;    // the reference's fp64 comparison, bit for bit
}

// single: the windows are narrower than the bin spacing (every symmetric boundary: eps = (hi-lo)/h < (hi-lo)/(h-1)), so
// only the NEAREST bin can contain u -- and near a midpoint, where rounding could pick the other neighbour, neither does.
__device__ __forceinline__ int thr_single(const DevParams &P, float u, double inv_step) {
  const double t = ((double)u - P.lo) * inv_step;
  if (!(t > -1.0 && t < (double)P.h)) return -1;
  int i = (int)rint(t);
  i = i < 0 ? 0 : (i > P.h - 1 ? P.h - 1 : i);
  return thr_hit(P, u, i) ? i : -1;
}

__device__ __forceinline__ void thr_scatter_plane(const DevParams &P, unsigned long long *bins, float u, float v,
                                                  double inv_step, double w, bool single, unsigned long long q) {
  if (single) {
    const int i = thr_single(P, u, inv_step), j = thr_single(P, v, inv_step);
    if (i >= 0 && j >= 0) atomicAdd(&bins[i * P.h + j], q);
    return;
  }
  int ul, uh, vl, vh;
  thr_range(P, u, inv_step, w, ul, uh);
  thr_range(P, v, inv_step, w, vl, vh);
  for (int i = ul; i <= uh; ++i) {
    if (!thr_hit(P, u, i)) continue;
    for (int j = vl; j <= vh; ++j)
      if (thr_hit(P, v, j)) atomicAdd(&bins[i * P.h + j], q);
  }
}

// exact (integer) total of a workgroup's fixed-point grid; NT threads, result in every thread
template <int NT>
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sm /* NT/64 */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long t = 0ull;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += sm[w];
  __syncthreads();
  return t;
}

// Slab sum + normalisation (stage 4, RGBuvHistBlock.py:224-228: hist / (sum + 1e-6)) in ONE launch.  The producing
// kernels leave their share of the image total in slab_tot (scatter paths: the exact integer sum of each fixed-point
// grid; k_hist_fwd: a fixed-order block sum), so the normaliser needs no pass of its own over the histogram.
__global__ __launch_bounds__(256) void k_hist_finish(const float *__restrict__ slabs, const double *__restrict__ slab_tot,
                                                     float *__restrict__ hist, float *__restrict__ sum_out, int S,
                                                     int ntot, int n_per_img) {
  __shared__ double st[256];
  const int b = blockIdx.y;
  // the ntot partial totals of the image (one per slab, times the bin blocks of the dense path) are fetched by ntot
  // threads at once (a serial loop would pay ntot memory latencies) and summed in order by everyone
  const int Sl = ntot < 256 ? ntot : 256;
  if ((int)threadIdx.x < Sl) st[threadIdx.x] = slab_tot[(long long)b * ntot + threadIdx.x];
  __syncthreads();
  double tot = 0.0;
  for (int s = 0; s < Sl; ++s) tot += st[s];
  for (int s = 256; s < ntot; ++s) tot += slab_tot[(long long)b * ntot + s];   // more totals than threads: never at these sizes
  const float den = (float)tot + kEps;
  if (blockIdx.x == 0 && threadIdx.x == 0) sum_out[b] = den;
  const float *src = slabs + (long long)b * S * n_per_img;
  const int e0 = blockIdx.x * 1024 + threadIdx.x;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; s += 4) {                                 // 16 independent loads in flight, summed in slab order
    float t[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        t[u][k] = (s + u < S && e < n_per_img) ? src[(long long)(s + u) * n_per_img + e] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += t[u][k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = e0 + k * 256;
    if (e < n_per_img) hist[(long long)b * n_per_img + e] = v[k] / den;
  }
}

// ---- fast window classification ------------------------------------------------------------------------------
// The scatter kernels were bound by the three fp64 logarithms per pixel (~35 us of fp64 VALU work at configs[1]
// against 4 us of HBM time).  Those logs exist to make u bit-identical to the reference so that the 0/1 window
// decision |u - b_i| <= eps/2 is the reference's; but the decision only depends on the last bits of u when u sits
// within rounding distance of a window edge.  So: classify with v_log_f32 (|error| <= kFastLogRel * |ln x|, checked
// exhaustively by hg_selftest_fastlog) and fp32 bin arithmetic carrying an explicit error bound `m`; a value
// farther than m from both window edges has the same decision as the exact evaluation.  Pixels with any of their six
// values inside a margin (about 1 in 3 000) take the exact fp64 path.  Identical histograms, bit for bit
// (tests: fast vs HG_THR_EXACT=1).
constexpr float kFastLogRel = 3.6e-7f;   // margin per unit of |ln x1| + |ln x2|: log error (measured max <= 2.4e-7) + rounding of the difference (6e-8)

// ln x for normal x > 0 on the transcendental unit: v_log_f32 (log2, ~1 ulp) times ln 2 -- two instructions.  (__logf
// would add denormal scaling and an extended-precision multiply: 12 instructions for the same 2-ulp result here.)
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// Classification runs in bin units: t = (v - lo)/step as ONE fma, nearest bin rint(t), distance |t - rint(t)| (exact)
// against the window half-width eps/(2 step).  mt0 bounds the roundings of that fma (constants and result, all
// proportional to the size of t <= h) in bin units.
struct ThrFast {
  float inv_step, c0;      // t = fma(v, inv_step, c0), c0 = -lo/step
  float thr_t;             // eps / (2 step)
  float mt0, mrel;         // margin (bin units) = mt0 + mrel * (|ln x1| + |ln x2|)
  unsigned hmax;           // h - 1
};

__device__ __forceinline__ ThrFast make_thr_fast(const DevParams &P) {
  ThrFast F;
  const double is = P.step > 0.0 ? 1.0 / P.step : 0.0;
  F.inv_step = (float)is;
  F.c0 = (float)(-P.lo * is);
  F.thr_t = (float)(P.half_eps * is);
  F.mt0 = 2.5e-7f * (float)P.h + 1.0e-7f * F.inv_step;
  F.mrel = kFastLogRel * F.inv_step;
  F.hmax = (unsigned)(P.h - 1);
  return F;
}

// one value: bin index or -1; unsafe |= decision within the margin
__device__ __forceinline__ int thr_fast_one(const ThrFast &F, float v, float m, bool &unsafe) {
  const float t = fmaf(v, F.inv_step, F.c0);
  const float fi = rintf(t);
  const float dt = fabsf(t - fi);
  unsafe |= fabsf(dt - F.thr_t) <= m;
  const int i = (int)fi;                                   // saturating conversion; out-of-range fails the unsigned test
  return (dt < F.thr_t && (unsigned)i <= F.hmax) ? i : -1;
}

// Iy -> 2^32 fixed point, round half up: exactly (unsigned long long)((double)iy * 2^32 + 0.5), from the float's bits
__device__ __forceinline__ unsigned long long iy_fixed(float iy) {
  const unsigned bits = __float_as_uint(iy);
  const int sh = (int)(bits >> 23) - 118;                  // iy = m * 2^(e-150), times 2^32
  const unsigned m = (bits & 0x7FFFFFu) | 0x800000u;
  if (sh >= 0) return (unsigned long long)m << sh;         // iy >= 2^-9 (and < 2^33): an integer already
  const int s = -sh;
  if (s >= 25) return 0ull;                                // iy < 2^-34: rounds to 0
  return (unsigned long long)((m >> s) + ((m >> (s - 1)) & 1u));
}

// RGB-uv, `single` windows: the six bin indices (plane p: idx[2p] = u bin, idx[2p+1] = v bin; -1 = outside every
// window) and the weight.  Returns false when the exact path has to decide.  SYM (lo == -hi): b_(h-1-i) = -b_i, so the
// window of -v is the mirror image of the window of v -- three classifications instead of six (the fp64 bin table is
// symmetric to 1e-16, far inside the margin).
template <bool SYM>
__device__ __forceinline__ bool thr_fast_pixel(const DevParams &P, const ThrFast &F, float r, float g, float b,
                                               int idx[6], float &iy) {
  const float lr = fast_ln(__fadd_rn(r, kEps)), lg = fast_ln(__fadd_rn(g, kEps)), lb = fast_ln(__fadd_rn(b, kEps));
  const float a = lr - lg, bb = lr - lb, c = lg - lb;
  const float ar = fabsf(lr), ag = fabsf(lg), ab = fabsf(lb);
  const float ma = fmaf(F.mrel, ar + ag, F.mt0), mb = fmaf(F.mrel, ar + ab, F.mt0), mc = fmaf(F.mrel, ag + ab, F.mt0);
  bool unsafe = false;
  idx[0] = thr_fast_one(F, a, ma, unsafe);
  idx[1] = thr_fast_one(F, bb, mb, unsafe);
  idx[3] = thr_fast_one(F, c, mc, unsafe);
  if constexpr (SYM) {
    const int hm = P.h - 1;
    idx[2] = idx[0] >= 0 ? hm - idx[0] : -1;
    idx[4] = idx[1] >= 0 ? hm - idx[1] : -1;
    idx[5] = idx[3] >= 0 ? hm - idx[3] : -1;
  } else {
    idx[2] = thr_fast_one(F, -a, ma, unsafe);
    idx[4] = thr_fast_one(F, -bb, mb, unsafe);
    idx[5] = thr_fast_one(F, -c, mc, unsafe);
  }
  iy = P.intensity ? __fsqrt_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(g, g)), __fmul_rn(b, b)), kEps)) : 1.f;
  return !unsafe;
}

// the same six indices by the reference's arithmetic (fp64 logs rounded to fp32, fp64 window comparison)
__device__ __forceinline__ void thr_exact_pixel(const DevParams &P, float r, float g, float b, double inv_step,
                                                int idx[6], float &iy) {
  float a, bb, c;
  project(P, r, g, b, a, bb, c, iy);
  idx[0] = thr_single(P, a, inv_step);   idx[1] = thr_single(P, bb, inv_step);
  idx[2] = thr_single(P, -a, inv_step);  idx[3] = thr_single(P, c, inv_step);
  idx[4] = thr_single(P, -bb, inv_step); idx[5] = thr_single(P, -c, inv_step);
}

// exhaustive check of the fast logarithm's error bound over every float in [1e-6, 1 + 2e-6] (the range of x + 1e-6
// for clamped pixels): out[0] = max |__logf - ln| / |ln| over |ln x| >= 1e-3, out[1] = max absolute error elsewhere
__global__ __launch_bounds__(256) void k_selftest_fastlog(uint32_t first, uint32_t count, unsigned int *out) {
  float mrel = 0.f, mabs = 0.f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
    union { uint32_t u; float f; } cv; cv.u = first + i;
    const float ex = (float)log((double)cv.f), fa = fast_ln(cv.f);
    const float e = fabsf(fa - ex);
    if (fabsf(ex) >= 1e-3f) mrel = fmaxf(mrel, e / fabsf(ex)); else mabs = fmaxf(mabs, e);
  }
  atomicMax(&out[0], __float_as_uint(mrel));      // non-negative floats order like their bit patterns
  atomicMax(&out[1], __float_as_uint(mabs));
}

// ALL3: the grids of all planes are in LDS at once (3 h^2 x 8 B <= 150 KB, h <= 79): every pixel is read and projected
// (3 fp64 logs) ONCE; otherwise plane after plane through one grid (the 2nd / 3rd read of a pixel hits L2).
template <bool ALL3>
__global__ __launch_bounds__(ALL3 ? 1024 : 256) void k_hist_thr_fwd(const DevParams P, const float *__restrict__ x,
                                                                    float *__restrict__ slabs,
                                                                    double *__restrict__ slab_tot, int per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned long long sm_tot[16];
  unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem);   // [planes][h][h]
  constexpr int NT = ALL3 ? 1024 : 256;
  unsigned long long tot = 0ull;
  const int b = blockIdx.y, s = blockIdx.x, S = gridDim.x, h = P.h, hh = h * h;
  const float *xb = x + (long long)b * P.sb;
  const int n0 = s * per_block, n1 = min(P.npix, n0 + per_block);
  float *slab = slabs + ((long long)(b * S + s) * P.P) * hh;
  const double inv_step = P.step > 0.0 ? 1.0 / P.step : 0.0, w = P.half_eps * inv_step;
  const bool single = P.h > 1 && P.step > 2.0 * P.half_eps * (1.0 + 1e-9);
  if constexpr (ALL3) {
    for (int e = threadIdx.x; e < P.P * hh; e += NT) bins[e] = 0ull;
    __syncthreads();
    for (int n = n0 + threadIdx.x; n < n1; n += NT) {
      float r, g, bl, a, bb, c, iy;
      sample_rgb(P, xb, n, r, g, bl);
      project(P, r, g, bl, a, bb, c, iy);
      const unsigned long long q = (unsigned long long)((double)iy * kThrScale + 0.5);
      if (P.green) {
        thr_scatter_plane(P, bins, -a, c, inv_step, w, single, q);
      } else {
        thr_scatter_plane(P, bins, a, bb, inv_step, w, single, q);
        thr_scatter_plane(P, bins + hh, -a, c, inv_step, w, single, q);
        thr_scatter_plane(P, bins + 2 * hh, -bb, -c, inv_step, w, single, q);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P.P * hh; e += NT) {
      const unsigned long long v = bins[e];
      tot += v;
      slab[e] = (float)((double)v * (1.0 / kThrScale));
    }
  } else {
    for (int p = 0; p < 3; ++p) {
      if (P.green && p != 1) continue;
      for (int e = threadIdx.x; e < hh; e += NT) bins[e] = 0ull;
      __syncthreads();
      for (int n = n0 + threadIdx.x; n < n1; n += NT) {
        float r, g, bl, a, bb, c, iy;
        sample_rgb(P, xb, n, r, g, bl);
        project(P, r, g, bl, a, bb, c, iy);
        const float u = p == 0 ? a : (p == 1 ? -a : -bb), v = p == 0 ? bb : (p == 1 ? c : -c);
        thr_scatter_plane(P, bins, u, v, inv_step, w, single, (unsigned long long)((double)iy * kThrScale + 0.5));
      }
      __syncthreads();
      float *dst = slab + (long long)(P.green ? 0 : p) * hh;
      for (int e = threadIdx.x; e < hh; e += NT) {
        const unsigned long long v = bins[e];
        tot += v;
        dst[e] = (float)((double)v * (1.0 / kThrScale));
      }
      __syncthreads();
    }
  }
  tot = block_sum_u64<NT>(tot, sm_tot);
  if (threadIdx.x == 0) slab_tot[b * S + s] = (double)tot * (1.0 / kThrScale);
}

// Backward of the thresholding histogram: the window has zero slope, so the only path to the pixel is the weight Iy:
// dL/dIy = sum over planes of Ghat at the pixel's bin(s) -- a gather -- and dx_c = dL/dIy * x_c / Iy (store_pixel_grad).
__global__ __launch_bounds__(256) void k_hist_thr_bwd(const DevParams P, const float *__restrict__ x,
                                                      const float *__restrict__ gh, float *__restrict__ gdst) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, h = P.h;
  if (n >= P.npix) return;
  const float *xb = x + (long long)b * P.sb;
  float r, g, bl, a, bb, c, iy;
  sample_rgb(P, xb, n, r, g, bl);
  project(P, r, g, bl, a, bb, c, iy);
  float dIy = 0.f;
  if (P.intensity) {
    const double inv_step = P.step > 0.0 ? 1.0 / P.step : 0.0, w = P.half_eps * inv_step;
    const bool single = P.h > 1 && P.step > 2.0 * P.half_eps * (1.0 + 1e-9);
    for (int p = 0; p < 3; ++p) {
      if (P.green && p != 1) continue;
      const float *G = gh + ((long long)b * P.P + (P.green ? 0 : p)) * h * h;
      const float u = p == 0 ? a : (p == 1 ? -a : -bb), v = p == 0 ? bb : (p == 1 ? c : -c);
      if (single) {
        const int i = thr_single(P, u, inv_step), j = thr_single(P, v, inv_step);
        if (i >= 0 && j >= 0) dIy += G[i * h + j];
        continue;
      }
      int ul, uh, vl, vh;
      thr_range(P, u, inv_step, w, ul, uh);
      thr_range(P, v, inv_step, w, vl, vh);
      for (int i = ul; i <= uh; ++i) {
        if (!thr_hit(P, u, i)) continue;
        for (int j = vl; j <= vh; ++j)
          if (thr_hit(P, v, j)) dIy += G[i * h + j];
      }
    }
  }
  if (P.proj != HG_PROJ_RGBUV) store_pixel_grad_proj(P, xb, b, n, r, g, bl, iy, 0.f, 0.f, dIy, gdst);
  else store_pixel_grad(P, xb, b, n, r, g, bl, iy, 0.f, 0.f, 0.f, dIy, gdst);
  if (P.mode == HG_RESIZE_NONE)
    for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + n] = 0.f;
}

// XCD-aware (image, slice) of a workgroup in a (S, B) grid.  Block L = y*S + x runs on XCD L % 8 (observed placement;
// a wrong guess is slower, never wrong): with the plain mapping the S workgroups of one image sit on S different XCDs,
// and every XCD's (non-coherent) L2 fetches that image's G / hist / slabs for itself -- measured at batch 32:
// 24 MB of the thresholding backward's 49 MB of fabric reads.  Here the blocks of one XCD take whole images:
// per-XCD index j -> (image (j / S) * 8 + xcd, slice j % S).  Needs B % 8 == 0; identity otherwise.
__device__ __forceinline__ void xcd_image_slice(int B, int &b, int &s) {
  const int S = gridDim.x;
  b = blockIdx.y; s = blockIdx.x;
  if ((B & 7) == 0) {
    const int L = blockIdx.y * S + blockIdx.x, xcd = L & 7, j = L >> 3;
    s = j % S;
    b = (j / S) * 8 + xcd;
  }
}

// ---- lean scatter kernels: RGB-uv, three planes, `single` windows, all three grids in LDS ----------------------------
// The configuration every default-constructed RGBuvHistBlock(method='thresholding') has.  DIRECT: no resize and
// contiguous planes -> 16-byte loads / stores, four pixels per thread and iteration; otherwise sample_rgb (bilinear /
// sampling / strided input).  exact_only (HG_THR_EXACT=1) sends every pixel through the fp64 classification.
template <bool SYM>
__device__ __forceinline__ void thr_lean_classify(const DevParams &P, const ThrFast &F, float r, float g, float b,
                                                  double inv_step, bool exact_only, int idx[6], float &iy) {
  if (exact_only || !thr_fast_pixel<SYM>(P, F, r, g, b, idx, iy)) thr_exact_pixel(P, r, g, b, inv_step, idx, iy);
}

template <bool DIRECT, bool SYM>
__global__ __launch_bounds__(1024) void k_thr_fwd_lean(const DevParams P, const float *__restrict__ x,
                                                       float *__restrict__ slabs, double *__restrict__ slab_tot,
                                                       float *__restrict__ hist, float *__restrict__ sum_out,
                                                       int per_block, const bool exact_only) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned long long sm_tot[16];
  unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem);   // [3][h][h]
  const int S = gridDim.x, h = P.h, hh = h * h;
  int b, s;
  xcd_image_slice(P.B, b, s);
  const float *xb = x + (long long)b * P.sb;
  const int n0 = s * per_block, n1 = min(P.npix, n0 + per_block);
  const double inv_step = 1.0 / P.step;
  const ThrFast F = make_thr_fast(P);
  // the first tile's loads are in flight while the grids are cleared
  float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = r4, b4 = r4;
  int n = n0 + 4 * threadIdx.x;
  if (DIRECT && n < n1) {
    r4 = *reinterpret_cast<const float4 *>(xb + n);
    g4 = *reinterpret_cast<const float4 *>(xb + P.sc + n);
    b4 = *reinterpret_cast<const float4 *>(xb + 2 * P.sc + n);
  }
  {
    ulonglong2 *b2 = reinterpret_cast<ulonglong2 *>(bins);
    for (int e = threadIdx.x; e < (3 * hh + 1) / 2; e += 1024) b2[e] = make_ulonglong2(0ull, 0ull);  // LDS is sized in 16-byte units
  }
  __syncthreads();
  auto pixel = [&](float r, float g, float bl) __attribute__((always_inline)) {
    int idx[6];
    float iy;
    thr_lean_classify<SYM>(P, F, r, g, bl, inv_step, exact_only, idx, iy);
    const unsigned long long q = iy_fixed(iy);
    if ((idx[0] | idx[1]) >= 0) atomicAdd(&bins[idx[0] * h + idx[1]], q);
    if ((idx[2] | idx[3]) >= 0) atomicAdd(&bins[hh + idx[2] * h + idx[3]], q);
    if ((idx[4] | idx[5]) >= 0) atomicAdd(&bins[2 * hh + idx[4] * h + idx[5]], q);
  };
  if constexpr (DIRECT) {
    for (; n < n1; n += 4096) {
      const float4 rc = r4, gc = g4, bc = b4;
      if (n + 4096 < n1) {                              // next tile's loads before this tile's arithmetic
        r4 = *reinterpret_cast<const float4 *>(xb + n + 4096);
        g4 = *reinterpret_cast<const float4 *>(xb + P.sc + n + 4096);
        b4 = *reinterpret_cast<const float4 *>(xb + 2 * P.sc + n + 4096);
      }
      pixel(clamp01(rc.x), clamp01(gc.x), clamp01(bc.x));
      pixel(clamp01(rc.y), clamp01(gc.y), clamp01(bc.y));
      pixel(clamp01(rc.z), clamp01(gc.z), clamp01(bc.z));
      pixel(clamp01(rc.w), clamp01(gc.w), clamp01(bc.w));
    }
  } else {
    for (int m = n0 + threadIdx.x; m < n1; m += 1024) {
      float r, g, bl;
      sample_rgb(P, xb, m, r, g, bl);
      pixel(r, g, bl);
    }
  }
  __syncthreads();
  if (S == 1) {
    // the workgroup holds the whole image: normalise here (RGBuvHistBlock.py:224-228), no slab and no k_hist_finish
    unsigned long long t1 = 0ull;
    for (int e = threadIdx.x; e < 3 * hh; e += 1024) t1 += bins[e];
    t1 = block_sum_u64<1024>(t1, sm_tot);
    const float den = (float)((double)t1 * (1.0 / kThrScale)) + kEps;
    if (threadIdx.x == 0) sum_out[b] = den;
    float *dst = hist + (long long)b * 3 * hh;
    for (int e = threadIdx.x; e < 3 * hh; e += 1024) dst[e] = (float)((double)bins[e] * (1.0 / kThrScale)) / den;
    return;
  }
  float *slab = slabs + ((long long)(b * S + s) * 3) * hh;
  unsigned long long tot = 0ull;
  const int nq = (3 * hh) >> 2;
  if ((((uintptr_t)slab) & 15) == 0) {
    for (int e4 = threadIdx.x; e4 < nq; e4 += 1024) {
      const ulonglong2 v01 = reinterpret_cast<const ulonglong2 *>(bins)[2 * e4], v23 = reinterpret_cast<const ulonglong2 *>(bins)[2 * e4 + 1];
      tot += (v01.x + v01.y) + (v23.x + v23.y);
      reinterpret_cast<float4 *>(slab)[e4] = make_float4((float)((double)v01.x * (1.0 / kThrScale)), (float)((double)v01.y * (1.0 / kThrScale)),
                                                         (float)((double)v23.x * (1.0 / kThrScale)), (float)((double)v23.y * (1.0 / kThrScale)));
    }
    for (int e = 4 * nq + threadIdx.x; e < 3 * hh; e += 1024) {
      const unsigned long long v = bins[e];
      tot += v;
      slab[e] = (float)((double)v * (1.0 / kThrScale));
    }
  } else {
    for (int e = threadIdx.x; e < 3 * hh; e += 1024) {
      const unsigned long long v = bins[e];
      tot += v;
      slab[e] = (float)((double)v * (1.0 / kThrScale));
    }
  }
  tot = block_sum_u64<1024>(tot, sm_tot);
  if (threadIdx.x == 0) slab_tot[b * S + s] = (double)tot * (1.0 / kThrScale);
}

// One-launch backward: <G, out> is rebuilt per workgroup (2 x 48 KB from L2, as k_hist_bwd does) instead of a
// k_hist_ghat launch + a Ghat buffer; the window has no slope, so the only path to the pixel is the weight Iy:
// dL/dIy = sum_planes Ghat[bin] with Ghat = (G - <G,out>) / S' formed on the fly, dx_c = dL/dIy * x_c / Iy, clamp-masked.
// Requires intensity_scale (without it the gradient is identically zero: the host clears grad_x instead).
template <bool DIRECT, bool SYM>
__global__ __launch_bounds__(1024) void k_thr_bwd_lean(const DevParams P, const float *__restrict__ x,
                                                       const float *__restrict__ gout, const float *__restrict__ hist,
                                                       const float *__restrict__ sums, float *__restrict__ gdst,
                                                       int per_block, const bool exact_only) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float sm16[16];
  float *gh = reinterpret_cast<float *>(smem);         // Ghat [3][h][h]: the per-pixel gathers hit LDS, not L2
  const int h = P.h, hh = h * h, nel = 3 * hh;
  int b, sl;
  xcd_image_slice(P.B, b, sl);
  const float *g = gout + (long long)b * nel, *o = hist + (long long)b * nel;
  const float *xb = x + (long long)b * P.sb;
  const int n0 = sl * per_block, n1 = min(P.npix, n0 + per_block);
  // the first tile's loads are in flight during the <G, out> prologue
  float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = r4, b4 = r4;
  int n = n0 + 4 * threadIdx.x;
  if (DIRECT && n < n1) {
    r4 = *reinterpret_cast<const float4 *>(xb + n);
    g4 = *reinterpret_cast<const float4 *>(xb + P.sc + n);
    b4 = *reinterpret_cast<const float4 *>(xb + 2 * P.sc + n);
  }
  const bool al16 = (((uintptr_t)g | (uintptr_t)o) & 15) == 0;
  // <G, out> with every load in flight at once: a thread owns groups of four consecutive bins (16-byte loads), keeps
  // its G values in registers, and writes Ghat = (G - <G,out>) / S' to LDS once the sum is known.  Fixed summation order.
  constexpr int MAXQ = 5;                              // 4 * 1024 * MAXQ >= 3 h^2 for h <= 79 (the lean limit)
  float4 gq[MAXQ];
  float d = 0.f;
  const int nq = nel >> 2;                             // nel = 3 h^2; when h is odd the tail (< 4 bins) is handled below
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int e4 = threadIdx.x + k * 1024;
    gq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 oq = gq[k];
    if (e4 < nq && al16) {
      gq[k] = reinterpret_cast<const float4 *>(g)[e4];
      oq = reinterpret_cast<const float4 *>(o)[e4];
    } else if (e4 < nq) {
      gq[k] = make_float4(g[4 * e4], g[4 * e4 + 1], g[4 * e4 + 2], g[4 * e4 + 3]);
      oq = make_float4(o[4 * e4], o[4 * e4 + 1], o[4 * e4 + 2], o[4 * e4 + 3]);
    }
    d = fmaf(gq[k].x, oq.x, d); d = fmaf(gq[k].y, oq.y, d); d = fmaf(gq[k].z, oq.z, d); d = fmaf(gq[k].w, oq.w, d);
  }
  float gt = 0.f;
  const int et = 4 * nq + (int)threadIdx.x;            // tail bins (nel % 4 of them)
  if (et < nel) { gt = g[et]; d = fmaf(gt, o[et], d); }
  d = hg_wave_sum(d);
  if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = d;
  __syncthreads();
  d = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) d += sm16[w];
  const float inv = 1.f / sums[b];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int e4 = threadIdx.x + k * 1024;
    if (e4 < nq)
      reinterpret_cast<float4 *>(gh)[e4] = make_float4((gq[k].x - d) * inv, (gq[k].y - d) * inv, (gq[k].z - d) * inv, (gq[k].w - d) * inv);
  }
  if (et < nel) gh[et] = (gt - d) * inv;
  __syncthreads();
  const double inv_step = 1.0 / P.step;
  const ThrFast F = make_thr_fast(P);
  auto pixel = [&](float r, float gg, float bl, float &dr, float &dg, float &db) __attribute__((always_inline)) {
    int idx[6];
    float iy;
    thr_lean_classify<SYM>(P, F, r, gg, bl, inv_step, exact_only, idx, iy);
    float dIy = 0.f;
    if ((idx[0] | idx[1]) >= 0) dIy += gh[idx[0] * h + idx[1]];
    if ((idx[2] | idx[3]) >= 0) dIy += gh[hh + idx[2] * h + idx[3]];
    if ((idx[4] | idx[5]) >= 0) dIy += gh[2 * hh + idx[4] * h + idx[5]];
    const float wgt = dIy / iy;
    dr = wgt * r; dg = wgt * gg; db = wgt * bl;
  };
  if constexpr (DIRECT) {
    float *gb = gdst + ((long long)b * P.C) * P.npix;
    for (; n < n1; n += 4096) {
      const float4 rc = r4, gc = g4, bc = b4;
      if (n + 4096 < n1) {                              // next tile's loads before this tile's arithmetic
        r4 = *reinterpret_cast<const float4 *>(xb + n + 4096);
        g4 = *reinterpret_cast<const float4 *>(xb + P.sc + n + 4096);
        b4 = *reinterpret_cast<const float4 *>(xb + 2 * P.sc + n + 4096);
      }
      float4 or4, og4, ob4;
      pixel(clamp01(rc.x), clamp01(gc.x), clamp01(bc.x), or4.x, og4.x, ob4.x);
      pixel(clamp01(rc.y), clamp01(gc.y), clamp01(bc.y), or4.y, og4.y, ob4.y);
      pixel(clamp01(rc.z), clamp01(gc.z), clamp01(bc.z), or4.z, og4.z, ob4.z);
      pixel(clamp01(rc.w), clamp01(gc.w), clamp01(bc.w), or4.w, og4.w, ob4.w);
      // clamp mask of RGBuvHistBlock.py:76, decided on the raw value
      auto m = [&](float raw, float v) { return grad_mask(P, raw) ? v : 0.f; };
      or4 = make_float4(m(rc.x, or4.x), m(rc.y, or4.y), m(rc.z, or4.z), m(rc.w, or4.w));
      og4 = make_float4(m(gc.x, og4.x), m(gc.y, og4.y), m(gc.z, og4.z), m(gc.w, og4.w));
      ob4 = make_float4(m(bc.x, ob4.x), m(bc.y, ob4.y), m(bc.z, ob4.z), m(bc.w, ob4.w));
      *reinterpret_cast<float4 *>(gb + n) = or4;
      *reinterpret_cast<float4 *>(gb + P.npix + n) = og4;
      *reinterpret_cast<float4 *>(gb + 2LL * P.npix + n) = ob4;
      for (int cc = 3; cc < P.C; ++cc)
        *reinterpret_cast<float4 *>(gb + (long long)cc * P.npix + n) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int m = n0 + threadIdx.x; m < n1; m += 1024) {
      float r, gg, bl, dr, dg, db;
      sample_rgb(P, xb, m, r, gg, bl);
      pixel(r, gg, bl, dr, dg, db);
      store_rgb_grad(P, xb, b, m, dr, dg, db, gdst);
      if (P.mode == HG_RESIZE_NONE)
        for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + m] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RBF with a narrow kernel (the default sigma = 0.02 against a bin spacing of 6/63: exp(-d^2/sigma^2) is 1.4e-10 one bin
// away): beyond R = ceil(5.26 sigma / spacing) bins the weights are < 1e-12 -- a pixel touches (2R+1)^2 bins per plane,
// so the histogram is a scatter-add like thresholding (same fixed-point LDS grids), not a dense h x h product, and the
// backward a (2R+1)^2 gather.  Used when R <= HG_RBF_RMAX; wider kernels take the MFMA path.
constexpr int HG_RBF_RMAX = 2;

// nearest bin of u (clamped) -- the centre of the (2R+1)-bin support
__device__ __forceinline__ int rbf_center(const DevParams &P, float u, double inv_step) {
  const double t = ((double)u - P.lo) * inv_step;
  if (!(t > -1.0e6 && t < 1.0e6)) return t > 0.0 ? 2 * P.h : -P.h;
  return (int)rint(t);
}

template <bool ALL3>
__global__ __launch_bounds__(ALL3 ? 1024 : 256) void k_hist_rbf_fwd(const DevParams P, const float *__restrict__ x,
                                                                    float *__restrict__ slabs,
                                                                    double *__restrict__ slab_tot, int per_block, int R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned long long sm_tot[16];
  unsigned long long *bins = reinterpret_cast<unsigned long long *>(smem);   // [planes][h][h]
  constexpr int NT = ALL3 ? 1024 : 256;
  unsigned long long tot = 0ull;
  const int b = blockIdx.y, s = blockIdx.x, S = gridDim.x, h = P.h, hh = h * h;
  const float *xb = x + (long long)b * P.sb;
  const int n0 = s * per_block, n1 = min(P.npix, n0 + per_block);
  float *slab = slabs + ((long long)(b * S + s) * P.P) * hh;
  const double inv_step = P.step > 0.0 ? 1.0 / P.step : 0.0;
  auto scatter = [&](unsigned long long *grid, float u, float v, float iy) __attribute__((always_inline)) {
    const int ic = rbf_center(P, u, inv_step), jc = rbf_center(P, v, inv_step);
    float ku[2 * HG_RBF_RMAX + 1], kv[2 * HG_RBF_RMAX + 1];
#pragma unroll
    for (int d = 0; d < 2 * HG_RBF_RMAX + 1; ++d) {
      const int i = ic + d - R, j = jc + d - R;
      ku[d] = (d <= 2 * R && i >= 0 && i < h) ? __fmul_rn(iy, kern_eval<HG_METHOD_RBF>(P, u, make_binc(P, i, false))) : 0.f;
      kv[d] = (d <= 2 * R && j >= 0 && j < h) ? kern_eval<HG_METHOD_RBF>(P, v, make_binc(P, j, false)) : 0.f;
    }
#pragma unroll
    for (int di = 0; di < 2 * HG_RBF_RMAX + 1; ++di) {
      if (ku[di] == 0.f) continue;
#pragma unroll
      for (int dj = 0; dj < 2 * HG_RBF_RMAX + 1; ++dj) {
        const float wgt = __fmul_rn(ku[di], kv[dj]);
        const unsigned long long q = (unsigned long long)((double)wgt * kThrScale + 0.5);
        if (q) atomicAdd(&grid[(ic + di - R) * h + (jc + dj - R)], q);
      }
    }
  };
  if constexpr (ALL3) {
    for (int e = threadIdx.x; e < P.P * hh; e += NT) bins[e] = 0ull;
    __syncthreads();
    for (int n = n0 + threadIdx.x; n < n1; n += NT) {
      float r, g, bl, a, bb, c, iy;
      sample_rgb(P, xb, n, r, g, bl);
      project(P, r, g, bl, a, bb, c, iy);
      if (P.green) {
        scatter(bins, -a, c, iy);
      } else {
        scatter(bins, a, bb, iy);
        scatter(bins + hh, -a, c, iy);
        scatter(bins + 2 * hh, -bb, -c, iy);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P.P * hh; e += NT) {
      const unsigned long long v = bins[e];
      tot += v;
      slab[e] = (float)((double)v * (1.0 / kThrScale));
    }
  } else {
    for (int p = 0; p < 3; ++p) {
      if (P.green && p != 1) continue;
      for (int e = threadIdx.x; e < hh; e += NT) bins[e] = 0ull;
      __syncthreads();
      for (int n = n0 + threadIdx.x; n < n1; n += NT) {
        float r, g, bl, a, bb, c, iy;
        sample_rgb(P, xb, n, r, g, bl);
        project(P, r, g, bl, a, bb, c, iy);
        scatter(bins, p == 0 ? a : (p == 1 ? -a : -bb), p == 0 ? bb : (p == 1 ? c : -c), iy);
      }
      __syncthreads();
      float *dst = slab + (long long)(P.green ? 0 : p) * hh;
      for (int e = threadIdx.x; e < hh; e += NT) {
        const unsigned long long v = bins[e];
        tot += v;
        dst[e] = (float)((double)v * (1.0 / kThrScale));
      }
      __syncthreads();
    }
  }
  tot = block_sum_u64<NT>(tot, sm_tot);
  if (threadIdx.x == 0) slab_tot[b * S + s] = (double)tot * (1.0 / kThrScale);
}

// Backward of the truncated RBF histogram: the generic backward's two mat-vecs per plane restricted to the (2R+1)^2
// support (kernel value and slope in fp64, as k_hist_bwd_generic).
__global__ __launch_bounds__(256) void k_hist_rbf_bwd(const DevParams P, const float *__restrict__ x,
                                                      const float *__restrict__ gh, float *__restrict__ gdst, int R) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, h = P.h;
  if (n >= P.npix) return;
  const float *xb = x + (long long)b * P.sb;
  float r, g, bl, a, bb, c, iy;
  sample_rgb(P, xb, n, r, g, bl);
  project(P, r, g, bl, a, bb, c, iy);
  const double inv_step = P.step > 0.0 ? 1.0 / P.step : 0.0;
  float gu[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f}, isum = 0.f;
  for (int p = 0; p < 3; ++p) {
    if (P.green && p != 1) continue;
    const float *G = gh + ((long long)b * P.P + (P.green ? 0 : p)) * h * h;
    const float u = p == 0 ? a : (p == 1 ? -a : -bb), v = p == 0 ? bb : (p == 1 ? c : -c);
    const int ic = rbf_center(P, u, inv_step), jc = rbf_center(P, v, inv_step);
    float ku[2 * HG_RBF_RMAX + 1], dku[2 * HG_RBF_RMAX + 1], kv[2 * HG_RBF_RMAX + 1], dkv[2 * HG_RBF_RMAX + 1];
#pragma unroll
    for (int d = 0; d < 2 * HG_RBF_RMAX + 1; ++d) {
      const int i = ic + d - R, j = jc + d - R;
      ku[d] = dku[d] = kv[d] = dkv[d] = 0.f;
      if (d <= 2 * R && i >= 0 && i < h) kern_eval_d<HG_METHOD_RBF>(P, u, i, ku[d], dku[d]);
      if (d <= 2 * R && j >= 0 && j < h) kern_eval_d<HG_METHOD_RBF>(P, v, j, kv[d], dkv[d]);
    }
    float Sx[2 * HG_RBF_RMAX + 1];
#pragma unroll
    for (int dj = 0; dj < 2 * HG_RBF_RMAX + 1; ++dj) Sx[dj] = 0.f;
#pragma unroll
    for (int di = 0; di < 2 * HG_RBF_RMAX + 1; ++di) {
      const int i = ic + di - R;
      if (di > 2 * R || i < 0 || i >= h) continue;
      float T = 0.f;
#pragma unroll
      for (int dj = 0; dj < 2 * HG_RBF_RMAX + 1; ++dj) {
        const int j = jc + dj - R;
        if (dj > 2 * R || j < 0 || j >= h) continue;
        const float Gij = G[i * h + j];
        T = fmaf(Gij, kv[dj], T);
        Sx[dj] = fmaf(Gij, ku[di], Sx[dj]);
      }
      gu[p] = fmaf(dku[di], T, gu[p]);
      isum = fmaf(ku[di], T, isum);
    }
#pragma unroll
    for (int dj = 0; dj < 2 * HG_RBF_RMAX + 1; ++dj) gv[p] = fmaf(dkv[dj], Sx[dj], gv[p]);
  }
  const float da = iy * (gu[0] - gu[1]), db = iy * (gv[0] - gu[2]), dc = iy * (gv[1] - gv[2]);
  const float dIy = P.intensity ? isum : 0.f;
  if (P.proj != HG_PROJ_RGBUV) store_pixel_grad_proj(P, xb, b, n, r, g, bl, iy, iy * gu[1], iy * gv[1], dIy, gdst);
  else store_pixel_grad(P, xb, b, n, r, g, bl, iy, da, db, dc, dIy, gdst);
  if (P.mode == HG_RESIZE_NONE)
    for (int cc = 3; cc < P.C; ++cc) gdst[((long long)b * P.C + cc) * P.npix + n] = 0.f;
}

// Adjoint of the bilinear resize (deterministic gather) fused with the clamp mask.
// grad_x[b][c][y][x] = mask(x) * sum_{Y,X} wy(Y->y) wx(X->x) gxs[b][c][Y][X]
__global__ __launch_bounds__(256) void k_bilinear_adjoint(const DevParams P, const float *__restrict__ x,
                                                          const float *__restrict__ gxs, float *__restrict__ gx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)P.B * P.H * P.W;
  if (idx >= total) return;
  const int xx = (int)(idx % P.W);
  const int yy = (int)((idx / P.W) % P.H);
  const int b = (int)(idx / ((long long)P.W * P.H));
  const float inv_h = 1.f / P.rscale_h, inv_w = 1.f / P.rscale_w;
  const int Ylo = max(0, (int)floorf(((float)yy - 0.5f) * inv_h - 0.5f) - 1);
  const int Yhi = min(P.Hs - 1, (int)ceilf(((float)yy + 1.5f) * inv_h - 0.5f) + 1);
  const int Xlo = max(0, (int)floorf(((float)xx - 0.5f) * inv_w - 0.5f) - 1);
  const int Xhi = min(P.Ws - 1, (int)ceilf(((float)xx + 1.5f) * inv_w - 0.5f) + 1);
  float acc[3] = {0.f, 0.f, 0.f};
  const float *gb = gxs + (long long)b * 3 * P.npix;
  for (int Y = Ylo; Y <= Yhi; ++Y) {
    const float sy = fmaxf(__fsub_rn(__fmul_rn(P.rscale_h, (float)Y + 0.5f), 0.5f), 0.f);
    const int y0 = min((int)sy, P.H - 1);
    const float ly = clamp01(sy - (float)y0);
    const int y1 = y0 + (y0 < P.H - 1 ? 1 : 0);
    float wy = 0.f;
    if (y0 == yy) wy += 1.f - ly;
    if (y1 == yy) wy += ly;
    if (wy == 0.f) continue;
    for (int X = Xlo; X <= Xhi; ++X) {
      const float sx = fmaxf(__fsub_rn(__fmul_rn(P.rscale_w, (float)X + 0.5f), 0.5f), 0.f);
      const int x0 = min((int)sx, P.W - 1);
      const float lx = clamp01(sx - (float)x0);
      const int x1 = x0 + (x0 < P.W - 1 ? 1 : 0);
      float wx = 0.f;
      if (x0 == xx) wx += 1.f - lx;
      if (x1 == xx) wx += lx;
      if (wx == 0.f) continue;
      const float w = wy * wx;
      const long long o = (long long)Y * P.Ws + X;
      acc[0] = fmaf(w, gb[o], acc[0]);
      acc[1] = fmaf(w, gb[o + P.npix], acc[1]);
      acc[2] = fmaf(w, gb[o + 2LL * P.npix], acc[2]);
    }
  }
  const float *xb = x + (long long)b * P.sb + yy * P.sh + xx * P.sw;
  float *dst = gx + ((long long)b * P.C) * P.H * P.W + (long long)yy * P.W + xx;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xv = xb[c * P.sc];
    dst[(long long)c * P.H * P.W] = grad_mask(P, xv) ? acc[c] : 0.f;
  }
}

// Adjoint of index_select sampling (RGBuvHistBlock.py:82-89): scatter-add (indices may repeat when
// the image side is shorter than h), fused with the clamp mask.  grad_x pre-zeroed.
__global__ __launch_bounds__(256) void k_sampling_adjoint(const DevParams P, const float *__restrict__ x,
                                                          const float *__restrict__ gxs, float *__restrict__ gx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)P.B * P.npix;
  if (idx >= total) return;
  const int n = (int)(idx % P.npix), b = (int)(idx / P.npix);
  const int ys = n / P.Ws, xs = n - ys * P.Ws;
  const int yy = P.rows[ys], xx = P.cols[xs];
  const float *xb = x + (long long)b * P.sb + yy * P.sh + xx * P.sw;
  float *dst = gx + ((long long)b * P.C) * P.H * P.W + (long long)yy * P.W + xx;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xv = xb[c * P.sc];
    if (grad_mask(P, xv)) atomicAdd(dst + (long long)c * P.H * P.W, gxs[((long long)b * 3 + c) * P.npix + n]);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
struct Plan {
  int T, BLK, nbd, HP;
  int S_fwd, chunk;          // forward: splits per image, pixels per wave (multiple of 64)
  int nparts;                // reduce blocks per image
  int S_bwd, rounds;         // backward: WGs per image, 32-pixel rounds per wave
  size_t slab_bytes, part_bytes, gh_bytes, gxs_bytes;
  int planes_rt;             // > 0: backward on k_hist_bwd_planes<planes_rt> (see bwd_planes_rt)
};

int validate(const hg_hist_params *p) {
  if (!p) return HG_EINVAL;
  if (p->struct_size != sizeof(hg_hist_params)) return HG_EINVAL;   // stale header / unzeroed struct (include/hg_hist.h)
  if (p->pre_relu != 0 && p->pre_relu != 1) return HG_EINVAL;
  if (p->proj_cache && ((uintptr_t)p->proj_cache & 15)) return HG_EINVAL;
  if (p->B <= 0 || p->C < 3 || p->H <= 0 || p->W <= 0 || p->Hs <= 0 || p->Ws <= 0 || p->h <= 0) return HG_EINVAL;
  if (p->method < 0 || p->method > 2) return HG_EMETHOD;
  if (p->resize_mode < 0 || p->resize_mode > 2) return HG_ERESIZE;
  if (p->resize_mode == HG_RESIZE_SAMPLING && (!p->row_idx || !p->col_idx)) return HG_EINVAL;
  if (p->resize_mode == HG_RESIZE_NONE && (p->Hs != p->H || p->Ws != p->W)) return HG_EINVAL;
  if (!(p->hi >= p->lo)) return HG_EINVAL;
  if (p->method != HG_METHOD_THRESHOLDING && !(p->sigma > 0.0)) return HG_EINVAL;
  if ((long long)p->Hs * p->Ws > 0x7fffffffLL) return HG_EINVAL;
  if (p->projection < 0 || p->projection > 2) return HG_EINVAL;
  return HG_OK;
}

// thresholding runs on the scatter kernels when the h x h 64-bit LDS grid fits (h <= 140)
inline bool thr_scatter(const hg_hist_params *p) {
  return p->method == HG_METHOD_THRESHOLDING && (size_t)p->h * p->h * 8 <= 156 * 1024;
}

// RBF: support radius in bins beyond which exp(-d^2/sigma^2) < 1e-12; 0 = use the dense MFMA path
inline int rbf_radius(const hg_hist_params *p) {
  if (p->method != HG_METHOD_RBF || p->h < 2 || (size_t)p->h * p->h * 8 > 156 * 1024) return 0;
  if (const char *e = getenv("HG_RBF_DENSE")) if (atoi(e)) return 0;     // A/B switch for measurements
  const double step = (p->hi - p->lo) / (double)(p->h - 1);
  if (!(step > 0.0)) return 0;
  const double r = 5.2565 * p->sigma / step;          // sqrt(-ln 1e-12) = 5.2565
  const int R = (int)ceil(r);
  return (R >= 1 && R <= HG_RBF_RMAX) ? R : 0;
}

inline bool sparse_path(const hg_hist_params *p) { return thr_scatter(p) || rbf_radius(p) > 0; }

// HG_THR_EXACT=1: every window decision by the fp64 path (A/B switch of the fast classification)
inline bool thr_exact_only() {
  if (const char *e = getenv("HG_THR_EXACT")) return atoi(e) != 0;
  return false;
}

// The lean scatter kernels (k_thr_fwd_lean / k_thr_bwd_lean) apply to: RGB-uv, three planes, `single` windows (narrower
// than the bin spacing: every symmetric boundary), all three 64-bit grids in LDS at once (h <= 79).
inline bool thr_lean(const hg_hist_params *p) {
  if (!thr_scatter(p) || p->projection != HG_PROJ_RGBUV || p->green_only || p->h < 2) return false;
  if ((size_t)3 * p->h * p->h * 8 > 150 * 1024) return false;
  const double step = (p->hi - p->lo) / (double)(p->h - 1);
  const double half_eps = ((p->lo < 0 ? -p->lo : p->lo) + (p->hi < 0 ? -p->hi : p->hi)) / (double)p->h / 2.0;
  return step > 2.0 * half_eps * (1.0 + 1e-9);
}

// no resize, contiguous planes, 16-byte aligned rows of four pixels: the float4 variant
inline bool thr_direct(const hg_hist_params *p, const float *x, const float *gx) {
  const long long npix = (long long)p->H * p->W;
  return p->resize_mode == HG_RESIZE_NONE && p->stride_w == 1 && p->stride_h == p->W && (npix & 3) == 0 &&
         (p->stride_c & 3) == 0 && (p->stride_b & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)gx & 15) == 0;
}

// Which backward kernel?  0: k_hist_bwd (symmetric boundary, h <= 64, RGB-uv) or, for thresholding beyond the scatter
// path / h > 128, k_hist_bwd_generic.  RT > 0: k_hist_bwd_planes<RT> -- smooth kernels with an asymmetric boundary,
// 64 < h <= 128, or a one-plane projection.  HG_BWD_PLANES=1 sends the symmetric h <= 64 case there too (A/B runs,
// and the parity test of one MFMA formulation against the other).
inline int bwd_planes_rt(const hg_hist_params *p, int nbd) {
  if (p->method == HG_METHOD_THRESHOLDING || sparse_path(p) || p->h > 128) return 0;
  bool want = (p->lo != -p->hi) || nbd != 1 || p->projection != HG_PROJ_RGBUV;
  if (const char *e = getenv("HG_BWD_PLANES")) { if (atoi(e) == 1) want = true; else if (atoi(e) == 0) want = false; }
  return want ? (p->h + 31) / 32 : 0;
}

Plan make_plan(const hg_hist_params *p) {
  Plan pl;
  pl.T = (p->h <= 32) ? 1 : 2;
  pl.BLK = 32 * pl.T;
  pl.nbd = (p->h + pl.BLK - 1) / pl.BLK;
  pl.HP = pl.nbd * pl.BLK;
  const long long npix = (long long)p->Hs * p->Ws;
  const int P = (p->green_only || p->projection) ? 1 : 3;
  // forward: aim at ~2 workgroups per CU (256 CUs), >= 64 pixels per wave
  const long long wg_fixed = (long long)p->B * pl.nbd * pl.nbd;
  long long target = 512;
  // the scatter kernel keeps up to 98 KB of LDS grids: one workgroup per CU is all that fits, more only add slabs
  if (sparse_path(p)) target = 256;
  if (const char *e = getenv("HG_FWD_WGS")) target = atoll(e) > 0 ? atoll(e) : target;  // tuning knob
  long long S = (target + wg_fixed - 1) / wg_fixed;
  const long long maxS = (npix + 255) / 256;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  long long chunk = (npix + 4 * S - 1) / (4 * S);
  chunk = (chunk + 63) / 64 * 64;
  S = (npix + 4 * chunk - 1) / (4 * chunk);
  pl.S_fwd = (int)S;
  pl.chunk = (int)chunk;
  const long long n_per_img = (long long)P * p->h * p->h;
  pl.nparts = (int)((n_per_img + 1023) / 1024);
  pl.slab_bytes = ((size_t)p->B * S * n_per_img * sizeof(float) + 255) / 256 * 256;
  pl.part_bytes = ((size_t)p->B * pl.nparts * sizeof(float) + 255) / 256 * 256;
  // slab_tot[B][S (x bin blocks)] doubles for k_hist_finish
  pl.part_bytes = ((size_t)p->B * S * pl.nbd * pl.nbd * sizeof(double) + 255) / 256 * 256;
  // backward: 1 workgroup per CU, rounds of 32 pixels per wave
  const long long rounds_total = (npix + 31) / 32;
  long long targetb = 512;
  if (const char *e = getenv("HG_BWD_WGS")) targetb = atoll(e) > 0 ? atoll(e) : targetb;  // tuning knob
  long long Sb = (targetb + p->B - 1) / p->B;
  const long long maxSb = (rounds_total + 3) / 4;
  if (Sb > maxSb) Sb = maxSb;
  if (Sb < 1) Sb = 1;
  long long rpw = (rounds_total + 4 * Sb - 1) / (4 * Sb);
  Sb = (rounds_total + 4 * rpw - 1) / (4 * rpw);
  pl.S_bwd = (int)Sb;
  pl.rounds = (int)rpw;
  // generic backward only (h > 64 or asymmetric boundary): Ghat in natural layout
  pl.planes_rt = bwd_planes_rt(p, pl.nbd);
  if (pl.planes_rt)   // (da, db, dc, dIy) per pixel, carried from plane to plane
    pl.gh_bytes = (P == 3) ? ((size_t)p->B * 4 * npix * sizeof(float) + 255) / 256 * 256 : 256;
  else
    pl.gh_bytes = ((p->lo != -p->hi) || pl.nbd != 1 || p->projection || sparse_path(p)) ? ((size_t)p->B * n_per_img * sizeof(float) + 255) / 256 * 256 : 256;
  pl.gxs_bytes = (p->resize_mode == HG_RESIZE_NONE) ? 0 : ((size_t)p->B * 3 * npix * sizeof(float) + 255) / 256 * 256;
  return pl;
}

DevParams make_dev(const hg_hist_params *p) {
  DevParams d;
  d.B = p->B; d.C = p->C; d.H = p->H; d.W = p->W;
  d.sb = p->stride_b; d.sc = p->stride_c; d.sh = p->stride_h; d.sw = p->stride_w;
  d.Hs = p->Hs; d.Ws = p->Ws; d.mode = p->resize_mode; d.rows = p->row_idx; d.cols = p->col_idx;
  d.proj = p->projection;
  d.pre_relu = p->pre_relu ? 1 : 0;
  d.cache = (float4 *)p->proj_cache;
  d.h = p->h; d.P = (p->green_only || p->projection) ? 1 : 3; d.method = p->method;
  d.intensity = p->intensity_scale ? 1 : 0; d.green = (p->green_only || p->projection) ? 1 : 0;
  d.npix = p->Hs * p->Ws;
  d.lo = p->lo; d.hi = p->hi; d.step = (p->h > 1) ? (p->hi - p->lo) / (double)(p->h - 1) : 0.0;
  const double sigma = (p->method == HG_METHOD_THRESHOLDING) ? 1.0 : p->sigma;
  d.inv_sigma = (float)(1.0 / sigma);
  d.inv_sigma_d = (double)d.inv_sigma;
  d.half_eps = ((p->lo < 0 ? -p->lo : p->lo) + (p->hi < 0 ? -p->hi : p->hi)) / (double)p->h / 2.0;
  d.rscale_h = (float)p->H / (float)p->Hs;
  d.rscale_w = (float)p->W / (float)p->Ws;
  d.inv_sigma_x = 1.0 / sigma;
  const double ds = d.step / sigma;
  union { float f; uint32_t u; } cv;
  cv.f = (float)ds; cv.u &= 0xFFFFFF00u;  // 16 significant bits: beta0(s) (< 64) * ds_hi is exact in fp32
  d.ds_hi = cv.f; d.ds_lo = (float)(ds - (double)d.ds_hi);
  d.dk_scale = (float)(-2.0 / sigma);
  return d;
}

// The shared-reciprocal operand generation of k_hist_fwd forms products of four denominators 1 + t^2, |t| <= (13.9 +
// max|boundary|) / sigma (log-chroma differences of clamped pixels lie in [-13.82, 13.82]): taken only when that product
// stays below 1e30 (its reciprocal then is a normal float with room to spare).  HG_FWD_SHARE_RCP=0 switches it off (A/B).
// (the kernels also evaluate the PADDED bins of a 32-wide tile, up to index 32 * ceil(h / 32) - 1 >= h - 1: the bound
// takes the farthest padded bin centre lo + (padded - 1) * step, not only the boundary)
static double share_rcp_tmax(const DevParams &d) {
  const int padded = (d.h + 31) / 32 * 32;
  const double far_hi = d.lo + (double)(padded - 1) * d.step;
  double bmax = fabs(d.lo) > fabs(d.hi) ? fabs(d.lo) : fabs(d.hi);
  bmax = fabs(far_hi) > bmax ? fabs(far_hi) : bmax;
  return (13.9 + bmax) * d.inv_sigma_x;
}

bool fwd_share_rcp_ok(const DevParams &d) {
  static const bool enabled = [] { const char *e = getenv("HG_FWD_SHARE_RCP"); return !(e && e[0] == '0'); }();
  if (!enabled || d.proj != HG_PROJ_RGBUV) return false;
  const double tmax = share_rcp_tmax(d), den = 1.0 + tmax * tmax;
  return den * den * den * den < 1e30;
}

bool bwd_share_rcp_ok(const DevParams &d) {      // the same product-of-four-denominators condition; HG_BWD_SHARE_RCP=0: A/B
  static const bool enabled = [] { const char *e = getenv("HG_BWD_SHARE_RCP"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  const double tmax = share_rcp_tmax(d), den = 1.0 + tmax * tmax;
  return den * den * den * den < 1e30;
}

template <int T, int METHOD, bool GREEN>
int launch_fwd_tmg(const DevParams &d, const Plan &pl, bool sym, const float *x, float *slabs, double *slab_tot,
                   hipStream_t st) {
  const dim3 grid(pl.S_fwd, pl.nbd * pl.nbd, d.B), block(256);
  const size_t lds = 4 * kFwdStage * 16 + (size_t)3 * pl.BLK * pl.BLK * sizeof(float);
  const bool diag = pl.nbd == 1;
  if constexpr (T == 2 && METHOD == HG_METHOD_INVERSE_QUADRATIC && !GREEN) {
    if (sym && diag && fwd_share_rcp_ok(d)) {
      hipLaunchKernelGGL((k_hist_fwd<T, METHOD, true, true, GREEN, true>), grid, block, lds, st, d, x, slabs, slab_tot, pl.chunk);
      HG_LAUNCH_CHECK();
      return HG_OK;
    }
  }
  if (sym && diag) hipLaunchKernelGGL((k_hist_fwd<T, METHOD, true, true, GREEN>), grid, block, lds, st, d, x, slabs, slab_tot, pl.chunk);
  else if (sym) hipLaunchKernelGGL((k_hist_fwd<T, METHOD, true, false, GREEN>), grid, block, lds, st, d, x, slabs, slab_tot, pl.chunk);
  else hipLaunchKernelGGL((k_hist_fwd<T, METHOD, false, false, GREEN>), grid, block, lds, st, d, x, slabs, slab_tot, pl.chunk);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

template <int T, int METHOD>
int launch_fwd_tm(const DevParams &d, const Plan &pl, bool sym, const float *x, float *slabs, double *slab_tot,
                  hipStream_t st) {
  return d.green ? launch_fwd_tmg<T, METHOD, true>(d, pl, sym, x, slabs, slab_tot, st)
                 : launch_fwd_tmg<T, METHOD, false>(d, pl, sym, x, slabs, slab_tot, st);
}

template <int T>
int launch_fwd_t(const DevParams &d, const Plan &pl, bool sym, const float *x, float *slabs, double *slab_tot,
                 hipStream_t st) {
  switch (d.method) {
    case HG_METHOD_THRESHOLDING: return launch_fwd_tm<T, HG_METHOD_THRESHOLDING>(d, pl, sym, x, slabs, slab_tot, st);
    case HG_METHOD_RBF: return launch_fwd_tm<T, HG_METHOD_RBF>(d, pl, sym, x, slabs, slab_tot, st);
    default: return launch_fwd_tm<T, HG_METHOD_INVERSE_QUADRATIC>(d, pl, sym, x, slabs, slab_tot, st);
  }
}

template <int T, int METHOD>
int launch_bwd_tm(const DevParams &d, const Plan &pl, const float *x, const float *gout, const float *hist,
                  const float *sums, float *gdst, hipStream_t st) {
  const dim3 grid(pl.S_bwd, d.B), block(256);
  const size_t lds = (size_t)3 * pl.BLK * (pl.BLK + 1) * sizeof(float);
  if constexpr (METHOD == HG_METHOD_INVERSE_QUADRATIC) {
    if (!d.green && bwd_share_rcp_ok(d)) {
      hipLaunchKernelGGL((k_hist_bwd<T, METHOD, false, true>), grid, block, lds, st, d, x, gout, hist, sums, gdst, pl.rounds);
      HG_LAUNCH_CHECK();
      return HG_OK;
    }
  }
  if (d.green) hipLaunchKernelGGL((k_hist_bwd<T, METHOD, true>), grid, block, lds, st, d, x, gout, hist, sums, gdst, pl.rounds);
  else hipLaunchKernelGGL((k_hist_bwd<T, METHOD, false>), grid, block, lds, st, d, x, gout, hist, sums, gdst, pl.rounds);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

template <int T>
int launch_bwd_t(const DevParams &d, const Plan &pl, const float *x, const float *gout, const float *hist,
                 const float *sums, float *gdst, hipStream_t st) {
  switch (d.method) {
    case HG_METHOD_THRESHOLDING: return launch_bwd_tm<T, HG_METHOD_THRESHOLDING>(d, pl, x, gout, hist, sums, gdst, st);
    case HG_METHOD_RBF: return launch_bwd_tm<T, HG_METHOD_RBF>(d, pl, x, gout, hist, sums, gdst, st);
    default: return launch_bwd_tm<T, HG_METHOD_INVERSE_QUADRATIC>(d, pl, x, gout, hist, sums, gdst, st);
  }
}

template <int RT>
int launch_bwd_planes_rt(const DevParams &d, const Plan &pl, const float *x, const float *gout, const float *hist,
                         const float *sums, float *part, float *gdst, hipStream_t st) {
  const dim3 grid(pl.S_bwd, d.B), block(256);
  const size_t lds = (size_t)(32 * RT) * (32 * RT + 1) * sizeof(float);
  const void *kern = (d.method == HG_METHOD_RBF) ? (const void *)k_hist_bwd_planes<RT, HG_METHOD_RBF>
                                                 : (const void *)k_hist_bwd_planes<RT, HG_METHOD_INVERSE_QUADRATIC>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (d.method == HG_METHOD_RBF)
    hipLaunchKernelGGL((k_hist_bwd_planes<RT, HG_METHOD_RBF>), grid, block, lds, st, d, x, gout, hist, sums, part, gdst, pl.rounds);
  else
    hipLaunchKernelGGL((k_hist_bwd_planes<RT, HG_METHOD_INVERSE_QUADRATIC>), grid, block, lds, st, d, x, gout, hist, sums, part, gdst, pl.rounds);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int launch_bwd_planes(const DevParams &d, const Plan &pl, const float *x, const float *gout, const float *hist,
                      const float *sums, float *part, float *gdst, hipStream_t st) {
  switch (pl.planes_rt) {
    case 1: return launch_bwd_planes_rt<1>(d, pl, x, gout, hist, sums, part, gdst, st);
    case 2: return launch_bwd_planes_rt<2>(d, pl, x, gout, hist, sums, part, gdst, st);
    case 3: return launch_bwd_planes_rt<3>(d, pl, x, gout, hist, sums, part, gdst, st);
    default: return launch_bwd_planes_rt<4>(d, pl, x, gout, hist, sums, part, gdst, st);
  }
}

}  // namespace

extern "C" {

#if HG_HIST_PROBE
// experiment builds only (not declared in include/hg_hist.h): copy the probe counters out and reset them
int hg_debug_hist_probe(unsigned long long *out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg_probe), sizeof(unsigned long long) * 16);
  if (e != hipSuccess) return (int)e;
  unsigned long long z[16] = {0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(hg_probe), z, sizeof(z));
}
#endif

int hg_version(void) { return HG_VERSION_NUM; }

int hg_rgbuv_hist_uses_proj_cache(const hg_hist_params *p) {
  const int rc = validate(p);
  if (rc) return rc;
  return sparse_path(p) ? 0 : 1;
}

const char *hg_error_string(int code) {
  switch (code) {
    case HG_OK: return "ok";
    case HG_EINVAL: return "invalid argument";
    case HG_EMETHOD: return "Wrong kernel method. It should be either thresholding, RBF, inverse-quadratic.";
    case HG_ERESIZE: return "Wrong resizing method. It should be: interpolation or sampling.";
    case HG_EWORKSPACE: return "workspace too small";
    case HG_EUNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int hg_selftest_fastlog(float *out2, void *stream) {
  if (!out2) return HG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out2, 0, 2 * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  union { float f; uint32_t u; } a, b;
  a.f = 1e-6f; b.f = 1.000002f;
  hipLaunchKernelGGL(k_selftest_fastlog, dim3(2048), dim3(256), 0, st, a.u, b.u - a.u + 1u, (unsigned int *)out2);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_rgbuv_hist_workspace_bytes(const hg_hist_params *p, size_t *fwd_bytes, size_t *bwd_bytes) {
  const int rc = validate(p);
  if (rc) return rc;
  const Plan pl = make_plan(p);
  if (fwd_bytes) *fwd_bytes = pl.part_bytes + pl.slab_bytes;
  if (bwd_bytes) *bwd_bytes = pl.gxs_bytes + pl.gh_bytes;
  return HG_OK;
}

int hg_rgbuv_hist_fwd(const hg_hist_params *p, const float *x, float *hist_out, float *sum_out, void *workspace,
                      size_t workspace_bytes, void *stream) {
  const int rc = validate(p);
  if (rc) return rc;
  if (!x || !hist_out || !sum_out || !workspace) return HG_EINVAL;
  const Plan pl = make_plan(p);
  if (workspace_bytes < pl.part_bytes + pl.slab_bytes) return HG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const DevParams d = make_dev(p);
  float *slabs = (float *)((char *)workspace + pl.part_bytes);
  const bool sym = (p->lo == -p->hi);
  if (sparse_path(p)) {
    const size_t one = (size_t)d.h * d.h * sizeof(unsigned long long);
    const bool all3 = one * d.P <= 150 * 1024;
    const size_t lds = ((all3 ? one * d.P : one) + 15) / 16 * 16;
    const int R = rbf_radius(p);
    const void *kern = R ? (all3 ? (const void *)k_hist_rbf_fwd<true> : (const void *)k_hist_rbf_fwd<false>)
                         : (all3 ? (const void *)k_hist_thr_fwd<true> : (const void *)k_hist_thr_fwd<false>);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    const dim3 grid(pl.S_fwd, d.B), block(all3 ? 1024 : 256);
    double *slab_tot = (double *)workspace;
    if (R) {
      if (all3) hipLaunchKernelGGL(k_hist_rbf_fwd<true>, grid, block, lds, st, d, x, slabs, slab_tot, 4 * pl.chunk, R);
      else hipLaunchKernelGGL(k_hist_rbf_fwd<false>, grid, block, lds, st, d, x, slabs, slab_tot, 4 * pl.chunk, R);
    } else if (thr_lean(p)) {
      const bool ex = thr_exact_only(), dir = thr_direct(p, x, x);
      const void *lk = dir ? (sym ? (const void *)k_thr_fwd_lean<true, true> : (const void *)k_thr_fwd_lean<true, false>)
                           : (sym ? (const void *)k_thr_fwd_lean<false, true> : (const void *)k_thr_fwd_lean<false, false>);
      if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(lk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
      }
      // (S > 1: k_hist_finish sums the slabs.  Round 3 tried ONE launch -- the last-arriving workgroup of an image, found
      // through arrival flags, summing them -- and measured 152 us instead of 22: the device-scope release / acquire fences
      // that make another XCD's slabs visible write back and invalidate whole L2s, once per wave of every workgroup.)
      if (dir && sym) hipLaunchKernelGGL((k_thr_fwd_lean<true, true>), grid, block, lds, st, d, x, slabs, slab_tot, hist_out, sum_out, 4 * pl.chunk, ex);
      else if (dir) hipLaunchKernelGGL((k_thr_fwd_lean<true, false>), grid, block, lds, st, d, x, slabs, slab_tot, hist_out, sum_out, 4 * pl.chunk, ex);
      else if (sym) hipLaunchKernelGGL((k_thr_fwd_lean<false, true>), grid, block, lds, st, d, x, slabs, slab_tot, hist_out, sum_out, 4 * pl.chunk, ex);
      else hipLaunchKernelGGL((k_thr_fwd_lean<false, false>), grid, block, lds, st, d, x, slabs, slab_tot, hist_out, sum_out, 4 * pl.chunk, ex);
    } else {
      if (all3) hipLaunchKernelGGL(k_hist_thr_fwd<true>, grid, block, lds, st, d, x, slabs, slab_tot, 4 * pl.chunk);
      else hipLaunchKernelGGL(k_hist_thr_fwd<false>, grid, block, lds, st, d, x, slabs, slab_tot, 4 * pl.chunk);
    }
    HG_LAUNCH_CHECK();
    if (!R && thr_lean(p) && pl.S_fwd == 1) return HG_OK;          // normalised in the scatter kernel
    hipLaunchKernelGGL(k_hist_finish, dim3(pl.nparts, d.B), dim3(256), 0, st, slabs, slab_tot, hist_out, sum_out,
                       pl.S_fwd, pl.S_fwd, d.P * d.h * d.h);
    HG_LAUNCH_CHECK();
    return HG_OK;
  } else {
    double *slab_tot = (double *)workspace;
    int r = (pl.T == 1) ? launch_fwd_t<1>(d, pl, sym, x, slabs, slab_tot, st) : launch_fwd_t<2>(d, pl, sym, x, slabs, slab_tot, st);
    if (r) return r;
  }
  // slab sum + normalisation in one launch (the MFMA kernel left every workgroup's share of the image total)
  hipLaunchKernelGGL(k_hist_finish, dim3(pl.nparts, d.B), dim3(256), 0, st, slabs, (const double *)workspace, hist_out,
                     sum_out, pl.S_fwd, pl.S_fwd * pl.nbd * pl.nbd, d.P * d.h * d.h);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_rgbuv_hist_bwd(const hg_hist_params *p, const float *x, const float *grad_out, const float *hist_out,
                      const float *sum_out, float *grad_x, void *workspace, size_t workspace_bytes, void *stream) {
  const int rc = validate(p);
  if (rc) return rc;
  if (!x || !grad_out || !hist_out || !sum_out || !grad_x || !workspace) return HG_EINVAL;
  const Plan pl = make_plan(p);
  if (workspace_bytes < pl.gxs_bytes + pl.gh_bytes) return HG_EWORKSPACE;
  const bool sym = (p->lo == -p->hi);
  const bool generic = !sym || pl.nbd != 1 || p->projection != HG_PROJ_RGBUV;
  hipStream_t st = (hipStream_t)stream;
  const DevParams d = make_dev(p);
  float *gxs = (float *)workspace;
  const size_t gx_bytes = (size_t)d.B * d.C * d.H * d.W * sizeof(float);
  float *gdst = grad_x;
  if (d.mode != HG_RESIZE_NONE) {
    gdst = gxs;
    if (d.mode == HG_RESIZE_SAMPLING || d.C > 3) {
      hipError_t e = hipMemsetAsync(grad_x, 0, gx_bytes, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (thr_lean(p) && !p->intensity_scale && d.mode == HG_RESIZE_NONE) {
    // a 0/1 window has no slope and there is no weight to differentiate: the gradient is identically zero
    hipError_t e = hipMemsetAsync(grad_x, 0, gx_bytes, st);
    if (e != hipSuccess) return (int)e;
  } else if (thr_lean(p) && p->intensity_scale) {
    const bool ex = thr_exact_only(), dir = thr_direct(p, x, grad_x);
    const dim3 grid(pl.S_fwd, d.B), block(1024);
    const size_t blds = (size_t)3 * d.h * d.h * sizeof(float);
    if (blds > 48 * 1024) {
      const void *lk = dir ? (sym ? (const void *)k_thr_bwd_lean<true, true> : (const void *)k_thr_bwd_lean<true, false>)
                           : (sym ? (const void *)k_thr_bwd_lean<false, true> : (const void *)k_thr_bwd_lean<false, false>);
      hipError_t e = hipFuncSetAttribute(lk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
      if (e != hipSuccess) return (int)e;
    }
    if (dir && sym) hipLaunchKernelGGL((k_thr_bwd_lean<true, true>), grid, block, blds, st, d, x, grad_out, hist_out, sum_out, gdst, 4 * pl.chunk, ex);
    else if (dir) hipLaunchKernelGGL((k_thr_bwd_lean<true, false>), grid, block, blds, st, d, x, grad_out, hist_out, sum_out, gdst, 4 * pl.chunk, ex);
    else if (sym) hipLaunchKernelGGL((k_thr_bwd_lean<false, true>), grid, block, blds, st, d, x, grad_out, hist_out, sum_out, gdst, 4 * pl.chunk, ex);
    else hipLaunchKernelGGL((k_thr_bwd_lean<false, false>), grid, block, blds, st, d, x, grad_out, hist_out, sum_out, gdst, 4 * pl.chunk, ex);
    HG_LAUNCH_CHECK();
  } else if (sparse_path(p)) {
    float *gh = (float *)((char *)workspace + pl.gxs_bytes);
    hipLaunchKernelGGL(k_hist_ghat, dim3(d.B), dim3(1024), 0, st, grad_out, hist_out, sum_out, gh, d.P * d.h * d.h);
    HG_LAUNCH_CHECK();
    const int R = rbf_radius(p);
    if (R) hipLaunchKernelGGL(k_hist_rbf_bwd, dim3((d.npix + 255) / 256, d.B), dim3(256), 0, st, d, x, gh, gdst, R);
    else hipLaunchKernelGGL(k_hist_thr_bwd, dim3((d.npix + 255) / 256, d.B), dim3(256), 0, st, d, x, gh, gdst);
    HG_LAUNCH_CHECK();
  } else if (pl.planes_rt) {
    float *part = (float *)((char *)workspace + pl.gxs_bytes);
    int r = launch_bwd_planes(d, pl, x, grad_out, hist_out, sum_out, part, gdst, st);
    if (r) return r;
  } else if (!generic) {
    int r = (pl.T == 1) ? launch_bwd_t<1>(d, pl, x, grad_out, hist_out, sum_out, gdst, st)
                        : launch_bwd_t<2>(d, pl, x, grad_out, hist_out, sum_out, gdst, st);
    if (r) return r;
  } else {
    float *gh = (float *)((char *)workspace + pl.gxs_bytes);
    const int n_per_img = d.P * d.h * d.h;
    hipLaunchKernelGGL(k_hist_ghat, dim3(d.B), dim3(1024), 0, st, grad_out, hist_out, sum_out, gh, n_per_img);
    HG_LAUNCH_CHECK();
    const size_t lds = (size_t)d.h * 64 * sizeof(float);
    if (lds > 160 * 1024) return HG_EUNSUPPORTED;  // h > 640
    const dim3 grid((d.npix + 63) / 64, d.B), block(64);
    switch (d.method) {
      case HG_METHOD_THRESHOLDING:
        hipLaunchKernelGGL((k_hist_bwd_generic<HG_METHOD_THRESHOLDING>), grid, block, lds, st, d, x, gh, gdst); break;
      case HG_METHOD_RBF:
        hipLaunchKernelGGL((k_hist_bwd_generic<HG_METHOD_RBF>), grid, block, lds, st, d, x, gh, gdst); break;
      default:
        hipLaunchKernelGGL((k_hist_bwd_generic<HG_METHOD_INVERSE_QUADRATIC>), grid, block, lds, st, d, x, gh, gdst); break;
    }
    HG_LAUNCH_CHECK();
  }
  if (d.mode == HG_RESIZE_BILINEAR) {
    const long long total = (long long)d.B * d.H * d.W;
    hipLaunchKernelGGL(k_bilinear_adjoint, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, x, gxs, grad_x);
    HG_LAUNCH_CHECK();
  } else if (d.mode == HG_RESIZE_SAMPLING) {
    const long long total = (long long)d.B * d.npix;
    hipLaunchKernelGGL(k_sampling_adjoint, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, x, gxs, grad_x);
    HG_LAUNCH_CHECK();
  }
  return HG_OK;
}

}  // extern "C"
