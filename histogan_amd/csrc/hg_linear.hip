// hg_linear.hip -- grouped nn.Linear layers (include/hg_linear.h): the generator's 21 style projections (histoGAN/histoGAN.py:
// 372, 450, 454) as ONE launch per pass instead of 21 library GEMMs of 32 ... 64 workgroups each.
//
// All three passes are skinny products with one dimension = the batch (<= 64): they run on v_mfma_f32_32x32x2_f32 with the
// batch as one 32-wide side of the tile, operands straight from global memory (the whole problem is 25 MB of weights and
// < 1 MB of activations at 256^2 / capacity 16: L2-resident after the first touch), every wave with all of its loads of a
// K range in flight before its MFMAs.  The layer table travels in the kernel arguments.
//
//   forward         y[b][n]  = sum_k x[b][k] w[n][k] + bias[n]      D[i = n][j = b], K range split over the 4 waves of a block
//   backward input  gx[b][k] = sum_l sum_n dy_l[b][n] w_l[n][k]     D[i = b][j = k], one wave per (layer, 128 n, 32 k): slabs
//                                                                   summed in fixed order by k_glin_sum
//   backward params gw[n][k] = sum_b dy[b][n] x[b][k], gb = sum_b   D[i = n][j = k], the batch is the (short) reduction
#include "hg_common.h"
#include "../../include/hg_hist.h"
#include "../../include/hg_linear.h"

namespace {

struct GArgs {
  hg_glin_layer L[HG_GLIN_MAX];
  int first[HG_GLIN_MAX + 1];   // first work tile of layer l (forward / params: 32-row tiles; input: 128-row chunks)
  int n, B, K;
};

__device__ __forceinline__ int find_layer(const GArgs &a, int t) {
  int l = 0;
  while (l + 1 < a.n && t >= a.first[l + 1]) ++l;
  return l;
}

// C/D layout of v_mfma_f32_32x32x2_f32: lane holds column j = lane & 31, rows i(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- forward ---------------------------------------------------------------------------------------------------------
// block = one tile of 32 output features; wave w takes k in [w K/4, (w+1) K/4); partial tiles combined through LDS in wave
// order (deterministic).  A[i = n][kk] = w[n][k], B[kk][j = b] = x[b][k]: lane (l & 31, l >> 5) reads 16 bytes of its row.
template <int TB>
__global__ __launch_bounds__(256) void k_glin_fwd(const GArgs a) {
  __shared__ float red[4][TB][32 * 33];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = lane >> 5, l31 = lane & 31;
  const int t = blockIdx.x;
  const int l = find_layer(a, t);
  const hg_glin_layer L = a.L[l];
  const int n0 = (t - a.first[l]) * 32, K = a.K, B = a.B;
  const int n = min(n0 + l31, L.N - 1);
  const float *wrow = L.w + (long long)n * K;
  const float *xrow[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) xrow[tb] = L.x + (long long)min(32 * tb + l31, B - 1) * K;
  f32x16 acc[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
  const int kq = K >> 2, k_lo = wave * kq;
  for (int kb = k_lo; kb < k_lo + kq; kb += 64) {        // 8 steps of 8 k: all loads of the batch first
    float4 wv[8], xv[TB][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + 8 * u + 4 * half;
      const bool ok = kb + 8 * u < k_lo + kq;
      wv[u] = ok ? *reinterpret_cast<const float4 *>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tb = 0; tb < TB; ++tb)
        xv[tb][u] = ok ? *reinterpret_cast<const float4 *>(xrow[tb] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u].x, xv[tb][u].x, acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u].y, xv[tb][u].y, acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u].z, xv[tb][u].z, acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u].w, xv[tb][u].w, acc[tb], 0, 0, 0);
      }
  }
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][tb][l31 * 33 + mfma_row(r, half)] = acc[tb][r];   // [b][n]
  __syncthreads();
  for (int e = threadIdx.x; e < TB * 1024; e += 256) {
    const int tb = e >> 10, b = 32 * tb + ((e >> 5) & 31), i = e & 31;
    if (b >= B || n0 + i >= L.N) continue;
    const int o = ((e >> 5) & 31) * 33 + i;
    float v = ((red[0][tb][o] + red[1][tb][o]) + red[2][tb][o]) + red[3][tb][o];
    if (L.b) v += L.b[n0 + i];
    L.y[(long long)b * L.N + n0 + i] = v;
  }
}

// ---- backward, input side ---------------------------------------------------------------------------------------------
// wave = (layer l, chunk of 128 output features, 32 input features): slab[b][k] = sum_{n in chunk} dy[b][n] w[n][k].
// A[i = b][kk = n] = dy[b][n] (16-byte loads along n), B[kk = n][j = k] = w[n][k0 + j] (coalesced rows).  MFMA step s uses
// n = n0 + s (half 0) and n0 + 64 + s (half 1).
template <int TB>
__global__ __launch_bounds__(256) void k_glin_bwd_input(const GArgs a, float *__restrict__ slabs) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = lane >> 5, l31 = lane & 31;
  const int K = a.K, B = a.B, kt_n = K >> 5;
  const int item = blockIdx.x * 4 + wave;                  // (chunk, k tile), chunk-major
  const int chunk = item / kt_n, kt = item - chunk * kt_n;
  if (chunk >= a.first[a.n]) return;
  const int l = find_layer(a, chunk);
  const hg_glin_layer L = a.L[l];
  const int n0 = (chunk - a.first[l]) * 128 + 64 * half, k0 = 32 * kt;
  f32x16 acc[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
  for (int sb = 0; sb < 64; sb += 16) {                     // 16 MFMA steps per batch of loads
    float4 dv[TB][4];
    float wv[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        const int b = 32 * tb + l31, nn = n0 + sb + 4 * u;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && nn < L.N) {                            // (N is a multiple of 4 for every layer here; guarded anyway)
          const float *p = L.y + (long long)b * L.N + nn;
          if (nn + 3 < L.N) v = *reinterpret_cast<const float4 *>(p);
          else { v.x = p[0]; if (nn + 1 < L.N) v.y = p[1]; if (nn + 2 < L.N) v.z = p[2]; }
        }
        dv[tb][u] = v;
      }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int nn = n0 + sb + u;
      wv[u] = L.w[(long long)min(nn, L.N - 1) * K + k0 + l31];   // (rows past N meet dy = 0)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[tb][u].x, wv[4 * u + 0], acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[tb][u].y, wv[4 * u + 1], acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[tb][u].z, wv[4 * u + 2], acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[tb][u].w, wv[4 * u + 3], acc[tb], 0, 0, 0);
      }
  }
  float *slab = slabs + (long long)chunk * B * K;
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int b = 32 * tb + mfma_row(r, half);
      if (b < B) slab[(long long)b * K + k0 + l31] = acc[tb][r];
    }
}

// gx_g[e] = sum of the slabs of group g in chunk order
struct SumArgs { float *gx[HG_GLIN_MAX]; int c0[HG_GLIN_MAX + 1]; int n_groups; };

__global__ __launch_bounds__(256) void k_glin_sum(const SumArgs a, const float *__restrict__ slabs, int BK) {
  const int g = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= BK) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = a.c0[g]; c < a.c0[g + 1]; ++c) {
    const float4 v = *reinterpret_cast<const float4 *>(slabs + (long long)c * BK + e);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4 *>(a.gx[g] + e) = s;
}

// ---- backward, parameter side -------------------------------------------------------------------------------------------
// block = 32 output features n; wave w takes the k tiles w, w + 4, ...  A[i = n][kk = b] = dy[b][n0 + i] (kept in registers),
// B[kk = b][j = k] = x[b][k0 + j].  The reduction runs over the batch only: B / 2 MFMAs per 32 x 32 tile of gw.
template <int TB>
__global__ __launch_bounds__(256) void k_glin_bwd_params(const GArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = lane >> 5, l31 = lane & 31;
  const int t = blockIdx.x;
  const int l = find_layer(a, t);
  const hg_glin_layer L = a.L[l];
  const int n0 = (t - a.first[l]) * 32, K = a.K, B = a.B;
  const int n = n0 + l31;
  constexpr int NS = 16 * TB;                               // MFMA steps over the (zero-padded) batch
  float dy[NS];
  float bsum = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int b = 2 * s + half;
    dy[s] = (b < B && n < L.N) ? L.y[(long long)b * L.N + n] : 0.f;
    bsum += dy[s];
  }
  if (wave == 0 && L.gb) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (half == 0 && n < L.N) L.gb[n] = bsum;
  }
  for (int kt = wave; kt < (K >> 5); kt += 4) {
    const int k0 = 32 * kt;
    float xv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int b = min(2 * s + half, B - 1);               // (rows past B meet dy = 0)
      xv[s] = L.x[(long long)b * K + k0 + l31];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dy[s], xv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nn = n0 + mfma_row(r, half);
      if (nn < L.N) L.gw[(long long)nn * K + k0 + l31] = acc[r];
    }
  }
}

int check_layers(const hg_glin_layer *layers, int n_layers, int B, int K, bool params) {
  if (!layers || n_layers <= 0 || n_layers > HG_GLIN_MAX || B <= 0 || K <= 0) return HG_EINVAL;
  if (B > 64 || (K & 31)) return HG_EUNSUPPORTED;
  for (int l = 0; l < n_layers; ++l) {
    const hg_glin_layer &L = layers[l];
    if (!L.x || !L.w || !L.y || L.N <= 0 || L.group < 0 || L.group >= HG_GLIN_MAX) return HG_EINVAL;
    if (l && (L.group < layers[l - 1].group || L.group > layers[l - 1].group + 1)) return HG_EINVAL;
    if (params && !L.gw) return HG_EINVAL;
    if (((uintptr_t)L.x | (uintptr_t)L.w | (uintptr_t)L.y) & 15) return HG_EINVAL;   // 16-byte loads along rows
    if (L.N & 3) return HG_EUNSUPPORTED;                                                // (rows of dy read 16 bytes at a time)
  }
  return layers[0].group == 0 ? HG_OK : HG_EINVAL;
}

void fill_args(GArgs &a, const hg_glin_layer *layers, int n_layers, int B, int K, int rows_per_tile) {
  a.n = n_layers; a.B = B; a.K = K;
  int t = 0;
  for (int l = 0; l < n_layers; ++l) {
    a.L[l] = layers[l];
    a.first[l] = t;
    t += (layers[l].N + rows_per_tile - 1) / rows_per_tile;
  }
  for (int l = n_layers; l <= HG_GLIN_MAX; ++l) a.first[l] = t;
}

}  // namespace

extern "C" {

int hg_grouped_linear_fwd(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K, void *stream) {
  const int rc = check_layers(layers, n_layers, B, K, false);
  if (rc) return rc;
  GArgs a;
  fill_args(a, layers, n_layers, B, K, 32);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.first[n_layers]), block(256);
  if (B <= 32) hipLaunchKernelGGL(k_glin_fwd<1>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(k_glin_fwd<2>, grid, block, 0, st, a);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

size_t hg_grouped_linear_bwd_input_workspace_bytes(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K) {
  if (check_layers(layers, n_layers, B, K, false)) return 0;
  size_t chunks = 0;
  for (int l = 0; l < n_layers; ++l) chunks += (size_t)(layers[l].N + 127) / 128;
  return chunks * (size_t)B * K * sizeof(float);
}

int hg_grouped_linear_bwd_input(const hg_glin_layer *layers, int32_t n_layers, float *const *gx, int32_t n_groups, int32_t B,
                                int32_t K, void *workspace, size_t workspace_bytes, void *stream) {
  const int rc = check_layers(layers, n_layers, B, K, false);
  if (rc) return rc;
  if (!gx || n_groups != layers[n_layers - 1].group + 1 || !workspace) return HG_EINVAL;
  if (workspace_bytes < hg_grouped_linear_bwd_input_workspace_bytes(layers, n_layers, B, K)) return HG_EWORKSPACE;
  GArgs a;
  fill_args(a, layers, n_layers, B, K, 128);
  SumArgs s;
  s.n_groups = n_groups;
  for (int g = 0, l = 0; g < n_groups; ++g) {
    if (!gx[g] || ((uintptr_t)gx[g] & 15)) return HG_EINVAL;
    s.gx[g] = gx[g];
    s.c0[g] = a.first[l];
    while (l < n_layers && layers[l].group == g) ++l;
    s.c0[g + 1] = a.first[l];
  }
  hipStream_t st = (hipStream_t)stream;
  const int items = a.first[n_layers] * (K >> 5);
  const dim3 grid((items + 3) / 4), block(256);
  if (B <= 32) hipLaunchKernelGGL(k_glin_bwd_input<1>, grid, block, 0, st, a, (float *)workspace);
  else hipLaunchKernelGGL(k_glin_bwd_input<2>, grid, block, 0, st, a, (float *)workspace);
  HG_LAUNCH_CHECK();
  const int BK = B * K;
  hipLaunchKernelGGL(k_glin_sum, dim3((BK / 4 + 255) / 256, n_groups), dim3(256), 0, st, s, (const float *)workspace, BK);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

int hg_grouped_linear_bwd_params(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K, void *stream) {
  const int rc = check_layers(layers, n_layers, B, K, true);
  if (rc) return rc;
  GArgs a;
  fill_args(a, layers, n_layers, B, K, 32);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.first[n_layers]), block(256);
  if (B <= 32) hipLaunchKernelGGL(k_glin_bwd_params<1>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(k_glin_bwd_params<2>, grid, block, 0, st, a);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
