// Do VALU instructions issue in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?  K independent VALU ops per MFMA (fma
// chains on registers the MFMAs do not touch), scheduled in groups of G MFMAs followed by G*K VALU ops; the MFMA operands
// are loop-invariant or (DEP) produced by those VALU ops; TRANS adds one v_rcp_f32 per MFMA.  2 waves per SIMD at
// grid 512.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int K, int G, int DEP, int TRANS>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  float v[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) v[j] = a0 * (j + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(DEP ? v[i % 12] : a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float &x = v[(i * K + j) % 12];
        if (TRANS && j == 0) x = __builtin_amdgcn_rcpf(x) + 1.0f;
        else x = __builtin_fmaf(x, 1.0001f, 0.5f);
      }
    }
#pragma unroll
    for (int g = 0; g < NACC / G; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
      if (K + TRANS > 0) __builtin_amdgcn_sched_group_barrier(0x002, G * (K + TRANS), 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int j = 0; j < 12; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int K, int G, int DEP, int TRANS>
void run(float *out, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2048;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, K, G, DEP, TRANS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  double flop = (double)grid * 4 * iters * NACC * 4096.0;
  printf("%2d acc  %2d VALU/MFMA  groups of %2d MFMA  dep %d  rcp %d   grid %4d: %7.3f ms  %6.1f TFLOP/s (%.2f of 157.3)\n", NACC, K, G,
         DEP, TRANS, grid, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int grid = 512;
  run<12, 0, 1, 0, 0>(out, grid);
  run<12, 2, 1, 0, 0>(out, grid); run<12, 2, 2, 0, 0>(out, grid); run<12, 2, 3, 0, 0>(out, grid); run<12, 2, 6, 0, 0>(out, grid); run<12, 2, 12, 0, 0>(out, grid);
  run<12, 3, 1, 0, 0>(out, grid); run<12, 3, 3, 0, 0>(out, grid); run<12, 3, 6, 0, 0>(out, grid); run<12, 3, 12, 0, 0>(out, grid);
  run<12, 4, 1, 0, 0>(out, grid); run<12, 4, 3, 0, 0>(out, grid); run<12, 4, 6, 0, 0>(out, grid); run<12, 4, 12, 0, 0>(out, grid);
  run<12, 6, 1, 0, 0>(out, grid); run<12, 6, 6, 0, 0>(out, grid); run<12, 6, 12, 0, 0>(out, grid);
  run<12, 3, 1, 1, 0>(out, grid); run<12, 3, 6, 1, 0>(out, grid); run<12, 3, 12, 1, 0>(out, grid);
  run<12, 2, 1, 0, 1>(out, grid); run<12, 2, 6, 0, 1>(out, grid); run<12, 2, 12, 0, 1>(out, grid);
  run<6, 3, 1, 0, 0>(out, grid); run<6, 3, 6, 0, 0>(out, grid);
  return 0;
}
