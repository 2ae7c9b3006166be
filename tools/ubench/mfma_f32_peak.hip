// Calibration: pure v_mfma_f32_32x32x2_f32 throughput with the forward kernel's accumulator shape
// (12 independent 32x32 tiles per wave, 1 or 2 waves per SIMD).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k_mfma(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  int iters = 4096;
  float *out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 512, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<12>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * 4 * iters * 12 * 4096.0;
      if (rep == 2) printf("grid %4d (x4 waves) 12 acc: %.3f ms  %.1f TFLOP/s\n", grid, ms, flop / ms / 1e9);
    }
  }
  for (int grid : {512}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters * 3, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * 4 * iters * 3 * 4 * 4096.0;
      if (rep == 2) printf("grid %4d (x4 waves)  4 acc: %.3f ms  %.1f TFLOP/s\n", grid, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
