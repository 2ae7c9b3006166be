// hg_common.h -- shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HG_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HG_LAUNCH_CHECK()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

__device__ __forceinline__ float hg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float hg_block_sum_256(float v, float *sm4 /* >= 4 floats of LDS */) {
  v = hg_wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm4[w] = v;
  __syncthreads();
  float r = (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]);
  __syncthreads();
  return r;
}
