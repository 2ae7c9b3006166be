// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the Winograd kernels: every kernel below reads a
// 1 GiB buffer (4x the Infinity Cache) exactly once, so the counter's answer can be compared with a known byte count.
//   stream16   16 B per lane, lanes contiguous (the pattern the guide calibrated: reports 1/2)
//   stream4     4 B per lane, lanes contiguous
//   rows8      k_wino's patch rows: a wave instruction reads two 256-byte image rows, 16 B per lane at an 8-byte lane stride,
//              starting one element left of the row (dword-aligned only; neighbouring lanes overlap by 8 B)
//   tiles8     k_wino_wgrad's rows: 8 channels x 8 tiles per wave instruction, 16 B per lane at an 8-byte stride inside a
//              64-byte segment per channel (segments 4 KiB apart)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
// (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- <repo>/tools/ubench/fetch_calib)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

__global__ __launch_bounds__(256) void stream16(const f32x4 *p, float *out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = p[i];
    s += v[0] + v[1] + v[2] + v[3];
  }
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void stream4(const float *p, float *out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
  if (s == 123.456f) out[0] = s;
}
// slab = 256 MiB window (32-bit buffer offsets); rows of 64 floats
__global__ __launch_bounds__(256) void rows8(const float *p, float *out, int rows_per_slab, int slabs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int sl = 0; sl < slabs; ++sl) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + (size_t)sl * rows_per_slab * 64, 0,
                                                                         rows_per_slab * 256, 0x00020000);
    for (int w = blockIdx.x * 4 + wave; w < rows_per_slab / 2; w += gridDim.x * 4) {
      const int row = 2 * w + (lane >> 5), x = 2 * (lane & 31) - 1;
      const unsigned off = (unsigned)(row * 64 + (x < 0 ? 0 : x)) * 4u;     // (the last lane's load reaches one element into the next row)
      const f32x4 v = bload4(r, off);
      s += v[0] + v[1] + v[2] + v[3];
    }
  }
  if (s == 123.456f) out[0] = s;
}
// "channels" 4 KiB apart (a 32 x 32 map), 8 tiles of a row per channel: one 64-byte segment per channel and instruction
__global__ __launch_bounds__(256) void tiles8(const float *p, float *out, int ch_per_slab, int slabs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = lane & 7, c8 = lane >> 3;
  float s = 0.f;
  for (int sl = 0; sl < slabs; ++sl) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + (size_t)sl * ch_per_slab * 1024, 0,
                                                                         ch_per_slab * 4096, 0x00020000);
    for (int g = blockIdx.x * 4 + wave; g < ch_per_slab / 8; g += gridDim.x * 4)   // 8 channels per wave
      for (int seg = 0; seg < 64; ++seg) {                                          // the channel's 64 segments of 64 B
        const unsigned e = (unsigned)((g * 8 + c8) * 1024 + seg * 16 + 2 * t);
        const f32x4 v = bload4(r, (e > 0 ? e - 1 : 0) * 4u);
        s += v[0] + v[1] + v[2] + v[3];
      }
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  const size_t bytes = 1ull << 30;
  float *buf, *out;
  if (hipMalloc(&buf, bytes + 4096) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, bytes + 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int k = 0; k < 4; ++k) {
      (void)hipEventRecord(e0);
      if (k == 0) hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const f32x4 *)buf, out, bytes / 16);
      if (k == 1) hipLaunchKernelGGL(stream4, dim3(4096), dim3(256), 0, 0, buf, out, bytes / 4);
      if (k == 2) hipLaunchKernelGGL(rows8, dim3(4096), dim3(256), 0, 0, buf, out, (int)((256u << 20) / 256), 4);
      if (k == 3) hipLaunchKernelGGL(tiles8, dim3(4096), dim3(256), 0, 0, buf, out, (int)((256u << 20) / 4096), 4);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      static const char *nm[4] = {"stream16", "stream4", "rows8", "tiles8"};
      if (rep == 2) printf("%-9s reads %zu bytes once: %.3f ms  %.2f TB/s\n", nm[k], bytes, ms, bytes / ms / 1e9);
    }
  return 0;
}
