// k_hist_fwd's K loop rebuilt piece by piece, with CYCLES (s_memtime) next to wall time: separates the shader clock under
// this instruction mix from the cycle efficiency of the loop.  2 waves per SIMD (grid 512 x 256 threads), 12 accumulator
// tiles, 12 v_mfma_f32_32x32x2_f32 per K step, variants of the operand generation between the MFMA groups:
//   0  none (operands loop-invariant)                     -- the matrix pipe alone
//   1  round 4's block: 18 packed VALU + 2 v_rcp_f32 (shared reciprocals), inline asm
//   2  round 3's form: 11 packed VALU + 6 v_rcp_f32 (compiler-scheduled C++)
//   3  variant 1 + the ds_read_b128 of the (a, b, c, w) tuple per step
//   4  variant 1 with the block issued BEFORE the MFMA group of the same step (operands consumed one step later)
//   5  variant 1, two steps' blocks back to back, then 24 MFMAs
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hist_fwd_loop.hip -o tools/ubench/hist_fwd_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Ops { float A0[2], A2[2], B0[2], B1[2]; };

__device__ __forceinline__ void gen_asm(const f32x4 &q, float inv_sigma, const f32x2 &chiA, const f32x2 &cloA,
                                        const f32x2 &chiB, const f32x2 &cloB, Ops &o) {
  const f32x2 qxy = __builtin_shufflevector(q, q, 0, 1), qzw = __builtin_shufflevector(q, q, 2, 3);
  const f32x2 is2 = {inv_sigma, inv_sigma};
  f32x2 a0, a2, b0, b1;
  asm volatile(
      "v_pk_fma_f32 v[244:245], %4, %6, %7 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 v[246:247], %4, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 v[248:249], %5, %6, %9 op_sel_hi:[0,1,1]\n\t"
      "v_pk_add_f32 v[244:245], v[244:245], %8\n\t"
      "v_pk_add_f32 v[246:247], v[246:247], %8\n\t"
      "v_pk_add_f32 v[248:249], v[248:249], %10\n\t"
      "v_pk_fma_f32 v[244:245], v[244:245], v[244:245], 1.0 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 v[246:247], v[246:247], v[246:247], 1.0 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 v[248:249], v[248:249], v[248:249], 1.0 op_sel_hi:[1,1,0]\n\t"
      "v_pk_mul_f32 v[250:251], v[244:245], v[246:247]\n\t"
      "v_mul_f32 v253, v248, v249\n\t"
      "v_mul_f32 v252, v250, v251\n\t"
      "v_rcp_f32 v253, v253\n\t"
      "v_rcp_f32 v252, v252\n\t"
      "s_nop 0\n\t"
      "v_pk_mul_f32 %3, v[248:249], v[252:253] op_sel:[1,1] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 v[250:251], v[250:251], v[252:253] op_sel:[1,0] op_sel_hi:[0,0]\n\t"
      "v_pk_mul_f32 v[254:255], v[250:251], %5 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %2, v[244:245], v[250:251]\n\t"
      "v_pk_mul_f32 %0, v[246:247], v[254:255]\n\t"
      "v_pk_mul_f32 %1, v[244:245], v[254:255]\n\t"
      "s_nop 1"
      : "=&v"(a0), "=&v"(a2), "=&v"(b0), "=&v"(b1)
      : "v"(qxy), "v"(qzw), "s"(is2), "v"(chiA), "v"(cloA), "v"(chiB), "v"(cloB)
      : "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
  o.A0[0] = a0.x; o.A0[1] = a0.y; o.A2[0] = a2.x; o.A2[1] = a2.y;
  o.B0[0] = b0.x; o.B0[1] = b0.y; o.B1[0] = b1.x; o.B1[1] = b1.y;
}

__device__ __forceinline__ void gen_cpp(const f32x4 &q, float inv_sigma, const f32x2 &chiA, const f32x2 &cloA,
                                        const f32x2 &chiB, const f32x2 &cloB, Ops &o) {
  auto iq2 = [&](float u, const f32x2 &chi, const f32x2 &clo) __attribute__((always_inline)) -> f32x2 {
    const f32x2 uu = {u, u}, is = {inv_sigma, inv_sigma}, one = {1.f, 1.f};
    const f32x2 t = __builtin_elementwise_fma(uu, is, chi) + clo;
    const f32x2 den = __builtin_elementwise_fma(t, t, one);
    return f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  };
  const f32x2 w2 = {q.w, q.w};
  const f32x2 ka = iq2(q.x, chiA, cloA), kb = iq2(q.y, chiA, cloA), kc = iq2(q.z, chiB, cloB);
  const f32x2 a0 = w2 * ka, a2 = w2 * kb;
  o.A0[0] = a0.x; o.A0[1] = a0.y; o.A2[0] = a2.x; o.A2[1] = a2.y;
  o.B0[0] = kb.x; o.B0[1] = kb.y; o.B1[0] = kc.x; o.B1[1] = kc.y;
}

#define MFMAS(U)                                                                                           \
  _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj) {      \
    acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(U.A0[ti], U.B0[tj], acc[0][ti][tj], 0, 0, 0);    \
    acc[1][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(U.A0[ti], U.B1[tj], acc[1][ti][tj], 0, 0, 0);    \
    acc[2][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(U.A2[ti], U.B1[tj], acc[2][ti][tj], 0, 0, 0);    \
  }

template <int V>
__global__ __launch_bounds__(256, 2) void k(float *out, long long *cyc, int steps, float inv_sigma) {
  __shared__ f32x4 stage[4 * 68];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int e = threadIdx.x; e < 4 * 68; e += 256) stage[e] = f32x4{0.01f * e, -0.02f * e, 0.005f * e, 1.f};
  __syncthreads();
  f32x16 acc[3][2][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;
  const f32x2 chiA = {0.1f * lane, 0.1f * lane + 3.f}, cloA = {1e-6f, 2e-6f}, chiB = {-0.1f * lane, 1.f}, cloB = {3e-7f, 1e-7f};
  const f32x4 *srow = stage + wave * 68 + (lane >> 5);
  f32x4 qa = srow[0], qb = srow[2];
  Ops opA, opB;
  gen_cpp(qa, inv_sigma, chiA, cloA, chiB, cloB, opA);
  gen_cpp(qb, inv_sigma, chiA, cloA, chiB, cloB, opB);
  const long long t0 = __builtin_readcyclecounter();
  for (int m = 0; m < steps; m += 2) {
    if constexpr (V == 0) {
      MFMAS(opA); __builtin_amdgcn_sched_barrier(0); MFMAS(opB); __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (V == 1 || V == 3) {
      if constexpr (V == 3) qb = srow[2 * (m & 31)];
      MFMAS(opA); __builtin_amdgcn_sched_barrier(0);
      gen_asm(qa, inv_sigma, chiA, cloA, chiB, cloB, opB); __builtin_amdgcn_sched_barrier(0);
      if constexpr (V == 3) { asm volatile("" : "+v"(qb)); qa = srow[2 * (m & 31) + 2]; }
      MFMAS(opB); __builtin_amdgcn_sched_barrier(0);
      gen_asm(qb, inv_sigma, chiA, cloA, chiB, cloB, opA); __builtin_amdgcn_sched_barrier(0);
      if constexpr (V == 3) asm volatile("" : "+v"(qa));
    } else if constexpr (V == 2) {
      gen_cpp(qa, inv_sigma, chiA, cloA, chiB, cloB, opB);
      MFMAS(opA);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); __builtin_amdgcn_sched_group_barrier(0x002, 48, 0);
      asm volatile("" : "+v"(qa));
      gen_cpp(qb, inv_sigma, chiA, cloA, chiB, cloB, opA);
      MFMAS(opB);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); __builtin_amdgcn_sched_group_barrier(0x002, 48, 0);
      asm volatile("" : "+v"(qb));
    } else if constexpr (V == 4) {
      gen_asm(qa, inv_sigma, chiA, cloA, chiB, cloB, opB); __builtin_amdgcn_sched_barrier(0);
      MFMAS(opA); __builtin_amdgcn_sched_barrier(0);
      gen_asm(qb, inv_sigma, chiA, cloA, chiB, cloB, opA); __builtin_amdgcn_sched_barrier(0);
      MFMAS(opB); __builtin_amdgcn_sched_barrier(0);
    } else {
      Ops t1, t2;
      gen_asm(qa, inv_sigma, chiA, cloA, chiB, cloB, t1); __builtin_amdgcn_sched_barrier(0);
      gen_asm(qb, inv_sigma, chiA, cloA, chiB, cloB, t2); __builtin_amdgcn_sched_barrier(0);
      MFMAS(opA); MFMAS(opB); __builtin_amdgcn_sched_barrier(0);
      opA = t1; opB = t2;
    }
    qa.x += 1e-3f; qb.y -= 1e-3f;
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int V>
void run(float *out, long long *cyc, const char *what) {
  const int grid = 512, steps = 4096;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, cyc, steps, 50.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  static long long h[2048];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; long long mx = 0;
  for (int i = 0; i < 2048; ++i) { avg += h[i]; if (h[i] > mx) mx = h[i]; }
  avg /= 2048;
  // A SIMD runs two waves and is busy until its LAST wave ends: cycles per K step of the SIMD = the longest wave / (2 x steps).
  // The AVERAGE wave time is not the SIMD's time: arbitration is oldest-first, so with nothing but MFMAs the older wave of a SIMD
  // takes every issue slot and ends at half time, the younger one runs alone afterwards -- average = 0.75 x longest, which is the
  // "576 cycles" round 4's file printed for variant 0 (0.75 x 768).  Both are printed; the ratio shows how the waves shared.
  const double cyc_per_step = (double)mx / steps / 2.0, cyc_avg = avg / steps / 2.0;
  const double flop = (double)grid * 4 * steps * 12 * 4096.0;
  printf("variant %d %-44s %7.3f ms %6.1f TFLOP/s (%.3f of 157.3) | %7.1f cycles per step per SIMD from its last wave (768 = matrix pipe full: %.3f); "
         "from the average wave %.1f (x%.3f) | clock %.0f MHz (longest wave's s_memtime / wall)\n",
         V, what, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, cyc_per_step, 768.0 / cyc_per_step, cyc_avg, avg / (double)mx, (double)mx / (ms * 1e3));
}

int main() {
  float *out; long long *cyc;
  (void)hipMalloc(&out, 512 * 256 * 4);
  (void)hipMalloc(&cyc, 2048 * 8);
  run<0>(out, cyc, "MFMA only");
  run<1>(out, cyc, "+ asm block (18 pk VALU + 2 rcp) after");
  run<2>(out, cyc, "+ C++ generation (11 pk VALU + 6 rcp)");
  run<3>(out, cyc, "+ asm block + ds_read_b128");
  run<4>(out, cyc, "+ asm block before the MFMA group");
  run<5>(out, cyc, "two blocks, then 24 MFMAs");
  run<0>(out, cyc, "MFMA only (again)");
  return 0;
}
