// xcd_handoff.hip -- can Z workgroups (on different XCDs) hand partial tiles to the last arriver WITHOUT device-scope fences?
//
// Round 3 folded the K-split reduction into k_conv with __threadfence() / acquire-release atomics and measured +15 ms per
// train step: on this multi-XCD part a device-scope release is buffer_wbl2 (write back the XCD's whole L2) and an acquire is
// buffer_inv (invalidate it), per workgroup, under the other workgroups' operand reuse.  The alternative measured here:
//   producers  : partial tile written with sc0 sc1 (write-through to device scope) stores, s_waitcnt vmcnt(0), then a
//                RELAXED agent-scope atomic add on the tile's arrival counter (no fence instruction at all)
//   last one   : reads the other Z-1 tiles with sc0 sc1 (bypass the non-coherent L2) loads and writes the sum
// against (a) the two-launch form (plain stores, second kernel sums) and (b) the fenced form.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_handoff.hip -o tools/ubench/xcd_handoff && tools/ubench/xcd_handoff [tiles] [Z]
// Prints time per variant and whether the sums are right (a stale read shows as a wrong sum).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE = 4096;   // floats per partial tile (16 KB), 256 threads x 4 float4

__device__ __forceinline__ void store_wt(float *p, f32x4 v) {      // write-through to device scope
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_dev(const float *p) {        // device-scope load (misses the local L2)
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ f32x4 load_dev_nt(const float *p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ f32x4 partial(int tile, int z, int e, unsigned tag) {
  const float b = (float)((tile * 131 + z * 17 + e + tag * 7) & 1023) * (1.f / 1024.f);
  return f32x4{b, b + 1.f, b + 2.f, b + 3.f};
}

// MODE 0: plain stores only (the producer half of the two-launch form)
// MODE 1: fenced hand-off (__threadfence + relaxed atomic, acquire fence in the last arriver)
// MODE 2: write-through stores + relaxed atomic, device-scope loads in the last arriver, no fence
// MODE 3: write-through stores + relaxed atomic; the last arriver invalidates its L2 (acquire fence only: buffer_inv, no
//         buffer_wbl2 anywhere) and reads with plain loads
// MODE 4: as 2 with nt sc0 sc1 loads
// MODE 5: plain stores, RELEASE-only fence in every producer (buffer_wbl2, no invalidate), ACQUIRE-only fence in the last
//         arriver (buffer_inv): the minimum the memory model asks for
// MODE 6: as 5 with write-through stores (is the write-back cheap when nothing is dirty?)
// MODE 7: the Z producers of a tile on ONE XCD (block id = 8 Z q + 8 z + x -> tile 8 q + x: workgroups go to the XCDs round
//         robin by id), whose L2 is then the coherence point: plain stores, s_waitcnt, relaxed atomic; the last arriver
//         reads with sc1 loads (past its CU's L1, from the shared L2).  `xcd_mismatch` counts tiles whose producers did
//         not all report the finisher's XCC_ID.
__device__ __forceinline__ f32x4 load_l2(const float *p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ unsigned g_xcd_mismatch;

template <int MODE>
__global__ __launch_bounds__(256) void k_produce(float *slabs, float *out, unsigned *cnt, int tiles, int Z, unsigned tag) {
  int tile = blockIdx.x % tiles, z = blockIdx.x / tiles;     // the Z producers of a tile are `tiles` block ids apart
  unsigned *xcc = cnt + tiles;                               // [tiles][Z] XCC ids (MODE 7)
  if (MODE == 7) {
    const int q = blockIdx.x / (8 * Z), r = blockIdx.x % (8 * Z);
    tile = q * 8 + (r & 7); z = r >> 3;
    if (threadIdx.x == 0) xcc[tile * Z + z] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
  }
  float *mine = slabs + ((size_t)z * tiles + tile) * TILE;
  for (int e = threadIdx.x; e < TILE / 4; e += 256) {
    const f32x4 v = partial(tile, z, e, tag);
    if ((MODE >= 2 && MODE <= 4) || MODE == 6) store_wt(mine + 4 * e, v);
    else *reinterpret_cast<f32x4 *>(mine + 4 * e) = v;
  }
  if (MODE == 0) return;
  __shared__ unsigned last;
  if (MODE == 1) __threadfence();
  else if (MODE == 5 || MODE == 6) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else __builtin_amdgcn_s_waitcnt(0);                 // vmcnt(0): this wave's stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag * Z + Z - 1;
  __syncthreads();
  if (!last) return;
  if (MODE == 1) __threadfence();
  if (MODE == 3 || MODE == 5 || MODE == 6) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (MODE == 7 && threadIdx.x == 0) {
    const unsigned me = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
    bool same = true;
    for (int zz = 0; zz < Z; ++zz) same &= __hip_atomic_load(xcc + tile * Z + zz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == me;
    if (!same) atomicAdd(&g_xcd_mismatch, 1u);
  }
  for (int e = threadIdx.x; e < TILE / 4; e += 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < Z; ++zz) {
      const float *p = slabs + ((size_t)zz * tiles + tile) * TILE + 4 * e;
      s += MODE == 2 ? load_dev(p) : MODE == 4 ? load_dev_nt(p) : MODE == 7 ? load_l2(p) : *reinterpret_cast<const f32x4 *>(p);
    }
    *reinterpret_cast<f32x4 *>(out + (size_t)tile * TILE + 4 * e) = s;
  }
}

__global__ __launch_bounds__(256) void k_reduce(const float *slabs, float *out, int tiles, int Z) {
  const int tile = blockIdx.x;
  for (int e = threadIdx.x; e < TILE / 4; e += 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < Z; ++zz) s += *reinterpret_cast<const f32x4 *>(slabs + ((size_t)zz * tiles + tile) * TILE + 4 * e);
    *reinterpret_cast<f32x4 *>(out + (size_t)tile * TILE + 4 * e) = s;
  }
}

int main(int argc, char **argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 1024, Z = argc > 2 ? atoi(argv[2]) : 4, iters = 50;
  float *slabs, *out;
  unsigned *cnt;
  hipMalloc(&slabs, (size_t)Z * tiles * TILE * 4);
  hipMalloc(&out, (size_t)tiles * TILE * 4);
  hipMalloc(&cnt, (size_t)tiles * 4 * (1 + Z));
  std::vector<float> h((size_t)tiles * TILE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto check = [&](const char *name, float ms) {
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int t = 0; t < tiles; ++t)
      for (int e = 0; e < TILE / 4; ++e) {
        float want = 0.f;
        for (int z = 0; z < Z; ++z) want += (float)((t * 131 + z * 17 + e + (iters + 1) * 7) & 1023) * (1.f / 1024.f);
        if (h[(size_t)t * TILE + 4 * e] != want) ++bad;
      }
    printf("%-44s %8.2f us per launch   wrong sums: %ld\n", name, ms / iters * 1e3f, bad);
  };
  auto run = [&](int mode) {
    hipMemset(cnt, 0, tiles * 4);
    hipMemset(out, 0, (size_t)tiles * TILE * 4);
    for (int it = -2; it < iters; ++it) {
      if (it == 0) hipEventRecord(e0);
      const unsigned tag = (unsigned)(it + 2);
      if (mode == 0) {
        hipLaunchKernelGGL(k_produce<0>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
        hipLaunchKernelGGL(k_reduce, dim3(tiles), dim3(256), 0, 0, slabs, out, tiles, Z);
      } else if (mode == 1) {
        hipLaunchKernelGGL(k_produce<1>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else if (mode == 2) {
        hipLaunchKernelGGL(k_produce<2>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else if (mode == 3) {
        hipLaunchKernelGGL(k_produce<3>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else if (mode == 4) {
        hipLaunchKernelGGL(k_produce<4>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else if (mode == 5) {
        hipLaunchKernelGGL(k_produce<5>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else if (mode == 6) {
        hipLaunchKernelGGL(k_produce<6>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      } else {
        hipLaunchKernelGGL(k_produce<7>, dim3(tiles * Z), dim3(256), 0, 0, slabs, out, cnt, tiles, Z, tag);
      }
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
  };
  printf("%d tiles of %d floats, %d partial tiles each\n", tiles, TILE, Z);
  check("two launches (plain stores + reduce kernel)", run(0));
  check("one launch, __threadfence hand-off", run(1));
  check("one launch, sc1 stores / loads, no fence", run(2));
  check("one launch, sc1 stores, inv-only acquire", run(3));
  check("one launch, sc1 stores, nt sc0 sc1 loads", run(4));
  check("one launch, release-only / acquire-only", run(5));
  check("one launch, sc1 stores + release / acquire", run(6));
  if (tiles % 8 == 0) {
    check("one launch, producers of a tile on one XCD", run(7));
    unsigned mm = 0;
    hipMemcpyFromSymbol(&mm, HIP_SYMBOL(g_xcd_mismatch), 4);
    printf("  xcd_mismatch (tile finishes whose producers sat on another XCD, over all launches): %u\n", mm);
  }
  return 0;
}
