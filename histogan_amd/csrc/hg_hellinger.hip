// hg_hellinger.hip -- Hellinger histogram loss, forward + gradient in two small launches.
//
// Reference: histoGAN/histoGAN.py:54,957-960 (and Histogram_loss.ipynb:415-417 without alpha):
//   loss = alpha * (1/sqrt 2) * sqrt( sum_{b,p,i,j} (sqrt(t) - sqrt(g))^2 ) / B
// ONE sqrt over the whole batch.  d loss / d g = alpha/(sqrt2 * B) * 1/(2 D) * (1 - sqrt(t)/sqrt(g)),
// D = the outer sqrt; like the reference autograd it is inf/NaN where g == 0 or D == 0.
#include "hg_common.h"
#include "../../include/hg_hist.h"

namespace {

constexpr int kMaxBlocks = 1024;

__global__ __launch_bounds__(256) void k_hell_partial(const float *__restrict__ t, const float *__restrict__ g,
                                                      long long n, float *__restrict__ partials) {
  __shared__ float sm4[4];
  float acc = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float d = sqrtf(t[e]) - sqrtf(g[e]);
    acc = fmaf(d, d, acc);
  }
  acc = hg_block_sum_256(acc, sm4);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_hell_final(const float *__restrict__ t, const float *__restrict__ g,
                                                    long long n, int batch, float alpha,
                                                    const float *__restrict__ partials, int nparts,
                                                    float *__restrict__ loss_out, float *__restrict__ grad) {
  __shared__ float sm4[4];
  float acc = 0.f;
  for (int k = threadIdx.x; k < nparts; k += 256) acc += partials[k];
  const float ssum = hg_block_sum_256(acc, sm4);
  const float D = sqrtf(ssum);
  const float scale = alpha * 0.70710678118654752440f / (float)batch;
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss_out = scale * D;
  if (grad) {
    const float c = scale * 0.5f / D;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
      grad[e] = c * (1.f - sqrtf(t[e]) / sqrtf(g[e]));
  }
}

inline int nblocks(long long n) {
  long long b = (n + 1023) / 1024;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

}  // namespace

extern "C" {

size_t hg_hellinger_workspace_bytes(int64_t n) { return (size_t)kMaxBlocks * sizeof(float); }

int hg_hellinger_fwd_bwd(const float *target, const float *gen, int64_t n, int32_t batch, float alpha,
                         float *loss_out, float *grad_gen, void *workspace, size_t workspace_bytes, void *stream) {
  if (!target || !gen || !loss_out || !workspace || n <= 0 || batch <= 0) return HG_EINVAL;
  if (workspace_bytes < (size_t)kMaxBlocks * sizeof(float)) return HG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float *partials = (float *)workspace;
  const int nb = nblocks(n);
  hipLaunchKernelGGL(k_hell_partial, dim3(nb), dim3(256), 0, st, target, gen, (long long)n, partials);
  HG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_hell_final, dim3(nb), dim3(256), 0, st, target, gen, (long long)n, (int)batch, alpha,
                     partials, nb, loss_out, grad_gen);
  HG_LAUNCH_CHECK();
  return HG_OK;
}

}  // extern "C"
