/* hg_linear.h -- C ABI of the grouped linear layers (libhistogan_hip.so).
 *
 * The generator's 21 style projections -- GeneratorBlock.to_style1 / to_style2 and RGBBlock.to_style, nn.Linear(512, C)
 * each (histoGAN/histoGAN.py:372, 450, 454; applied at :462-470, 381) -- are 21 skinny products (B x 512) @ (512 x C)
 * with B = 32 and C = 32 ... 2048 per train-step forward, and 42 more in its backward.  As library GEMMs each is a
 * 12 ... 28 us launch of 32 ... 64 workgroups (1.8 ms of a 46 ms step, three quarters of the chip idle); together they are
 * 0.4 GFLOP over 25 MB of weights.  Here ALL layers of a pass are ONE launch: the layers are described by a small table
 * passed BY VALUE (host memory, copied into the kernel arguments -- nothing is uploaded), rows of 32 output features per
 * wavefront on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation in k order).
 *
 * Conventions as in hg_hist.h: return 0 / negative HG_E* / positive hipError_t; device pointers; fp32; row-major contiguous;
 * enqueue on `stream`; never allocate or synchronise.  Limits: batch <= 64, in_features % 32 == 0,
 * out_features % 4 == 0 per layer, x / w / y 16-byte aligned, <= HG_GLIN_MAX layers (anything else: HG_EUNSUPPORTED).
 */
#ifndef HG_LINEAR_H
#define HG_LINEAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_GLIN_MAX 32

/* One nn.Linear(K, N): y = x @ w^T + b.  Layers with the same `group` share their input x (the three projections of one
 * generator block); groups are numbered 0 .. n_groups-1 and the layers of a group are adjacent in the table. */
typedef struct hg_glin_layer {
  const float *x;   /* (B, K) input of the layer's group                                             */
  const float *w;   /* (N, K) weight                                                                  */
  const float *b;   /* (N) bias, or NULL                                     [forward]                 */
  float *y;         /* (B, N): output [forward]; the incoming gradient dL/dy [backward, read-only]    */
  float *gw;        /* (N, K) weight gradient, overwritten                   [backward_params]         */
  float *gb;        /* (N) bias gradient, overwritten, or NULL               [backward_params]         */
  int32_t N;
  int32_t group;
} hg_glin_layer;

/* y_l = x_g(l) @ w_l^T + b_l for every layer: one launch. */
int hg_grouped_linear_fwd(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K, void *stream);

/* gx_g = sum over the layers l of group g of dy_l @ w_l  -- gx: n_groups pointers (host array) to (B, K) outputs,
 * overwritten.  The sum over each group's output features is split over workgroups into slabs combined in fixed order
 * (deterministic): two launches.  workspace: hg_grouped_linear_bwd_input_workspace_bytes. */
size_t hg_grouped_linear_bwd_input_workspace_bytes(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K);
int hg_grouped_linear_bwd_input(const hg_glin_layer *layers, int32_t n_layers, float *const *gx, int32_t n_groups, int32_t B,
                                int32_t K, void *workspace, size_t workspace_bytes, void *stream);

/* gw_l = dy_l^T @ x_g(l),  gb_l = sum_b dy_l[b, :]  for every layer: one launch. */
int hg_grouped_linear_bwd_params(const hg_glin_layer *layers, int32_t n_layers, int32_t B, int32_t K, void *stream);

#ifdef __cplusplus
}
#endif
#endif
