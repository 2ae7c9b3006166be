/* hg_recolor.h -- C ABI of the ReHistoGAN-specific kernels (libhistogan_hip.so); SURVEY.md section 8, row f-1.
 *
 * The recolouring encoder-decoder reuses the convolution kernels of hg_conv.h and the generator-block kernels of
 * hg_nets.h; what it adds to the path (reference file ReHistoGAN/rehistoGAN.py) is
 *   nn.InstanceNorm2d + LeakyReLU(0.2)       EncoderBlock.net, :489-496  (affine=False, biased variance, eps 1e-5)
 *   sobel_op / laplacian_op                  :235-256, reconstruction_loss :279-326: F.conv2d with ONE 3x3 stencil
 *                                            expanded over the 3 input channels -> 1 output channel, padding 1
 *   gaussian_op                              :207-232: 15x15 depthwise filter, NO padding (variance loss :1022-1029)
 *
 * Conventions as in hg_hist.h: return 0 / negative HG_E* / positive hipError_t; device pointers; fp32; contiguous;
 * enqueue on `stream`; never allocate or synchronise.
 */
#ifndef HG_RECOLOR_H
#define HG_RECOLOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scratch for the two instance-norm entry points over P planes (partials combined in fixed order: deterministic). */
size_t hg_instnorm_workspace_bytes(int64_t P);

/* out[p,:] = lrelu_slope((x[p,:] - mean_p) * rstd_p),  rstd_p = 1/sqrt(var_p + eps), var biased, over the HW
 * elements of plane p (P = B*C planes).  stats[p] = {mean_p, rstd_p} (2*P floats) is kept for the backward. */
int hg_instnorm_lrelu_fwd(const float *x, float *out, float *stats, int64_t P, int32_t HW, float eps, float slope,
                          void *workspace, size_t workspace_bytes, void *stream);
/* m = gout * (out > 0 ? 1 : slope);  xhat recovered from out;
 * gx = rstd * (m - mean(m) - xhat * mean(m * xhat))   per plane. */
int hg_instnorm_lrelu_bwd(const float *gout, const float *out, const float *stats, float *gx, int64_t P, int32_t HW,
                          float slope, void *workspace, size_t workspace_bytes, void *stream);

/* out[b,0,y,x] = sum_c sum_{i,j} taps[3i+j] * x[b,c,y+i-1,x+j-1]   (zero padding; x: (B,C,H,W), out: (B,1,H,W)).
 * adjoint != 0: x is (B,1,H,W), out is (B,C,H,W):  out[b,c,y,x] = sum_{i,j} taps[3i+j] * x[b,0,y-i+1,x-j+1]. */
int hg_stencil3(const float *x, float *out, const float *taps9_host, int32_t B, int32_t C, int32_t H, int32_t W,
                int32_t adjoint, void *stream);

/* Depthwise KSxKS filter (the SAME KSxKS kernel `k`, a device pointer, on every plane; KS odd, <= 15).
 * adjoint == 0:  x (P,H,W) -> out (P,H-KS+1,W-KS+1), out[y,x] = sum k[i,j] x[y+i,x+j]          (no padding)
 * adjoint != 0:  x (P,H-KS+1,W-KS+1) -> out (P,H,W),  out[y,x] = sum k[i,j] x[y-i,x-j]          (its transpose)
 * H, W are always the sizes of the LARGER (unfiltered) image. */
int hg_depthwise_valid(const float *x, const float *k, float *out, int64_t P, int32_t H, int32_t W, int32_t KS,
                       int32_t adjoint, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_RECOLOR_H */
