/* hg_conv.h -- C ABI of the dense contraction of the HistoGAN networks (libhistogan_hip.so).
 *
 * Hand-written fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM convolutions for gfx950.  They
 * replace the `F.conv2d` calls of the reference:
 *   Conv2DMod.forward           histoGAN/histoGAN.py:420-440  (grouped conv with per-sample weights;
 *                               here: shared weights, modulation as a per-(b,k) input scale and
 *                               demodulation as a per-(b,n) output scale, both fused into the kernel)
 *   RGBBlock.conv (1x1)         histoGAN/histoGAN.py:375, 383
 *   DiscriminatorBlock convs    histoGAN/histoGAN.py:510-518  (3x3 / 1x1, stride 1; bias fused)
 * and the three autograd products of such a convolution (output, data gradient, weight gradient).
 *
 * Scope: square kernels ksize 1 or 3, stride 1, dilation 1, "same" zero padding (ksize/2), fp32,
 * contiguous NCHW, any B, K, N, H, W >= 1.  Conventions as in hg_hist.h: return 0 / negative HG_E* /
 * positive hipError_t; device pointers; work is enqueued on `stream`; nothing is allocated or synchronised.
 */
#ifndef HG_CONV_H
#define HG_CONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_CONV_PACK_FWD 0   /* Wt[t][ci][co] = W[co][ci][t]             (K = Ci, N = Co) */
#define HG_CONV_PACK_DGRAD 1 /* Wt[t][co][ci] = W[co][ci][ksize^2-1-t]   (K = Co, N = Ci) */

/* Packed-weight operand of hg_conv2d_same: Wt[ksize^2][Kp][Np] fp32, Kp = roundup(K,16),
 * Np = roundup(N,128), zero padded.  Returns the number of floats. */
size_t hg_conv_packed_elems(int32_t Co, int32_t Ci, int32_t ksize, int32_t mode);
/* w: (Co, Ci, ksize, ksize) contiguous -> wt (hg_conv_packed_elems floats, fully written). */
int hg_conv_pack_weights(const float *w, float *wt, int32_t Co, int32_t Ci, int32_t ksize, int32_t mode,
                         void *stream);

/* out[b,n,y,x] = oscale[b,n] * sum_{k,dy,dx} iscale[b,k] * in[b,k,y+dy-p,x+dx-p] * Wt[dy*ksize+dx][k][n] + bias[n]
 *   in (B,K,H,W), out (B,N,H,W), p = ksize/2, zeros outside the image.
 *   iscale (B,K), oscale (B,N), bias (N): each may be NULL (= 1, 1, 0).
 * With HG_CONV_PACK_FWD weights this is the forward convolution (iscale = style+1, oscale = demod);
 * with HG_CONV_PACK_DGRAD weights and in = grad_out it is the data gradient. */
int hg_conv2d_same(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                   const float *bias, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, int32_t ksize,
                   void *stream);

/* Weight gradient: gw[n,k,dy,dx] = sum_{b,y,x} gscale[b,n]*gout[b,n,y,x] * iscale[b,k]*in[b,k,y+dy-p,x+dx-p]
 *   gw (N,K,ksize,ksize) contiguous, fully written (deterministic: split-K slabs in the workspace are
 *   summed in a fixed order, no atomics).  iscale / gscale may be NULL. */
size_t hg_conv2d_wgrad_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, int32_t ksize);
int hg_conv2d_wgrad(const float *in, const float *gout, float *gw, const float *iscale, const float *gscale,
                    int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, int32_t ksize, void *workspace,
                    size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_CONV_H */
