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
 * Scope: square kernels ksize 1 or 3, stride 1 (or 2 for ksize 3: the discriminator's down-sampling
 * convolution, :517-518), dilation 1, zero padding ksize/2, fp32, contiguous NCHW, any B, K, N, H, W >= 1.  Conventions as in hg_hist.h: return 0 / negative HG_E* /
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

/* Packed-weight operand of hg_conv2d_fwd / hg_conv2d_dgrad: Wt[ksize^2][Kp][Np] fp32, Kp = roundup(K,16),
 * Np = roundup(N,128), zero padded.  Returns the number of floats. */
size_t hg_conv_packed_elems(int32_t Co, int32_t Ci, int32_t ksize, int32_t mode);
/* w: (Co, Ci, ksize, ksize) contiguous -> wt (hg_conv_packed_elems floats, fully written). */
int hg_conv_pack_weights(const float *w, float *wt, int32_t Co, int32_t Ci, int32_t ksize, int32_t mode,
                         void *stream);

/* Both packings from one read of w (wt_fwd: hg_conv_packed_elems(.., HG_CONV_PACK_FWD) floats, wt_dgrad: .._DGRAD). */
int hg_conv_pack_weights_both(const float *w, float *wt_fwd, float *wt_dgrad, int32_t Co, int32_t Ci, int32_t ksize,
                              void *stream);

/* Output of the convolution (stride 1 or 2; stride 2 needs ksize 3), p = ksize/2, zeros outside the image:
 *   out[b,n,y,x] = oscale[b,n] * sum_{k,dy,dx} iscale[b,k] * in[b,k,y*stride+dy-p,x*stride+dx-p] * Wt[dy*ksize+dx][k][n] + bias[n]
 *   in (B,K,Hi,Wi), out (B,N,Ho,Wo) with Ho = (Hi-1)/stride + 1;  wt packed with HG_CONV_PACK_FWD.
 *   iscale (B,K), oscale (B,N), bias (N): each may be NULL (= 1, 1, 0)   (iscale = style+1, oscale = demod). */
int hg_conv2d_fwd(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                  const float *bias, int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                  int32_t stride, void *workspace, size_t workspace_bytes, void *stream);

/* out = conv(in) + bias + addend, addend (B,N,Ho,Wo) like out: the residual sum of a discriminator block
 * (`x = self.net(x); x = x + res`, histoGAN/histoGAN.py:520-524) folded into the epilogue of its 1x1 `conv_res`
 * launch -- (conv + bias) + addend in that order, i.e. bit-identical to the convolution followed by an add.
 * No input / output scales.  Workspace as hg_conv2d_workspace_bytes(..., dgrad 0). */
int hg_conv2d_fwd_add(const float *in, const float *wt, float *out, const float *addend, const float *bias, int32_t B,
                      int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride, void *workspace,
                      size_t workspace_bytes, void *stream);

/* The generator's modulated convolution stage in ONE launch (Conv2DMod.forward + noise + LeakyReLU of
 * GeneratorBlock.forward, histoGAN/histoGAN.py:420-440 and :465-476), stride 1:
 *   out[b,n,y,x] = lrelu_slope( oscale[b,n] * sum_{k,dy,dx} iscale[b,k]*in[b,k,y+dy-p,x+dx-p]*Wt[dy*ksize+dx][k][n]
 *                               + bias[n] + noise_w[n] * noise_img[b,y,x] )
 *   iscale = style+1 (modulation), oscale = demodulation coefficient, bias / noise_w = to_noise Linear(1,N) bias /
 *   weight, noise_img (B, noise_S, noise_S) with noise_S >= H, W (the noise image, already transposed as the
 *   reference's permute requires, see hg_nets.h), lrelu_slope 0 = no activation.  Every pointer but in/wt/out may be
 *   NULL (noise_w and noise_img only together).  Workspace as hg_conv2d_workspace_bytes(..., stride 1, dgrad 0). */
int hg_modconv2d_fwd(const float *in, const float *wt, float *out, const float *iscale, const float *oscale,
                     const float *bias, const float *noise_w, const float *noise_img, int32_t noise_S,
                     float lrelu_slope, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, int32_t ksize,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Scratch for hg_conv2d_fwd (dgrad = 0) / hg_conv2d_dgrad (dgrad = 1) with the same B,K,N,Hi,Wi,ksize,stride:
 * launches with few output pixels and many channels (the 2x2 ... 8x8 maps) split the reduction over K into
 * slabs that a second kernel sums in fixed order.  0 = none needed; workspace may also be NULL (no K split). */
size_t hg_conv2d_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                                 int32_t stride, int32_t dgrad);

/* Both packed operands of MANY weights in one launch (a model's convolution weights after an optimizer step: ~50
 * launches of hg_conv_pack_weights_both become one per flat buffer).  `items_dev`: device array of n_items descriptors,
 * block_begin = running sum of hg_conv_pack_blocks(Co, Ci) over the preceding items (first = 0); total_blocks = the
 * sum over all items.  ksize 1 or 3; wt_fwd / wt_dgrad sized by hg_conv_packed_elems as for the single call. */
typedef struct hg_pack_item {
  const float *w;
  float *wt_fwd, *wt_dgrad;
  int32_t Co, Ci, ksize, block_begin;
  /* optional (NULL: none): wsq[co][ci] = sum_taps W[co][ci][t]^2, (Co, Ci) contiguous -- the weight-only factor of the
   * demodulation coefficient d[b,o] = rsqrt(sum_i (s[b,i]+1)^2 wsq[o][i] + 1e-8) (Conv2DMod, histoGAN/histoGAN.py:427-429),
   * formed from the tile the packing already holds instead of a pow + reduce pass over every weight per step */
  float *wsq;
} hg_pack_item;
int32_t hg_conv_pack_blocks(int32_t Co, int32_t Ci);
int hg_conv_pack_weights_multi(const hg_pack_item *items_dev, int32_t n_items, int32_t total_blocks, void *stream);

/* The launch plan hg_conv2d_fwd (dgrad = 0) / the stride-1 hg_conv2d_dgrad (dgrad = 1) take for these arguments (host logic
 * only, no device work; with no GPU present 256 CUs are assumed):
 *   out[0] tile (0: 16 ch x 256 px, 1: 32 x 256, 2: 64 x 256, 3: 128 x 128, 4: 128 x 128 small-map, 5: 64 x 64),
 *   out[1] K split, out[2] input channels per K chunk, out[3] blocks of the launch, out[4] CUs planned for.
 * The block count decides the K split and the chunk size: a launch should be a whole number of rounds of
 * (CUs x blocks per CU) -- DESIGN.md section 8. */
int hg_conv2d_plan(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride, int32_t dgrad,
                   int32_t out[5]);

/* Data gradient of that convolution:  gin (B,N,Hi,Wi) <- gout (B,K,Ho,Wo), K = the convolution's OUTPUT
 * channels, N = its INPUT channels, (Hi,Wi) = the size of the convolution's input; wt packed with
 * HG_CONV_PACK_DGRAD.   gin[b,n] = oscale[b,n] * sum_k dgrad(iscale[b,k] * gout[b,k]).
 * Stride 2 runs as four launches (one per parity class of the output pixel). */
int hg_conv2d_dgrad(const float *gout, const float *wt, float *gin, const float *iscale, const float *oscale,
                    int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradient: gw[n,k,dy,dx] = sum_{b,y,x} gscale[b,n]*gout[b,n,y,x] * iscale[b,k]*in[b,k,y*stride+dy-p,x*stride+dx-p]
 *   in (B,K,Hi,Wi), gout (B,N,Ho,Wo), gw (N,K,ksize,ksize) contiguous, fully written (deterministic: split-K
 *   slabs in the workspace are summed in a fixed order, no atomics).  iscale / gscale may be NULL. */
size_t hg_conv2d_wgrad_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize,
                                       int32_t stride);
int hg_conv2d_wgrad(const float *in, const float *gout, float *gw, const float *iscale, const float *gscale,
                    int32_t B, int32_t K, int32_t N, int32_t Hi, int32_t Wi, int32_t ksize, int32_t stride,
                    void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_CONV_H */
