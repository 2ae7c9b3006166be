/* hg_wino.h -- C ABI of the Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions (libhistogan_hip.so).
 *
 * The contraction inside Conv2DMod.forward (histoGAN/histoGAN.py:431-439), the 3x3 convolutions of
 * DiscriminatorBlock.net (:510-515) and their data gradients are 3x3, stride 1, zero padding 1.  For those the
 * minimal-filtering form  Y = A^T [ (G g G^T) . (B^T d B) ] A  needs 16 multiplications per 2x2 output tile and input
 * channel instead of 36: the sixteen transform positions are sixteen independent (N x K) x (K x tiles) contractions on
 * v_mfma_f32_32x32x2_f32 -- still exact fp32 products with fp32 accumulation, 2.25x fewer of them.  Measured against
 * fp64 the result is CLOSER than the direct fma chain (sixteen chains of depth K instead of one of depth 9 K; transforms
 * are additions and halvings only): 4e-7 ... 1.6e-6 max-norm relative where the direct kernel sits at 1.1e-6 ... 3.0e-6.
 *
 * Conventions as hg_conv.h: 0 / negative HG_E* / positive hipError_t; device pointers; the caller's stream; nothing is
 * allocated or synchronised.  HG_EUNSUPPORTED = this shape is not served (odd H / W, K % 8, too few tiles): the caller
 * uses hg_conv2d_fwd / hg_conv2d_dgrad.
 */
#ifndef HG_WINO_H
#define HG_WINO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1 when hg_wino_conv2d serves (B, K, N, H, W) -- K input channels, N output channels of the launch (for a data
 * gradient K = the convolution's output channels) -- AND is expected to beat the direct kernel there. */
int hg_wino_supported(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W);

/* Transformed-weight operand U = G g G^T in the kernel's lane order.  mode as hg_conv_pack_weights (HG_CONV_PACK_FWD:
 * K = Ci, N = Co; HG_CONV_PACK_DGRAD: K = Co, N = Ci, taps flipped).  Returns floats (0: weight shape not served). */
size_t hg_wino_packed_elems(int32_t Co, int32_t Ci, int32_t mode);
int hg_wino_pack_weights(const float *w, float *u, int32_t Co, int32_t Ci, int32_t mode, void *stream);

/* Every 3x3 weight of a model in ONE launch (after an optimizer step), as hg_conv_pack_weights_multi: `items_dev` =
 * device array of n_items descriptors; u_fwd / u_dgrad (hg_wino_packed_elems floats each) may be NULL (that operand is
 * not written); block_begin = running sum of hg_wino_pack_blocks(Co, Ci, u_fwd != NULL, u_dgrad != NULL) over the
 * preceding items, total_blocks = the sum over all. */
typedef struct hg_wino_pack_item {
  const float *w;
  float *u_fwd, *u_dgrad;
  /* optional (NULL: none; needs u_fwd): wsq[co][ci] = sum_taps W[co][ci][t]^2, as hg_pack_item.wsq of hg_conv.h (the weight
   * factor of the demodulation coefficient) -- so that a weight whose direct operands nobody asks for needs no direct pack */
  float *wsq;
  int32_t Co, Ci, block_begin, reserved;
} hg_wino_pack_item;
int32_t hg_wino_pack_blocks(int32_t Co, int32_t Ci, int32_t want_fwd, int32_t want_dgrad);
int hg_wino_pack_weights_multi(const hg_wino_pack_item *items_dev, int32_t n_items, int32_t total_blocks, void *stream);

/* Scratch of hg_wino_conv2d for these arguments (K-split slabs of launches that cannot fill the chip with output
 * tiles); 0 = none. */
size_t hg_wino_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W);

/* out[b,n,y,x] = lrelu_slope( oscale[b,n] * sum_{k,dy,dx} iscale[b,k] in[b,k,y+dy-1,x+dx-1] W[n,k,dy,dx]
 *                             + bias[n] + noise_w[n] * noise_img[b,y,x] ) + addend[b,n,y,x]
 * (same epilogue as hg_modconv2d_fwd / hg_conv2d_fwd_add; every optional pointer may be NULL, lrelu_slope 0 = none,
 * addend only without scales / noise / activation); u from hg_wino_pack_weights. */
int hg_wino_conv2d(const float *in, const float *u, float *out, const float *iscale, const float *oscale,
                   const float *bias, const float *noise_w, const float *noise_img, int32_t noise_S, float lrelu_slope,
                   const float *addend, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Weight gradient of that convolution on the same transform:  gw = G^T [ sum_tiles (A gout A^T) . (B^T in B) ] G
 *   in (B,K,H,W), gout (B,N,H,W), gw (N,K,3,3) contiguous, fully written; deterministic (per-split slabs in the
 *   workspace, summed in fixed order).  Maps whose sides are 2 x a power of two.  hg_wino_wgrad_supported: served AND
 *   expected to beat hg_conv2d_wgrad (>= 64 channels on both sides, maps >= 4x4). */
int hg_wino_wgrad_supported(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W);
size_t hg_wino_wgrad_workspace_bytes(int32_t B, int32_t K, int32_t N, int32_t H, int32_t W);
int hg_wino_wgrad(const float *in, const float *gout, float *gw, int32_t B, int32_t K, int32_t N, int32_t H, int32_t W,
                  void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_WINO_H */
