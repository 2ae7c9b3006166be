/* hg_nets.h -- C ABI of the generator/optimizer elementwise kernels (libhistogan_hip.so).
 *
 * These replace the aten op-chains around the dense contraction of the reference's modulated
 * convolution and its optimizer:
 *   Conv2DMod.forward            histoGAN/histoGAN.py:420-440   (activation-modulation form:
 *                                conv(x*(s+1), W) * d  ==  grouped conv with per-sample weights)
 *   nn.Upsample(bilinear x2)     histoGAN/histoGAN.py:377-378, 447-448, 462-463
 *   noise add + LeakyReLU(0.2)   histoGAN/histoGAN.py:465-476   (noise permute (0,3,2,1): H<->W)
 *   DiffGrad.step                torch_optimizer (third-party), ctor at histoGAN/histoGAN.py:670-671
 *   HistoGAN.EMA                 histoGAN/histoGAN.py:698-707
 *
 * Conventions as in hg_hist.h: return 0 / negative HG_E* / positive hipError_t; device pointers;
 * fp32; contiguous NCHW; enqueue on `stream`; never allocate or synchronise.
 */
#ifndef HG_NETS_H
#define HG_NETS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[b,c,:,:] = up(x[b,c,:,:]) * (s[b,c] + 1);  up = identity (upsample=0) or bilinear x2
 * (align_corners=False, edge clamp).  x: (B,C,H,W); out: (B,C,H*f,W*f), f = upsample ? 2 : 1.
 * s may be NULL (no modulation: plain upsample). */
int hg_modulate_fwd(const float *x, const float *s, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t upsample, void *stream);
/* gx = up^T(gout * (s+1));  gs[b,c] = sum(gout * up(x))  (gs may be NULL when s is NULL). */
int hg_modulate_bwd(const float *gout, const float *x, const float *s, float *gx, float *gs, int32_t B,
                    int32_t C, int32_t H, int32_t W, int32_t upsample, void *workspace, size_t workspace_bytes,
                    void *stream);

/* The to-RGB path of a generator block as ONE stream over x (RGBBlock.forward, histoGAN/histoGAN.py:380-390: the 1x1
 * modulated convolution without demodulation onto C = 3 (4: rgba) channels, plus the previous block's RGB image):
 *   out[b,c,p] = sum_o w[c,o] * (s[b,o] + 1) * x[b,o,p] + prev[b,c,p]
 * x (B,O,HW), s (B,O) or NULL, w (C,O), prev (B,C,HW) or NULL, out (B,C,HW); HW % 4 == 0, C <= 4, C*O*4 B of LDS
 * (HG_EUNSUPPORTED otherwise: the caller takes the modulated-convolution path). */
int hg_torgb_fwd(const float *x, const float *s, const float *w, const float *prev, float *out, int32_t B, int32_t O,
                 int32_t C, int32_t HW, void *stream);
/* Its adjoint from ONE pass over x and the C-channel gradient g (B,C,HW):
 *   gx[b,o,p] = (s[b,o] + 1) * sum_c w[c,o] g[b,c,p]
 *   gs[b,o]   = sum_p x[b,o,p] * sum_c w[c,o] g[b,c,p]          (NULL iff s is NULL)
 *   gw[c,o]   = sum_{b,p} g[b,c,p] * (s[b,o] + 1) * x[b,o,p]
 * (the gradient of prev is g itself).  Deterministic: per-block partial sums in the workspace, combined in fixed order. */
size_t hg_torgb_bwd_workspace_bytes(int32_t B, int32_t O, int32_t C, int32_t HW);
int hg_torgb_bwd(const float *g, const float *x, const float *s, const float *w, float *gx, float *gs, float *gw, int32_t B,
                 int32_t O, int32_t C, int32_t HW, void *workspace, size_t workspace_bytes, void *stream);

/* ---- everything between two convolutions of the generator's backward as ONE pass (hg_gstage.hip) ----------------------
 * A stage output  out = lrelu_0.2(d conv + wn nz + bn)  (GeneratorBlock.forward, histoGAN/histoGAN.py:461-479) with its
 * consumers: A = the next modulated convolution (same resolution, up = 0: Conv2DMod :420-424; or behind the bilinear x2 of
 * the next block, up = 1: :447-448, 463-464) and R = the block's to-RGB convolution (RGBBlock :380-390).  Replaces the chain
 * hg_modulate_bwd + hg_torgb_bwd + add + hg_demod_noise_lrelu_bwd (conv = NULL form) with one read of each operand:
 *   t = up ? up2^T(ga) : ga;  tr = sum_k w_rgb[k,c] g_rgb[b,k,p];  G = t (sa + 1) + tr (s_rgb + 1);  m = G (out > 0 ? 1 : 0.2)
 *   gconv = m d[b,c]                                  (B,C,H,H): the upstream gradient of the stage's convolution
 *   gs_a[b,c]  = sum_p out t                          (NULL iff sa is NULL)          style gradient of A (modulation part)
 *   gs_rgb[b,c] = sum_p out tr,  gw_rgb[k,c] = sum_{b,p} g_rgb[b,k,p] (s_rgb[b,c] + 1) out[b,c,p]
 *   gd[b,c] = sum_p m conv  (conv recovered from out; NULL iff d is NULL),  gwn[c] = sum_{b,p} m nz,  gbn[c] = sum_{b,p} m
 * out (B,C,H,H); ga (B,C,H,H) or (B,C,2H,2H) or NULL; g_rgb (B,Cr,H,H) or NULL (not both NULL), Cr <= 4; nzt (B,S,S) as in
 * hg_demod_noise_lrelu_fwd.  H % 4 == 0, S % 4 == 0 (HG_EUNSUPPORTED otherwise).  Deterministic (partial sums in the
 * workspace, combined in fixed order). */
size_t hg_gstage_bwd_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t up);
int hg_gstage_bwd(const float *out, const float *ga, const float *sa, int32_t up, const float *g_rgb, const float *w_rgb,
                  const float *s_rgb, int32_t Cr, const float *d, const float *nzt, const float *wn, const float *bn, int32_t S,
                  float *gconv, float *gs_a, float *gs_rgb, float *gw_rgb, float *gd, float *gwn, float *gbn, int32_t B,
                  int32_t C, int32_t H, void *workspace, size_t workspace_bytes, void *stream);

/* Scratch (partial sums, combined in fixed order: deterministic) for hg_modulate_bwd / hg_demod_noise_lrelu_bwd /
 * hg_channel_sum on a (B, C, H, W) tensor. */
size_t hg_nets_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);

/* out[b,o,i,j] = lrelu_0.2( conv[b,o,i,j] * d[b,o] + wn[o] * nzt[b,i,j] + bn[o] )
 * conv/out: (B,O,H,H);  d: (B,O) or NULL (no demodulation);  wn, bn: (O) = to_noise Linear(1,O);
 * nzt: (B,S,S), S >= H, the noise image ALREADY TRANSPOSED (nzt[b][i][j] = inoise[b][j][i][0]):
 * the reference's `.permute((0,3,2,1))` swaps H and W, so position (i,j) sees inoise[b,j,i]. */
int hg_demod_noise_lrelu_fwd(const float *conv, const float *d, const float *nzt, const float *wn,
                             const float *bn, float *out, int32_t B, int32_t O, int32_t H, int32_t S,
                             void *stream);
/* m = gout * (out > 0 ? 1 : 0.2):  gconv = m * d;  gd[b,o] = sum m*conv  (gd may be NULL when d is);
 * gwn_part[b,o] = sum m * nzt[b,i,j];  gbn_part[b,o] = sum m   (caller sums the parts over b).
 * conv may be NULL (the fused forward hg_modconv2d_fwd never stores it): then wn, bn (O) are needed, conv*d is
 * recovered from out (pre = out > 0 ? out : 5 out, minus the noise term) and the sum of m*conv*d is divided by d[b,o]
 * before it is stored: gd has the same meaning either way. */
int hg_demod_noise_lrelu_bwd(const float *gout, const float *out, const float *conv, const float *d,
                             const float *nzt, const float *wn, const float *bn, float *gconv, float *gd,
                             float *gwn_part, float *gbn_part, int32_t B, int32_t O, int32_t H, int32_t S,
                             void *workspace, size_t workspace_bytes, void *stream);

/* out[c] = sum_{b,p} g[b,c,p]: the bias gradient of a convolution (nn.Conv2d bias, histoGAN/histoGAN.py:510-518). */
int hg_channel_sum(const float *g, float *out, int32_t B, int32_t C, int32_t HW, void *workspace,
                   size_t workspace_bytes, void *stream);

/* gm = g * (out > 0 ? 1 : slope): the gradient through `nn.Conv2d -> LeakyReLU(slope)` of a discriminator block
 * (histoGAN/histoGAN.py:510-515), masked on the activation's OUTPUT (== aten leaky_relu_backward(g, out, slope, True)),
 * and csum[c] = sum_{b,p} gm[b,c,p], the bias gradient of that convolution, from the same pass (csum may be NULL).
 * g, out, gm (B, C, HW) contiguous; workspace as hg_channel_sum. */
int hg_lrelu_bwd_channel_sum(const float *g, const float *out, float slope, float *gm, float *csum, int32_t B, int32_t C,
                             int32_t HW, void *workspace, size_t workspace_bytes, void *stream);

/* Demodulation backward, style side (autograd's backward of the Conv2DMod demodulation coefficient with respect to the
 * style, histoGAN/histoGAN.py:427-429):  gy[b,i] = 2 s1[b,i] sum_o gq[b,o] wsq[o,i],  gq = gd * (-0.5) * d^3,
 * wsq[o,i] = sum_t w[o,i,t]^2 (N, K);  gd, d (B, N);  s1 = style + 1, gy (B, K).  Deterministic (fixed-order partial sums). */
size_t hg_demod_style_grad_workspace_bytes(int32_t B, int32_t N, int32_t K);
int hg_demod_style_grad(const float *gd, const float *d, const float *s1, const float *wsq, float *gy, int32_t B, int32_t N,
                        int32_t K, void *workspace, size_t workspace_bytes, void *stream);

/* Demodulation backward, weight side (Conv2DMod, histoGAN/histoGAN.py:427-429 on the shared weight; autograd's backward
 * of `d = rsqrt(sum((w * (s+1))^2) + eps)` with respect to w):
 *   gw[o,i,t] (+)= 2 w[o,i,t] sum_b gq[b,o] s1[b,i]^2,   gq = gd * (-0.5) * d^3,   s1 = style + 1
 * w, gw (N, K, taps) contiguous; gd, d (B, N); s1 (B, K).  accumulate != 0: added to gw, else gw is overwritten. */
int hg_demod_weight_term(const float *w, const float *gd, const float *d, const float *s1, float *gw, int32_t B,
                         int32_t N, int32_t K, int32_t taps, int32_t accumulate, void *stream);

/* Fused multi-tensor DiffGrad step over one flat parameter buffer of n floats:
 *   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  dfc = 1/(1+exp(-|g_prev-g|));  g_prev = g
 *   p -= lr*sqrt(1-b2^t)/(1-b1^t) * (m*dfc) / (sqrt(v)+eps)                         (t = step >= 1) */
int hg_diffgrad_step(float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *prev_grad,
                     int64_t n, float lr, float beta1, float beta2, float eps, int32_t step, void *stream);

/* The same step with the bias-corrected step size lr*sqrt(1-b2^t)/(1-b1^t) read from device memory (one float):
 * no per-step host value in the launch, so a captured hipGraph of the train step can replay it; the host writes
 * hg_diffgrad_step_size(lr, b1, b2, t) there before each replay. */
float hg_diffgrad_step_size(float lr, float beta1, float beta2, int32_t step);
int hg_diffgrad_step_dev(float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *prev_grad, int64_t n,
                         const float *step_size_dev, float beta1, float beta2, float eps, void *stream);

/* ma = beta*ma + (1-beta)*p over a flat buffer (HistoGAN.EMA). */
int hg_ema_update(float *ma, const float *p, int64_t n, float beta, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_NETS_H */
