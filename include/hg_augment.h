/* hg_augment.h -- C ABI of the DiffAugment kernels (libhistogan_hip.so); SURVEY.md section 8, row f-4.
 *
 * The reference augments every image batch entering the discriminator when --aug_prob > 0
 * (AugWrapper, histoGAN/histoGAN.py:312-331 -> utils/diff_augment.py:9-107): a chain of aten index/gather ops per
 * augmentation.  Here one launch applies a whole run of spatial augmentations and one launch the colour ones, each
 * parameterised PER SAMPLE by a small device table; every augmentation is linear in the image, so the adjoint entry
 * points below are their exact backward (and the backward of the backward is the forward again: the gradient
 * penalty differentiates twice through the augmented real images).
 *
 * Conventions as in hg_hist.h: return 0 / negative HG_E* / positive hipError_t; device pointers; fp32; contiguous NCHW;
 * enqueue on `stream`; never allocate or synchronise.
 */
#ifndef HG_AUGMENT_H
#define HG_AUGMENT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-sample spatial parameters: HG_AUG_NP int32 each, applied in this order (identity values in brackets):
 *   [0] flip      horizontal flip, torch.flip(dims=(3,))                      random_hflip, histoGAN.py:312-315   [0]
 *   [1] roll_h    torch.roll(img, roll_h, dim H)                              rand_offset, diff_augment.py:52-70  [0]
 *   [2] roll_w    torch.roll(img, roll_w, dim W)                                                                  [0]
 *   [3] shift_h   out[i,j] = in[i+shift_h, j+shift_w], zero outside           rand_translation, :33-50            [0]
 *   [4] shift_w                                                                                                   [0]
 *   [5..8] r0, r1, c0, c1   rows r0..r1 x cols c0..c1 (inclusive) set to zero  rand_cutout, :78-97   [r0 > r1: none] */
#define HG_AUG_NP 9

/* out[b,c,:,:] = cutout(shift(roll(flip(x[b,c,:,:]))))   with params[b*HG_AUG_NP + ...];   x, out: (B,C,H,W).
 * adjoint != 0 applies the transpose of that map (the backward of the call with adjoint == 0). */
int hg_augment_spatial(const float *x, const int32_t *params, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                       int32_t adjoint, void *stream);

/* mean[b] = mean over (C,H,W) of x[b]   (rand_contrast's per-image mean, diff_augment.py:27-31).
 * workspace: hg_augment_workspace_bytes(B) bytes. */
size_t hg_augment_workspace_bytes(int32_t B);
int hg_sample_mean(const float *x, float *mean, int32_t B, int64_t CHW, void *workspace, size_t workspace_bytes,
                   void *stream);

/* Colour augmentations, per sample b with color[b*3 + {0,1,2}] = {brightness offset, saturation factor, contrast factor}
 * (identity: 0, 1, 1), in the reference's order brightness -> saturation -> contrast (diff_augment.py:16-31):
 *   x1 = x + br;   x2 = (x1 - mean_c(x1)) * sat + mean_c(x1);   out = (x2 - m) * con + m,   m = mean[b] + br
 * (the saturation step keeps the per-pixel channel mean, so m is the image mean of x2 as well).
 * `mean` = hg_sample_mean(x).  adjoint != 0: the transpose of the LINEAR part (backward): mean = hg_sample_mean(g),
 *   g2 = con * g + (1 - con) * mean[b];   out = sat * g2 + (1 - sat) * mean_c(g2). */
int hg_augment_color(const float *x, const float *mean, const float *color, float *out, int32_t B, int32_t C,
                     int32_t HW, int32_t adjoint, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_AUGMENT_H */
