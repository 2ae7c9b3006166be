/* hg_hist.h -- C ABI of the MI355X-native RGB-uv histogram (libhistogan_hip.so).
 *
 * Drop-in boundary for histogram_classes/RGBuvHistBlock.py:75-228 of the
 * reference (RGBuvHistBlock.forward) and for the autograd replay of that chain
 * (SURVEY.md section 8a, rows a2-a7).  The reference has no native code; these
 * entry points are what a ctypes binding inside RGBuvHistBlock.forward calls
 * (see INTEGRATION.md).
 *
 * Conventions: every function returns 0 on success, a negative HG_E* code on an
 * argument error, or a positive hipError_t.  Functions never allocate, free or
 * synchronise; all device work is enqueued on `stream` (a hipStream_t passed as
 * void*; NULL = the legacy default stream).  All pointers except the params
 * struct are DEVICE pointers.  fp32 everywhere.  Re-entrant.
 */
#ifndef HG_HIST_H
#define HG_HIST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_OK 0
#define HG_EINVAL (-1)    /* bad argument (NULL pointer, non-positive size, ...)        */
#define HG_EMETHOD (-2)   /* unknown kernel method  (RGBuvHistBlock.py:141-144)          */
#define HG_ERESIZE (-3)   /* unknown resize mode    (RGBuvHistBlock.py:90-93)            */
#define HG_EWORKSPACE (-4)/* workspace too small                                         */
#define HG_EUNSUPPORTED (-5)

/* soft-bin kernel, RGBuvHistBlock.py:124-140 */
#define HG_METHOD_THRESHOLDING 0
#define HG_METHOD_RBF 1
#define HG_METHOD_INVERSE_QUADRATIC 2

#define HG_PROJ_RGBUV 0
#define HG_PROJ_RGCHROMA 1
#define HG_PROJ_DIRECT 2

/* stage-0 resize, RGBuvHistBlock.py:77-95 */
#define HG_RESIZE_NONE 0      /* H<=insz and W<=insz: pixels used as they are            */
#define HG_RESIZE_BILINEAR 1  /* F.interpolate(size=(Hs,Ws), bilinear, align_corners=False) */
#define HG_RESIZE_SAMPLING 2  /* x.index_select(2,row_idx).index_select(3,col_idx)        */

typedef struct hg_hist_params {
  /* sizeof(hg_hist_params) as the CALLER compiled it (ABI guard, since version 102): every entry point returns
   * HG_EINVAL when it differs from the library's, so a caller built against an older header -- or one that did not
   * zero the struct and fill this in -- is rejected instead of having trailing fields (proj_cache: a pointer the
   * forward WRITES through) read as garbage. */
  uint32_t struct_size;
  /* input image batch x: (B, C>=3, H, W), element strides (any layout) */
  int32_t B, C, H, W;
  int64_t stride_b, stride_c, stride_h, stride_w;
  /* pixels entering the histogram: Hs x Ws (== H x W for HG_RESIZE_NONE) */
  int32_t Hs, Ws;
  int32_t resize_mode;
  const int32_t *row_idx; /* device, Hs entries, HG_RESIZE_SAMPLING only */
  const int32_t *col_idx; /* device, Ws entries, HG_RESIZE_SAMPLING only */
  /* histogram definition (ctor args of RGBuvHistBlock, RGBuvHistBlock.py:29-73) */
  int32_t h;              /* bins per axis                                               */
  double lo, hi;          /* sorted hist_boundary                                        */
  int32_t method;         /* HG_METHOD_*                                                 */
  double sigma;           /* RBF / inverse-quadratic width (ignored for thresholding)    */
  int32_t intensity_scale;
  int32_t green_only;     /* only plane 1 (log g/r, log g/b), written at plane index 0   */
  /* 2-D projection of a pixel (HG_PROJ_*).  0: RGB-uv log-chroma, 3 planes.  The two one-plane variants of the
   * reference's other histogram blocks share everything else (clamp, resize, kernels, normalisation):
   * 1: rg-chroma (u,v) = (R,G)/(R+G+B+1e-6), weight Iy   (histogram_classes/rgChromaHistBlock.py:100-141)
   * 2: direct     (u,v) = channels (1,2), weight = channel 0  (histogram_classes/LabHistBlock.py:102-140) */
  int32_t projection;
  /* 1: the F.relu the train step puts in front of the block (histoGAN/histoGAN.py:955) is part of the call -- identical
   * forward (clamp(relu(x)) == clamp(x)); the backward masks x <= 0 instead of x < 0.  Saves that aten launch and node. */
  int32_t pre_relu;
  /* Optional device buffer of B * Hs * Ws * 32 bytes (16-byte aligned), or NULL.  The forward stores every pixel's
   * projection (log-chroma differences, weight, clamped / resized colour) there; a backward call given the SAME
   * buffer reads it instead of re-sampling and re-projecting (bilinear taps + three fp64 logarithms per pixel).
   * Smooth kernels (inverse-quadratic, dense RBF) only; the scatter paths ignore it. */
  void *proj_cache;
} hg_hist_params;

/* library / build identification */
int hg_version(void);
const char *hg_error_string(int code);

/* Self-test of the scatter path's fast window classification (method = thresholding): exhaustive error of the
 * hardware logarithm it uses over every float in [1e-6, 1 + 2e-6], against the fp64 logarithm rounded to fp32 that
 * the exact path (and the reference's CPU logf, RGBuvHistBlock.py:112-114) evaluates.  out2 (device, 2 floats):
 * [0] = max relative error where |ln x| >= 1e-3, [1] = max absolute error elsewhere.  The classification's margins
 * assume [0] <= 3e-7 and [1] <= 3e-7 (hg_hist.hip: kFastLogRel, kFastAbs carry the roundings on top). */
int hg_selftest_fastlog(float *out2, void *stream);

/* 1 when a forward / backward pair with these params uses `proj_cache` (the dense MFMA kernels), 0 when it would be
 * ignored (thresholding, the truncated-RBF scatter / gather pair): lets the caller skip the 32 B / pixel allocation.
 * Negative HG_E* on bad params. */
int hg_rgbuv_hist_uses_proj_cache(const hg_hist_params *p);

/* Bytes of scratch each call needs for these params (both may be queried at once). */
int hg_rgbuv_hist_workspace_bytes(const hg_hist_params *p, size_t *fwd_bytes, size_t *bwd_bytes);

/* Forward: hist_out (B, P, h, h) contiguous, P = green_only ? 1 : 3, L1-normalised per
 * image as RGBuvHistBlock.py:224-228;  sum_out (B) = sum of the raw histogram + 1e-6
 * (the normaliser, needed by the backward). */
int hg_rgbuv_hist_fwd(const hg_hist_params *p, const float *x, float *hist_out, float *sum_out,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Backward: grad_x (B, C, H, W) contiguous, fully written (channels >= 3 get 0).
 * grad_out (B, P, h, h) contiguous; hist_out / sum_out as produced by the forward. */
int hg_rgbuv_hist_bwd(const hg_hist_params *p, const float *x, const float *grad_out,
                      const float *hist_out, const float *sum_out, float *grad_x,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Hellinger histogram loss, histoGAN/histoGAN.py:957-960:
 *   loss = alpha / sqrt(2) * sqrt( sum_{b,p,i,j} (sqrt(t) - sqrt(g))^2 ) / B
 * n = B*P*h*h elements.  loss_out (1); grad_gen (n) may be NULL (forward only).
 * grad_gen = d loss / d gen (not yet multiplied by an upstream gradient).
 * workspace: hg_hellinger_workspace_bytes(n). */
size_t hg_hellinger_workspace_bytes(int64_t n);
int hg_hellinger_fwd_bwd(const float *target, const float *gen, int64_t n, int32_t batch,
                         float alpha, float *loss_out, float *grad_gen, void *workspace,
                         size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_HIST_H */
