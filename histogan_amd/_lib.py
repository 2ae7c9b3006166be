"""ctypes binding of libhistogan_hip.so (the C ABI declared in include/hg_hist.h).

There is deliberately NO fallback: if the library is missing or fails to load,
importing this module raises, and every op built on it is unusable.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_TAG = os.environ.get('HG_LIB_TAG', '')  # experiment builds only (histogan_amd/build.py)
LIB_PATH = os.path.join(_PKG, 'libhistogan_hip' + ('_' + _TAG if _TAG else '') + '.so')

HG_METHOD = {'thresholding': 0, 'RBF': 1, 'inverse-quadratic': 2}
HG_RESIZE_NONE, HG_RESIZE_BILINEAR, HG_RESIZE_SAMPLING = 0, 1, 2
HG_PROJ = {'rgbuv': 0, 'rgchroma': 1, 'direct': 2}


class HgHistParams(ctypes.Structure):
    """struct hg_hist_params (include/hg_hist.h)."""
    _fields_ = [
        ('struct_size', ctypes.c_uint32),
        ('B', ctypes.c_int32), ('C', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
        ('stride_b', ctypes.c_int64), ('stride_c', ctypes.c_int64),
        ('stride_h', ctypes.c_int64), ('stride_w', ctypes.c_int64),
        ('Hs', ctypes.c_int32), ('Ws', ctypes.c_int32),
        ('resize_mode', ctypes.c_int32),
        ('row_idx', ctypes.c_void_p), ('col_idx', ctypes.c_void_p),
        ('h', ctypes.c_int32),
        ('lo', ctypes.c_double), ('hi', ctypes.c_double),
        ('method', ctypes.c_int32),
        ('sigma', ctypes.c_double),
        ('intensity_scale', ctypes.c_int32),
        ('green_only', ctypes.c_int32),
        ('projection', ctypes.c_int32),
        ('pre_relu', ctypes.c_int32),
        ('proj_cache', ctypes.c_void_p),
    ]


class GlinLayer(ctypes.Structure):
    """struct hg_glin_layer (include/hg_linear.h)."""
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('b', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('gw', ctypes.c_void_p), ('gb', ctypes.c_void_p), ('N', ctypes.c_int32), ('group', ctypes.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: the gfx950 HIP library has not been built. '
            f'Run `python -m histogan_amd.build` (needs hipcc). There is no CPU/PyTorch fallback.')
    # torch must be imported first so that libamdhip64.so.7 resolves to the runtime torch uses
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, sz, i32, i64, f32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    PP = ctypes.POINTER(HgHistParams)
    lib.hg_version.restype = ctypes.c_int
    lib.hg_version.argtypes = []
    lib.hg_error_string.restype = ctypes.c_char_p
    lib.hg_error_string.argtypes = [ctypes.c_int]
    lib.hg_rgbuv_hist_workspace_bytes.restype = ctypes.c_int
    lib.hg_rgbuv_hist_workspace_bytes.argtypes = [PP, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.hg_rgbuv_hist_uses_proj_cache.restype = ctypes.c_int
    lib.hg_rgbuv_hist_uses_proj_cache.argtypes = [PP]
    lib.hg_rgbuv_hist_fwd.restype = ctypes.c_int
    lib.hg_rgbuv_hist_fwd.argtypes = [PP, vp, vp, vp, vp, sz, vp]
    lib.hg_rgbuv_hist_bwd.restype = ctypes.c_int
    lib.hg_rgbuv_hist_bwd.argtypes = [PP, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.hg_hellinger_workspace_bytes.restype = sz
    lib.hg_hellinger_workspace_bytes.argtypes = [i64]
    lib.hg_hellinger_fwd_bwd.restype = ctypes.c_int
    lib.hg_hellinger_fwd_bwd.argtypes = [vp, vp, i64, i32, f32, vp, vp, vp, sz, vp]
    lib.hg_selftest_fastlog.restype = ctypes.c_int
    lib.hg_selftest_fastlog.argtypes = [vp, vp]
    # include/hg_nets.h
    lib.hg_modulate_fwd.restype = ctypes.c_int
    lib.hg_modulate_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.hg_modulate_bwd.restype = ctypes.c_int
    lib.hg_modulate_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_nets_workspace_bytes.restype = sz
    lib.hg_nets_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.hg_torgb_fwd.restype = ctypes.c_int
    lib.hg_torgb_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.hg_torgb_bwd_workspace_bytes.restype = sz
    lib.hg_torgb_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.hg_torgb_bwd.restype = ctypes.c_int
    lib.hg_torgb_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_gstage_bwd_workspace_bytes.restype = sz
    lib.hg_gstage_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.hg_gstage_bwd.restype = ctypes.c_int
    lib.hg_gstage_bwd.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                  vp, sz, vp]
    lib.hg_channel_sum.restype = ctypes.c_int
    lib.hg_channel_sum.argtypes = [vp, vp, i32, i32, i32, vp, sz, vp]
    lib.hg_demod_noise_lrelu_fwd.restype = ctypes.c_int
    lib.hg_demod_noise_lrelu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.hg_demod_noise_lrelu_bwd.restype = ctypes.c_int
    lib.hg_demod_noise_lrelu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_lrelu_bwd_channel_sum.restype = ctypes.c_int
    lib.hg_lrelu_bwd_channel_sum.argtypes = [vp, vp, f32, vp, vp, i32, i32, i32, vp, sz, vp]
    lib.hg_demod_style_grad_workspace_bytes.restype = ctypes.c_size_t
    lib.hg_demod_style_grad_workspace_bytes.argtypes = [i32, i32, i32]
    lib.hg_demod_style_grad.restype = ctypes.c_int
    lib.hg_demod_style_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]
    lib.hg_demod_weight_term.restype = ctypes.c_int
    lib.hg_demod_weight_term.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.hg_diffgrad_step.restype = ctypes.c_int
    lib.hg_diffgrad_step.argtypes = [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]
    lib.hg_diffgrad_step_size.restype = f32
    lib.hg_diffgrad_step_size.argtypes = [f32, f32, f32, i32]
    lib.hg_diffgrad_step_dev.restype = ctypes.c_int
    lib.hg_diffgrad_step_dev.argtypes = [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, vp]
    lib.hg_ema_update.restype = ctypes.c_int
    lib.hg_ema_update.argtypes = [vp, vp, i64, f32, vp]
    # include/hg_conv.h
    lib.hg_conv_packed_elems.restype = sz
    lib.hg_conv_packed_elems.argtypes = [i32, i32, i32, i32]
    lib.hg_conv_pack_weights.restype = ctypes.c_int
    lib.hg_conv_pack_weights.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.hg_conv_pack_weights_both.restype = ctypes.c_int
    lib.hg_conv_pack_weights_both.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.hg_conv2d_fwd.restype = ctypes.c_int
    lib.hg_conv2d_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_conv2d_fwd_add.restype = ctypes.c_int
    lib.hg_conv2d_fwd_add.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_modconv2d_fwd.restype = ctypes.c_int
    lib.hg_modconv2d_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_conv2d_workspace_bytes.restype = sz
    lib.hg_conv2d_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32]
    lib.hg_conv_pack_blocks.restype = ctypes.c_int32
    lib.hg_conv_pack_blocks.argtypes = [i32, i32]
    lib.hg_conv_pack_weights_multi.restype = ctypes.c_int
    lib.hg_conv_pack_weights_multi.argtypes = [vp, i32, i32, vp]
    lib.hg_conv2d_plan.restype = ctypes.c_int
    lib.hg_conv2d_plan.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int32)]
    lib.hg_conv2d_dgrad.restype = ctypes.c_int
    lib.hg_conv2d_dgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_conv2d_wgrad_workspace_bytes.restype = sz
    lib.hg_conv2d_wgrad_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32]
    lib.hg_conv2d_wgrad.restype = ctypes.c_int
    lib.hg_conv2d_wgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]
    # include/hg_recolor.h
    lib.hg_instnorm_workspace_bytes.restype = sz
    lib.hg_instnorm_workspace_bytes.argtypes = [i64]
    lib.hg_instnorm_lrelu_fwd.restype = ctypes.c_int
    lib.hg_instnorm_lrelu_fwd.argtypes = [vp, vp, vp, i64, i32, f32, f32, vp, sz, vp]
    lib.hg_instnorm_lrelu_bwd.restype = ctypes.c_int
    lib.hg_instnorm_lrelu_bwd.argtypes = [vp, vp, vp, vp, i64, i32, f32, vp, sz, vp]
    lib.hg_stencil3.restype = ctypes.c_int
    lib.hg_stencil3.argtypes = [vp, vp, ctypes.POINTER(f32), i32, i32, i32, i32, i32, vp]
    lib.hg_depthwise_valid.restype = ctypes.c_int
    lib.hg_depthwise_valid.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp]
    # include/hg_linear.h
    GL = ctypes.POINTER(GlinLayer)
    lib.hg_grouped_linear_fwd.restype = ctypes.c_int
    lib.hg_grouped_linear_fwd.argtypes = [GL, i32, i32, i32, vp]
    lib.hg_grouped_linear_bwd_input_workspace_bytes.restype = sz
    lib.hg_grouped_linear_bwd_input_workspace_bytes.argtypes = [GL, i32, i32, i32]
    lib.hg_grouped_linear_bwd_input.restype = ctypes.c_int
    lib.hg_grouped_linear_bwd_input.argtypes = [GL, i32, ctypes.POINTER(vp), i32, i32, i32, vp, sz, vp]
    lib.hg_grouped_linear_bwd_params.restype = ctypes.c_int
    lib.hg_grouped_linear_bwd_params.argtypes = [GL, i32, i32, i32, vp]
    # include/hg_wino.h
    lib.hg_wino_supported.restype = ctypes.c_int
    lib.hg_wino_supported.argtypes = [i32, i32, i32, i32, i32]
    lib.hg_wino_packed_elems.restype = sz
    lib.hg_wino_packed_elems.argtypes = [i32, i32, i32]
    lib.hg_wino_pack_weights.restype = ctypes.c_int
    lib.hg_wino_pack_weights.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.hg_wino_pack_blocks.restype = ctypes.c_int32
    lib.hg_wino_pack_blocks.argtypes = [i32, i32, i32, i32]
    lib.hg_wino_pack_weights_multi.restype = ctypes.c_int
    lib.hg_wino_pack_weights_multi.argtypes = [vp, i32, i32, vp]
    lib.hg_wino_workspace_bytes.restype = sz
    lib.hg_wino_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.hg_wino_conv2d.restype = ctypes.c_int
    lib.hg_wino_conv2d.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.hg_wino_wgrad_supported.restype = ctypes.c_int
    lib.hg_wino_wgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    lib.hg_wino_wgrad_workspace_bytes.restype = sz
    lib.hg_wino_wgrad_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.hg_wino_wgrad.restype = ctypes.c_int
    lib.hg_wino_wgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]
    # include/hg_augment.h
    lib.hg_augment_spatial.restype = ctypes.c_int
    lib.hg_augment_spatial.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.hg_augment_workspace_bytes.restype = sz
    lib.hg_augment_workspace_bytes.argtypes = [i32]
    lib.hg_sample_mean.restype = ctypes.c_int
    lib.hg_sample_mean.argtypes = [vp, vp, i32, i64, vp, sz, vp]
    lib.hg_augment_color.restype = ctypes.c_int
    lib.hg_augment_color.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    return lib


lib = _load()

# every symbol include/hg_hist.h, hg_nets.h, hg_conv.h, hg_recolor.h, hg_augment.h, hg_linear.h and hg_wino.h declare
EXPORTS = ('hg_version', 'hg_error_string', 'hg_rgbuv_hist_workspace_bytes', 'hg_rgbuv_hist_uses_proj_cache', 'hg_rgbuv_hist_fwd',
           'hg_rgbuv_hist_bwd', 'hg_hellinger_workspace_bytes', 'hg_hellinger_fwd_bwd', 'hg_selftest_fastlog',
           'hg_modulate_fwd', 'hg_modulate_bwd', 'hg_demod_noise_lrelu_fwd', 'hg_demod_noise_lrelu_bwd',
           'hg_diffgrad_step', 'hg_diffgrad_step_size', 'hg_diffgrad_step_dev', 'hg_ema_update', 'hg_nets_workspace_bytes', 'hg_channel_sum', 'hg_lrelu_bwd_channel_sum', 'hg_demod_weight_term', 'hg_demod_style_grad', 'hg_demod_style_grad_workspace_bytes',
           'hg_conv_packed_elems', 'hg_conv_pack_weights', 'hg_conv_pack_weights_both', 'hg_conv_pack_blocks', 'hg_conv_pack_weights_multi', 'hg_conv2d_fwd', 'hg_conv2d_fwd_add', 'hg_modconv2d_fwd', 'hg_conv2d_workspace_bytes', 'hg_conv2d_plan', 'hg_conv2d_dgrad',
           'hg_conv2d_wgrad_workspace_bytes',
           'hg_conv2d_wgrad',
           'hg_instnorm_workspace_bytes', 'hg_instnorm_lrelu_fwd', 'hg_instnorm_lrelu_bwd', 'hg_stencil3',
           'hg_depthwise_valid', 'hg_augment_spatial', 'hg_augment_workspace_bytes', 'hg_sample_mean',
           'hg_augment_color', 'hg_grouped_linear_fwd', 'hg_grouped_linear_bwd_input_workspace_bytes',
           'hg_grouped_linear_bwd_input', 'hg_grouped_linear_bwd_params',
           'hg_wino_supported', 'hg_wino_packed_elems', 'hg_wino_pack_weights', 'hg_wino_pack_blocks', 'hg_wino_pack_weights_multi', 'hg_wino_workspace_bytes', 'hg_wino_conv2d',
           'hg_wino_wgrad_supported', 'hg_wino_wgrad_workspace_bytes', 'hg_wino_wgrad',
           'hg_torgb_fwd', 'hg_torgb_bwd_workspace_bytes', 'hg_torgb_bwd', 'hg_gstage_bwd_workspace_bytes', 'hg_gstage_bwd')


class HgError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib.hg_error_string(rc).decode()
        # the two argument errors of the reference are bare Exceptions with these messages
        # (histogram_classes/RGBuvHistBlock.py:90-93, 141-144)
        raise HgError(f'{what}: {msg} (code {rc})')


# ---- cheap host-side plumbing (the train step makes ~1 500 C-ABI calls; each microsecond here is 1.5 ms per step) ------
class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def on_device(device):
    """`with torch.cuda.device(device)` only when `device` is not already current (the context manager costs two driver
    calls; one process drives one GPU, so this is almost always the no-op)."""
    import torch
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NULL
    return torch.cuda.device(device)


def raw_stream(device):
    """The caller's current HIP stream on `device` as the void* the C ABI takes."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
