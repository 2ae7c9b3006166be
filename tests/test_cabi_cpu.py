"""CPU-side checks of the C ABI: the library builds/loads, exports every symbol include/hg_hist.h
declares, and its host-only entry points (argument validation, workspace sizing) behave.
No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    from histogan_amd import build
    build.build()
    import histogan_amd._lib as L
    return L


def _declared_symbols():
    syms = set()
    for fn in os.listdir(os.path.join(ROOT, 'include')):
        txt = open(os.path.join(ROOT, 'include', fn)).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        syms |= set(re.findall(r'\b(hg_[a-z0-9_]+)\s*\(', txt))
    return syms


def test_exports_match_header(L):
    declared = _declared_symbols()
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(L.lib, name), f'{name} declared in include/ but not exported'
    assert set(L.EXPORTS) <= declared


def test_version_and_error_strings(L):
    assert L.lib.hg_version() >= 100
    assert b'kernel method' in L.lib.hg_error_string(-2)
    assert b'resizing method' in L.lib.hg_error_string(-3)


def _params(L, **kw):
    p = L.HgHistParams()
    p.struct_size = ctypes.sizeof(L.HgHistParams)
    p.B, p.C, p.H, p.W = 2, 3, 16, 16
    p.stride_b, p.stride_c, p.stride_h, p.stride_w = 3 * 256, 256, 16, 1
    p.Hs, p.Ws, p.resize_mode = 16, 16, 0
    p.h, p.lo, p.hi, p.method, p.sigma = 64, -3.0, 3.0, 2, 0.02
    p.intensity_scale, p.green_only = 1, 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_workspace_query_and_validation(L):
    f, b = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.lib.hg_rgbuv_hist_workspace_bytes(ctypes.byref(_params(L)), ctypes.byref(f), ctypes.byref(b)) == 0
    assert f.value >= 2 * 3 * 64 * 64 * 4 and b.value > 0
    bad = [dict(method=7), dict(resize_mode=9), dict(h=0), dict(C=2), dict(sigma=0.0), dict(Hs=8)]
    codes = [L.lib.hg_rgbuv_hist_workspace_bytes(ctypes.byref(_params(L, **kw)), ctypes.byref(f), ctypes.byref(b))
             for kw in bad]
    assert codes[0] == -2 and codes[1] == -3 and all(c < 0 for c in codes)


def test_abi_guard_rejects_stale_or_unfilled_structs(L):
    """hg_hist_params.struct_size (version 102): a caller compiled against another layout, or one that did not fill the
    field in, is rejected before anything reads proj_cache / pre_relu; a garbage pre_relu or a misaligned cache pointer
    is rejected too."""
    f, b = ctypes.c_size_t(), ctypes.c_size_t()
    q = lambda **kw: L.lib.hg_rgbuv_hist_workspace_bytes(ctypes.byref(_params(L, **kw)), ctypes.byref(f), ctypes.byref(b))
    assert L.lib.hg_version() >= 102
    assert q() == 0
    assert q(struct_size=0) == -1 and q(struct_size=ctypes.sizeof(L.HgHistParams) - 16) == -1
    assert q(pre_relu=7) == -1 and q(proj_cache=0x1004) == -1 and q(proj_cache=0x1000) == 0


def test_proj_cache_query(L):
    """The 32 B / pixel projection cache is only wanted by the dense MFMA kernels."""
    use = lambda **kw: L.lib.hg_rgbuv_hist_uses_proj_cache(ctypes.byref(_params(L, **kw)))
    assert use() == 1                                    # inverse-quadratic: dense
    assert use(method=0) == 0                            # thresholding: scatter path
    assert use(method=1, sigma=0.02) == 0                # narrow RBF: truncated scatter / gather pair
    assert use(method=1, sigma=0.5) == 1                 # wide RBF: dense
    assert use(method=9) == -2


def test_null_pointers_rejected_before_any_launch(L):
    p = _params(L)
    assert L.lib.hg_rgbuv_hist_fwd(ctypes.byref(p), None, None, None, None, 0, None) == -1
    assert L.lib.hg_rgbuv_hist_bwd(ctypes.byref(p), None, None, None, None, None, None, 0, None) == -1
    assert L.lib.hg_hellinger_fwd_bwd(None, None, 10, 1, 1.0, None, None, None, 0, None) == -1


def test_cpu_tensor_is_refused_loudly(L):
    """The HIP path has no CPU fallback: its autograd Functions refuse CPU tensors.  (A module built with device='cpu' --
    the reference's Dataset use -- is a separate, HIP-free implementation: tests/test_hist_cpu_path.py.)"""
    import torch
    from histogan_amd.hist import HistConfig, HellingerFunction, RGBuvHistFunction
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        RGBuvHistFunction.apply(torch.rand(1, 3, 8, 8), HistConfig(h=16), False)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        HellingerFunction.apply(torch.rand(1, 3, 4, 4), torch.rand(1, 3, 4, 4), 1.0)


def test_pack_cache_registration_is_additive_and_weak():
    """a second model's enable_pack_cache() must not drop the first one's entries, and entries of a freed
    model must not be hit by a later tensor (host logic only, no kernels)."""
    import gc
    import torch
    from histogan_amd import conv as C
    C.enable_pack_cache(None)
    a = [torch.nn.Parameter(torch.zeros(4, 3, 3, 3)), torch.nn.Parameter(torch.zeros(5))]
    b = [torch.nn.Parameter(torch.zeros(2, 2, 1, 1))]
    C.enable_pack_cache(a)
    C.enable_pack_cache(b)
    ka, kb = (a[0].data_ptr(), (4, 3, 3, 3)), (b[0].data_ptr(), (2, 2, 1, 1))
    assert C._registered_owner(a[0], ka) is not None and C._registered_owner(b[0], kb) is not None
    assert len(C._cacheable) == 2                       # the 1-D parameter is not a convolution weight
    calls = []
    v1 = C.cached(a[0], 'tag', lambda t: calls.append(1) or 'x')
    v2 = C.cached(a[0], 'tag', lambda t: calls.append(1) or 'y')
    assert (v1, v2, len(calls)) == ('x', 'x', 1)        # cached
    C.weights_changed(a[0].data)
    assert C.cached(a[0], 'tag', lambda t: 'z') == 'z'  # invalidated through its owner buffer
    del a
    gc.collect()
    fake = torch.zeros(4, 3, 3, 3)
    assert C._registered_owner(fake, ka) is None and ka not in C._cacheable     # dead owner: dropped on lookup
    C.enable_pack_cache(None)
    assert not C._cacheable


def test_conv_launch_plan_fills_whole_rounds(L):
    """hg_conv2d_plan (host logic of hg_conv.hip: plan_conv / pick_ksplit_128 / short_k_chunks): the generator's layers at
    the benchmark shape (256^2, capacity 16, batch 32) get block counts that are whole rounds of (CUs x blocks per CU) --
    DESIGN.md section 8: 1024 blocks -> the 2-channel K-chunk kernels (4 blocks per CU), 256-block launches with a deep
    K -> K split 3 (768 blocks = 3 per CU), the 32-channel tile always on 2-channel chunks."""
    import ctypes

    def plan(B, K, N, S, k=3, stride=1, dgrad=0):
        out = (ctypes.c_int32 * 5)()
        assert L.lib.hg_conv2d_plan(B, K, N, S, S, k, stride, dgrad, out) == 0
        return dict(tile=out[0], ksplit=out[1], kc=out[2], blocks=out[3], cus=out[4])

    if plan(32, 256, 128, 64)['cus'] != 256:
        pytest.skip('plan expectations are written for the 256 CUs of an MI355X')
    T16, T32, T64W, T128, T128SM, T64 = range(6)
    p = plan(32, 256, 128, 64)                       # bench.py's roofline launch
    assert (p['tile'], p['ksplit'], p['kc'], p['blocks']) == (T128, 1, 2, 1024)
    p = plan(32, 1024, 512, 16)                      # 256 output tiles, 256 K chunks
    assert (p['tile'], p['ksplit'], p['blocks']) == (T128, 3, 768) and p['kc'] == 4
    p = plan(32, 512, 256, 32)                       # 512 blocks = 2 per CU: nothing to gain from a split
    assert (p['tile'], p['ksplit'], p['kc'], p['blocks']) == (T128, 1, 4, 512)
    p = plan(32, 128, 64, 128)                       # 64 ch x 256 px tile, 2048 blocks
    assert (p['tile'], p['kc'], p['blocks']) == (T64W, 2, 2048)
    assert plan(32, 64, 32, 256)['tile'] == T32 and plan(32, 64, 32, 256)['kc'] == 2
    assert plan(64, 16, 16, 256)['tile'] == T16      # first discriminator block: the 16x16x4 MFMA tile
    p = plan(32, 2048, 1024, 8)                      # 8x8 maps: 128 output tiles, split to whole rounds
    assert p['tile'] == T128 and p['ksplit'] > 1 and p['blocks'] % 256 == 0
    assert plan(32, 2048, 2048, 4)['tile'] == T128SM
    # the data gradient is planned as the convolution with the channel roles swapped
    assert plan(32, 128, 256, 64, dgrad=1) == plan(32, 128, 256, 64)
    # validation
    out = (ctypes.c_int32 * 5)()
    assert L.lib.hg_conv2d_plan(0, 1, 1, 4, 4, 3, 1, 0, out) < 0 and L.lib.hg_conv2d_plan(1, 1, 1, 4, 4, 5, 1, 0, out) < 0


def test_pack_multi_block_table_is_consistent(L):
    """hg_conv_pack_blocks (the per-weight block count of hg_conv_pack_weights_multi's table) covers both padded operands of
    hg_conv_packed_elems with 32 x 32 tiles, and the batched entry point validates its arguments before any launch."""
    for Co, Ci in [(16, 3), (2048, 2048), (3, 512), (130, 70), (1, 1)]:
        nb = L.lib.hg_conv_pack_blocks(Co, Ci)
        up = lambda v, m: (v + m - 1) // m * m
        assert nb == (up(Ci, 128) // 32) * (up(Co, 128) // 32)
        for k in (1, 3):
            # every element of both operands lies in one of the nb tiles (tile = 32 co x 32 ci x k*k)
            assert L.lib.hg_conv_packed_elems(Co, Ci, k, 0) <= nb * 32 * 32 * k * k
            assert L.lib.hg_conv_packed_elems(Co, Ci, k, 1) <= nb * 32 * 32 * k * k
    assert L.lib.hg_conv_pack_blocks(0, 4) == 0
    assert L.lib.hg_conv_pack_weights_multi(None, 1, 1, None) < 0
    assert L.lib.hg_conv_pack_weights_multi(1, 0, 1, None) < 0


def test_grouped_linear_argument_validation(L):
    """include/hg_linear.h, host side only: table checks and the workspace query (no launch)."""
    T = L.GlinLayer
    def tab(*rows):
        t = (T * len(rows))()
        for i, (N, g) in enumerate(rows):
            t[i].x, t[i].w, t[i].y, t[i].N, t[i].group = 0x1000, 0x2000, 0x3000, N, g
        return t
    q = lambda t, n, B, K: L.lib.hg_grouped_linear_bwd_input_workspace_bytes(t, n, B, K)
    t = tab((64, 0), (2048, 0), (2048, 0), (1024, 1))
    assert q(t, 4, 32, 512) == (1 + 16 + 16 + 8) * 32 * 512 * 4          # one slab per 128 output features
    assert q(t, 4, 65, 512) == 0 and q(t, 4, 32, 500) == 0               # batch > 64, K not a multiple of 32: unsupported
    assert q(tab((64, 1)), 1, 32, 512) == 0                                # groups start at 0
    assert q(tab((64, 0), (64, 2)), 2, 32, 512) == 0                       # ... and are contiguous
    assert q(tab((62, 0)), 1, 32, 512) == 0                                # rows of dy are read 16 bytes at a time
    assert L.lib.hg_grouped_linear_fwd(t, 0, 32, 512, None) == -1 and L.lib.hg_grouped_linear_fwd(None, 1, 32, 512, None) == -1
    assert L.lib.hg_grouped_linear_fwd(t, 4, 65, 512, None) == -5
    bad = tab((64, 0)); bad[0].w = 0x2004
    assert L.lib.hg_grouped_linear_fwd(bad, 1, 32, 512, None) == -1        # misaligned
    assert ctypes.sizeof(T) == 56


def test_wino_host_logic(L):
    """include/hg_wino.h, host side only (no launch): operand sizes / pack blocks of both channel-block variants, which
    shapes are served, workspace queries, argument validation."""
    lib = L.lib
    # 64-channel blocks, 8-channel chunks: 16 positions x blocks x chunks x 512 floats
    assert lib.hg_wino_packed_elems(128, 256, 0) == 16 * 2 * 32 * 512
    assert lib.hg_wino_packed_elems(128, 256, 1) == 16 * 4 * 16 * 512          # data gradient: K = Co, N = Ci
    # 32-channel variant (N <= 32): 4-channel chunks, 128 floats per (position, block, chunk)
    assert lib.hg_wino_packed_elems(32, 64, 0) == 16 * 1 * 16 * 128
    assert lib.hg_wino_packed_elems(64, 36, 0) == 0 and lib.hg_wino_packed_elems(0, 8, 0) == 0 and lib.hg_wino_packed_elems(8, 8, 2) == 0
    assert lib.hg_wino_pack_blocks(128, 256, 1, 1) == 2 * 32 + 4 * 16
    assert lib.hg_wino_pack_blocks(32, 64, 1, 0) == 4 * 1                      # variant 1: four chunks per 512-thread block
    # served shapes (256 CUs assumed without a GPU)
    assert lib.hg_wino_supported(32, 256, 128, 64, 64) == 1 and lib.hg_wino_supported(64, 1024, 2048, 2, 2) == 1
    assert lib.hg_wino_supported(32, 256, 128, 64, 63) == 0 and lib.hg_wino_supported(32, 16, 32, 128, 128) == 0
    assert lib.hg_wino_supported(32, 32, 32, 256, 256) == 1 and lib.hg_wino_supported(32, 64, 32, 256, 256) == 1
    assert lib.hg_wino_wgrad_supported(32, 256, 128, 64, 64) == 1 and lib.hg_wino_wgrad_supported(32, 2048, 2048, 4, 4) == 1
    assert lib.hg_wino_wgrad_supported(64, 1024, 2048, 2, 2) == 0 and lib.hg_wino_wgrad_supported(32, 32, 64, 64, 64) == 0
    # a launch that cannot fill the chip with output tiles splits K into slabs; a full one needs no scratch
    assert lib.hg_wino_workspace_bytes(32, 256, 128, 64, 64) == 0
    assert lib.hg_wino_workspace_bytes(32, 2048, 1024, 8, 8) % (32 * 1024 * 64 * 4) == 0 and lib.hg_wino_workspace_bytes(32, 2048, 1024, 8, 8) > 0
    assert lib.hg_wino_wgrad_workspace_bytes(32, 256, 128, 64, 64) == 32 * 16 * 128 * 256 * 4     # 8 (n, k) tiles -> 32 splits
    assert lib.hg_wino_wgrad_workspace_bytes(32, 64, 64, 24, 24) == 0
    # NULL pointers / bad epilogue combinations are rejected before any launch
    assert lib.hg_wino_conv2d(None, None, None, None, None, None, None, None, 0, 0.0, None, 1, 8, 8, 4, 4, None, 0, None) < 0
    assert lib.hg_wino_wgrad(None, None, None, 1, 8, 8, 4, 4, None, 0, None) < 0
    assert lib.hg_wino_pack_weights(None, None, 8, 8, 0, None) < 0


def test_torgb_host_logic(L):
    """hg_torgb_* (include/hg_nets.h), host side only: the adjoint's workspace follows the block geometry (<= 8 pixel blocks per
    image, (1 + C) partial sums per block and channel), unsupported shapes and bad arguments are refused before any launch."""
    import ctypes
    lib = L.lib
    # 256^2 map: 16 384 pixel quads, 64 per block -> 256 chunks -> 8 pixel blocks per image
    assert lib.hg_torgb_bwd_workspace_bytes(32, 32, 3, 256 * 256) == 32 * 8 * 32 * (1 + 3) * 4
    # 4x4 map: 4 quads -> one block of 4 quads x 64 channel groups
    assert lib.hg_torgb_bwd_workspace_bytes(32, 2048, 3, 16) == 32 * 1 * 2048 * (1 + 3) * 4
    assert lib.hg_torgb_bwd_workspace_bytes(2, 64, 4, 64 * 64) == 2 * 8 * 64 * (1 + 4) * 4        # rgba
    assert lib.hg_torgb_bwd_workspace_bytes(32, 32, 5, 256 * 256) == 0 and lib.hg_torgb_bwd_workspace_bytes(32, 32, 3, 6) == 0
    p = ctypes.c_void_p(4096)          # never dereferenced: every call below returns before a launch
    einval = lib.hg_torgb_fwd(None, None, None, None, None, 1, 8, 3, 16, None)
    unsup = lib.hg_torgb_fwd(p, p, p, None, p, 1, 8, 5, 16, None)
    assert einval < 0 and unsup < 0 and einval != unsup
    assert lib.hg_torgb_fwd(p, p, p, None, p, 1, 8, 3, 18, None) == unsup                         # pixels not a multiple of 4
    assert lib.hg_torgb_fwd(p, p, p, None, p, 1, 8192, 3, 16, None) == unsup                      # 3 x 8192 weights exceed the LDS plan
    assert lib.hg_torgb_bwd(p, p, p, p, p, None, p, 1, 8, 3, 16, p, 1 << 20, None) == einval      # style without its gradient
    ws = lib.hg_torgb_bwd_workspace_bytes(1, 8, 3, 16)
    short = lib.hg_torgb_bwd(p, p, p, p, p, p, p, 1, 8, 3, 16, p, ws - 1, None)
    assert short < 0 and short not in (einval, unsup)

