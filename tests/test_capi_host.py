"""not-gpu: the C-ABI library loads and exports every symbol include/occnet_amd.h declares; argument
validation runs without a GPU (no compute call is made); the product refuses host tensors."""
import ctypes

import pytest
import torch

from occnet_amd import _lib, ext


def test_library_exports_declared_symbols():
    lib = _lib.lib()
    declared = _lib.declared_symbols()
    assert 'occ_ms_deform_attn_forward_f32' in declared and 'occ_sca_fused_forward_f32' in declared
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.occ_abi_version() >= 1


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    rc = lib.occ_ms_deform_attn_forward_f32(null, null, null, null, null, null, 1, 1, 1, 1, 1, 1, 1,
                                            64, null)
    assert rc == -1 and b'null' in lib.occ_last_error()
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.occ_ms_deform_attn_forward_f32(p, p, p, p, p, p, 3, 1, 1, 1, 1, 1, 1, 2, null)
    assert rc == -1 and b'im2col_step' in lib.occ_last_error()     # batch 3 vs step 2
    rc = lib.occ_sca_fused_forward_f32(p, p, p, p, ctypes.c_int64(512), p, ctypes.c_int64(256), p, p,
                                       null, p, null, 1, 6, 100, 4, 64, 4, 8, 8, 10, null)
    assert rc == -3                                                # M=4,D=64: no fused kernel
    with pytest.raises(_lib.OccAmdUnsupported):
        _lib.check(rc, 'sca')
    rc = lib.occ_point_sampling_f32(p, p, p, p, ctypes.c_float(0.0), ctypes.c_float(1.0), p, p, null,
                                    1, 6, 4, 8, null)
    assert rc == -1


def test_product_refuses_host_tensors():
    v = torch.zeros(1, 4, 8, 32)
    shapes = torch.tensor([[2, 2]])
    start = torch.tensor([0])
    loc = torch.zeros(1, 3, 8, 1, 2, 2)
    aw = torch.zeros(1, 3, 8, 1, 2)
    with pytest.raises(_lib.OccAmdError, match='no CPU fallback'):
        ext.ms_deform_attn_forward(v, shapes, start, loc, aw, im2col_step=64)
    from occnet_amd.plugin import build_head
    from tests.util import head_cfg, small_cfg
    from occnet_amd import synthetic
    g = small_cfg(bev=(4, 4), feat_shapes=((2, 2), (1, 1), (1, 1), (1, 1)), num_layers=1)
    head = build_head(head_cfg(g)).eval()
    with pytest.raises(_lib.OccAmdError, match='no CPU fallback'), torch.no_grad():
        head(synthetic.make_features(g), synthetic.make_img_metas(g))


def test_ext_loader_surface():
    from occnet_amd import ext_loader
    m = ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])
    assert callable(m.ms_deform_attn_forward) and callable(m.ms_deform_attn_backward)
    with pytest.raises(AssertionError):
        ext_loader.load_ext('_ext', ['no_such_op'])
    from occnet_amd.plugin import (MultiScaleDeformableAttnFunction_fp16,
                                   MultiScaleDeformableAttnFunction_fp32)
    assert hasattr(MultiScaleDeformableAttnFunction_fp32, 'apply')
    assert hasattr(MultiScaleDeformableAttnFunction_fp16, 'apply')
