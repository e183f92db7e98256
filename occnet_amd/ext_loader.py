"""`ext_loader.load_ext('_ext', [...])` — same call shape as mmcv.utils.ext_loader (reference:
projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:10-12), resolving
to occnet_amd.ext backed by libocc_amd.so instead of mmcv's CUDA extension."""
import importlib


def load_ext(name, funcs):
    if name != '_ext':
        raise ImportError(f"unknown extension module {name!r}")
    ext = importlib.import_module('occnet_amd.ext')
    for fun in funcs:
        assert hasattr(ext, fun), f'{fun} miss in module {name}'
    return ext
