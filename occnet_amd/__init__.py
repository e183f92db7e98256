"""occnet_amd — MI355X (gfx950) native forward hot path of OccNet / BEVFormer-occ.

Layout:
  csrc/      hand-written HIP kernels + the C ABI declared in include/occnet_amd.h
  lib/       libocc_amd.so (built in-tree by `python -m occnet_amd.build`)
  _lib.py    ctypes binding of the C ABI (fails loudly if the library is missing)
  ext.py     mirror of the reference's `mmcv._ext` operator module
  plugin/    host-side mirror of projects/mmdet3d_plugin (registry names, modules, configs)
"""
__version__ = "0.1.0"


# Derived-weight caches (packed hi/lo Linear weights, concatenated query Linears, folded TSA position terms,
# per-(level, camera) value-projection biases, the decoder's folded BatchNorm, the folded inference backbone)
# are keyed on (data_ptr, tensor._version, cache_epoch()).  In-place updates through autograd-visible ops
# (optimizer steps, p.copy_ under no_grad) bump _version; writes through `param.data` do not — so everything
# that may write that way bumps the epoch instead: load_state_dict on any plugin module (post hook) and an explicit
# occnet_amd.invalidate_caches() after EMA / manual .data surgery.  (A train()/eval() mode change does not: the folded
# inference backbone, the one cache that holds COPIES without per-tensor keys, re-checks a (data_ptr, _version)
# signature of its source parameters the first time it is used after the model has been in training mode.)
_CACHE_EPOCH = 0


def cache_epoch():
    return _CACHE_EPOCH


def invalidate_caches(model=None):
    """Drop every derived-weight cache; with `model` (a BEVFormerOcc) also rebuild its folded inference
    backbone from the live parameters.  Call after writing parameters through `.data`."""
    global _CACHE_EPOCH
    _CACHE_EPOCH += 1
    from . import ext
    ext._PACKED_W.clear()
    ext._STACKED_VP.clear()
    ext._STACKED_GB.clear()
    ext._CHAIN_PACKS.clear()
    if model is not None and getattr(model, '_inference_backbone', None) is not None:
        model.enable_fused_backbone(**model._inference_backbone_args)
