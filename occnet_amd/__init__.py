"""occnet_amd — MI355X (gfx950) native forward hot path of OccNet / BEVFormer-occ.

Layout:
  csrc/      hand-written HIP kernels + the C ABI declared in include/occnet_amd.h
  lib/       libocc_amd.so (built in-tree by `python -m occnet_amd.build`)
  _lib.py    ctypes binding of the C ABI (fails loudly if the library is missing)
  ext.py     mirror of the reference's `mmcv._ext` operator module
  plugin/    host-side mirror of projects/mmdet3d_plugin (registry names, modules, configs)
"""
__version__ = "0.1.0"
