"""Run the reference's OWN module files under thin third-party stubs — TEST INFRASTRUCTURE.

The reference hot path is pure Python but imports mmcv / mmdet / mmdet3d / torchvision / cv2 at module
top level and its package __init__ chain JIT-compiles a CUDA extension, so it cannot
be imported as a package here.  This shim
  * registers stub modules for those third-party names in sys.modules (registries, BaseModule,
    init helpers, no-op fp16 decorators, FFN / ConvModule / LearnedPositionalEncoding / losses from
    oracle/thirdparty.py — the oracle's own restatements of mmcv/mmdet behaviour, SURVEY.md Appendix B.3-B.5,
    independent of the product package since round 4 — and `multi_scale_deformable_attn_pytorch` from
    oracle/msda.py, Appendix B.1);
  * creates empty package objects for `projects.mmdet3d_plugin...` whose __path__ points into
    /root/reference, so `importlib` executes the reference's module FILES (spatial_cross_attention.py,
    temporal_self_attention.py, encoder.py, custom_base_transformer_layer.py, transformer_occ.py,
    bevformer_occ_head.py, ...) unmodified, in place, without running any package __init__.
Everything the reference authored (rebatch, scatter, camera mean, TSA queue logic, reference points,
point_sampling, lifter, decoder wiring, head) therefore runs from the reference's real source; only
the third-party leaves are restated.  Used by oracle/gen_golden.py (in this container only —
/root/reference does not exist on the GPU box).  Nothing is copied out of the reference tree.
"""
import copy
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('OCC_REFERENCE_ROOT', '/root/reference')


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _pkg(name, path):
    m = _mod(name)
    m.__path__ = [path]
    return m


_INSTALLED = None


def install():
    """Install the stubs and return a namespace with the reference classes + builders (idempotent: the reference's
    modules register themselves into the registries of the FIRST call when they are imported)."""
    global _INSTALLED
    if _INSTALLED is not None:
        return _INSTALLED
    from oracle import thirdparty as pb          # NOT the product's bricks: the product must not define the golden
    from oracle.thirdparty import ConfigDict, Registry, build_from_cfg
    from oracle import model as om
    from oracle.msda import multi_scale_deformable_attn_pytorch

    ATTENTION = Registry('attention')
    TRANSFORMER_LAYER = Registry('transformerLayer')
    TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
    FEEDFORWARD_NETWORK = Registry('feed-forward Network')
    POSITIONAL_ENCODING = Registry('position encoding')
    PLUGIN_LAYERS = Registry('plugin layer')
    TRANSFORMER = Registry('Transformer')
    HEADS = Registry('head')
    LOSSES = Registry('loss')
    DETECTORS = Registry('detector')
    FEEDFORWARD_NETWORK.register_module(name='FFN', module=pb.FFN)
    POSITIONAL_ENCODING.register_module(name='LearnedPositionalEncoding',
                                        module=pb.LearnedPositionalEncoding)
    LOSSES.register_module(name='CrossEntropyLoss', module=pb.CrossEntropyLoss)
    LOSSES.register_module(name='L1Loss', module=pb.L1Loss)

    def build_attention(cfg, default_args=None):
        return build_from_cfg(cfg, ATTENTION, default_args)

    def build_feedforward_network(cfg, default_args=None):
        return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)

    def build_positional_encoding(cfg, default_args=None):
        return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)

    def build_transformer_layer(cfg, default_args=None):
        return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)

    def build_transformer_layer_sequence(cfg, default_args=None):
        return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)

    class TransformerLayerSequence(pb.BaseModule):
        def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
            super().__init__(init_cfg)
            if isinstance(transformerlayers, dict):
                transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
            self.num_layers = num_layers
            self.layers = nn.ModuleList([build_transformer_layer(c) for c in transformerlayers])
            self.embed_dims = self.layers[0].embed_dims
            self.pre_norm = self.layers[0].pre_norm

    def _identity_decorator_factory(*dargs, **dkwargs):
        def deco(fn):
            return fn
        return deco

    class _ExtStub:
        def __getattr__(self, name):
            def _raise(*a, **k):
                raise RuntimeError(f'mmcv._ext.{name} is CUDA-only; the CPU branch must be taken')
            return _raise

    def digit_version(v):
        out = []
        for p in str(v).split('+')[0].split('.'):
            digits = ''.join(ch for ch in p if ch.isdigit())
            out.append(int(digits) if digits else 0)
        return tuple(out)

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mmcv = _mod('mmcv', ConfigDict=ConfigDict, deprecated_api_warning=_identity_decorator_factory)
    _mod('mmcv.utils', ext_loader=types.SimpleNamespace(load_ext=lambda name, funcs: _ExtStub()),
         TORCH_VERSION=torch.__version__, digit_version=digit_version, ConfigDict=ConfigDict,
         build_from_cfg=build_from_cfg, deprecated_api_warning=_identity_decorator_factory,
         to_2tuple=lambda x: (x, x))
    _mod('mmcv.cnn', xavier_init=pb.xavier_init, constant_init=pb.constant_init, Linear=nn.Linear,
         build_activation_layer=pb.build_activation_layer, build_norm_layer=pb.build_norm_layer,
         PLUGIN_LAYERS=PLUGIN_LAYERS, Conv2d=nn.Conv2d, Conv3d=nn.Conv3d, ConvModule=pb.ConvModule,
         caffe2_xavier_init=lambda m, bias=0: nn.init.kaiming_uniform_(m.weight, a=1),
         bias_init_with_prob=lambda p: float(-torch.log(torch.tensor((1 - p) / p))))
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.registry', ATTENTION=ATTENTION, TRANSFORMER_LAYER=TRANSFORMER_LAYER,
         TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE,
         FEEDFORWARD_NETWORK=FEEDFORWARD_NETWORK, POSITIONAL_ENCODING=POSITIONAL_ENCODING)
    _mod('mmcv.cnn.bricks.transformer', build_attention=build_attention,
         build_feedforward_network=build_feedforward_network,
         build_positional_encoding=build_positional_encoding,
         build_transformer_layer=build_transformer_layer,
         build_transformer_layer_sequence=build_transformer_layer_sequence,
         TransformerLayerSequence=TransformerLayerSequence)
    _mod('mmcv.runner', force_fp32=_identity_decorator_factory, auto_fp16=_identity_decorator_factory,
         BaseModule=pb.BaseModule, ModuleList=nn.ModuleList, Sequential=nn.Sequential)
    _mod('mmcv.runner.base_module', BaseModule=pb.BaseModule, ModuleList=nn.ModuleList,
         Sequential=nn.Sequential)
    _mod('mmcv.ops')
    _mod('mmcv.ops.multi_scale_deform_attn',
         multi_scale_deformable_attn_pytorch=multi_scale_deformable_attn_pytorch,
         MultiScaleDeformableAttention=_Dummy)
    _mod('cv2')
    if 'matplotlib' not in sys.modules:
        try:
            importlib.import_module('matplotlib.pyplot')
        except Exception:
            _mod('matplotlib')
            _mod('matplotlib.pyplot')
    _mod('torchvision')
    _mod('torchvision.utils', make_grid=None)
    _mod('torchvision.transforms')
    _mod('torchvision.transforms.functional',
         rotate=lambda img, angle, center=None: om.rotate_nearest(img, angle, center))
    _mod('mmdet')
    _mod('mmdet.core', multi_apply=None, reduce_mean=None)
    _mod('mmdet.models', HEADS=HEADS, DETECTORS=DETECTORS)
    _mod('mmdet.models.utils', build_transformer=lambda cfg, default_args=None:
         build_from_cfg(cfg, TRANSFORMER, default_args))
    _mod('mmdet.models.utils.builder', TRANSFORMER=TRANSFORMER)
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=None)
    _mod('mmdet.models.builder', build_loss=lambda cfg: build_from_cfg(cfg, LOSSES))
    _mod('mmdet.models.dense_heads', DETRHead=_Dummy)
    _mod('mmdet3d')
    _mod('mmdet3d.core')
    _mod('mmdet3d.core.bbox')
    _mod('mmdet3d.core.bbox.coders', build_bbox_coder=None)

    P = os.path.join(REF_ROOT, 'projects')
    PL = os.path.join(P, 'mmdet3d_plugin')
    _pkg('projects', P)
    _pkg('projects.mmdet3d_plugin', PL)
    _pkg('projects.mmdet3d_plugin.bevformer', os.path.join(PL, 'bevformer'))
    _pkg('projects.mmdet3d_plugin.bevformer.modules', os.path.join(PL, 'bevformer', 'modules'))
    _pkg('projects.mmdet3d_plugin.bevformer.dense_heads', os.path.join(PL, 'bevformer', 'dense_heads'))
    _pkg('projects.mmdet3d_plugin.models', os.path.join(PL, 'models'))
    _pkg('projects.mmdet3d_plugin.models.utils', os.path.join(PL, 'models', 'utils'))
    _pkg('projects.mmdet3d_plugin.core', os.path.join(PL, 'core'))
    _pkg('projects.mmdet3d_plugin.core.bbox', os.path.join(PL, 'core', 'bbox'))
    _mod('projects.mmdet3d_plugin.models.utils.visual', save_tensor=lambda *a, **k: None)

    mods = 'projects.mmdet3d_plugin.bevformer.modules.'
    sca = importlib.import_module(mods + 'spatial_cross_attention')
    tsa = importlib.import_module(mods + 'temporal_self_attention')
    enc = importlib.import_module(mods + 'encoder')
    trf = importlib.import_module(mods + 'transformer_occ')
    head = importlib.import_module('projects.mmdet3d_plugin.bevformer.dense_heads.bevformer_occ_head')
    for m in (sca, tsa, enc, trf, head):
        assert m.__file__.startswith(REF_ROOT), m.__file__
    _INSTALLED = types.SimpleNamespace(
        SpatialCrossAttention=sca.SpatialCrossAttention,
        MSDeformableAttention3D=sca.MSDeformableAttention3D,
        TemporalSelfAttention=tsa.TemporalSelfAttention,
        BEVFormerEncoder=enc.BEVFormerEncoder, BEVFormerLayer=enc.BEVFormerLayer,
        TransformerOcc=trf.TransformerOcc, BEVFormerOccHead=head.BEVFormerOccHead,
        build_head=lambda cfg: build_from_cfg(cfg, HEADS),
        files=[m.__file__ for m in (sca, tsa, enc, trf, head)])
    return _INSTALLED


# ------------------------------------------------------------------------------------------------------------
# The reference's METRIC code, executed from where it lies.
#   projects/mmdet3d_plugin/datasets/ray_metrics.py JIT-compiles its CUDA ray caster at import
#   (`dvr = load("dvr", sources=[...dvr.cpp, ...dvr.cu])`, :12), imports prettytable and moves tensors with
#   .cuda() (:117-120); tools/ray_iou/metric.py is plain numpy.  With
#     torch.utils.cpp_extension.load -> an object whose render_forward() runs the reference's OWN kernel body
#                                       compiled for the host (oracle/_ref/libdvr_reference.so, oracle/build_ref.py),
#     prettytable                    -> a 10-line stand-in,
#     Tensor.cuda                    -> identity (while a reference function runs)
#   both files run unmodified; oracle/gen_golden.py records their outputs.
def _reference_dvr(lib_name='libdvr_reference.so'):
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', lib_name)
    if not os.path.exists(so):
        from oracle import build_ref
        build_ref.build(verbose=False)
    lib = ctypes.CDLL(so)

    def render_forward(sigma, origin, points, tindex, grid=None, phase_name="test"):
        sigma, origin, points, tindex = (t.detach().cpu().contiguous().float()
                                         for t in (sigma, origin, points, tindex))
        N, T, Z, Y, X = sigma.shape
        assert grid is None or [int(v) for v in grid] == [T, Z, Y, X]
        M = points.shape[1]
        pred, gt, coord = torch.empty(N, M), torch.empty(N, M), torch.empty(N, M, 3)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        lib.dvr_render_forward_reference(p(sigma), p(origin), p(points), p(tindex), p(pred), p(gt), p(coord),
                                         N, T, Z, Y, X, M, points.shape[2], 1 if phase_name == "train" else 0)
        return pred, gt, coord
    return types.SimpleNamespace(render_forward=render_forward)


class _cpu_cuda:
    """`tensor.cuda()` is the identity while the reference's metric functions run on this GPU-less host."""

    def __enter__(self):
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        self._ec = torch.cuda.empty_cache
        torch.cuda.empty_cache = lambda: None

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._orig
        torch.cuda.empty_cache = self._ec
        return False


def install_metrics():
    """-> namespace(ray_metrics=<reference module>, metric=<reference module>, on_host=<context manager>,
    files=[...]) with both reference files executed in place."""
    import importlib.util
    import torch.utils.cpp_extension as cpp_ext

    class PrettyTable:
        def __init__(self, field_names=None):
            self.field_names, self.rows, self.float_format = field_names, [], ''

        def add_row(self, row, divider=False):
            self.rows.append(list(row))

        def __str__(self):
            return '\n'.join(' | '.join(f'{v:.3f}' if isinstance(v, float) else str(v) for v in r)
                             for r in [self.field_names] + self.rows)
    saved_load, saved_pt = cpp_ext.load, sys.modules.get('prettytable')
    cpp_ext.load = lambda name, sources=None, **kw: _reference_dvr()
    _mod('prettytable', PrettyTable=PrettyTable)
    try:
        mods = {}
        for name, rel in (('ref_ray_metrics', 'projects/mmdet3d_plugin/datasets/ray_metrics.py'),
                          ('ref_ray_iou_metric', 'tools/ray_iou/metric.py')):
            path = os.path.join(REF_ROOT, rel)
            spec = importlib.util.spec_from_file_location(name, path)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            assert m.__file__.startswith(REF_ROOT)
            mods[name] = m
    finally:
        cpp_ext.load = saved_load
        if saved_pt is None:
            sys.modules.pop('prettytable', None)
        else:
            sys.modules['prettytable'] = saved_pt
    return types.SimpleNamespace(ray_metrics=mods['ref_ray_metrics'], metric=mods['ref_ray_iou_metric'],
                                 on_host=_cpu_cuda, files=[m.__file__ for m in mods.values()])


# ------------------------------------------------------------------------------------------------------------
# The reference's INPUT / OUTPUT FORMAT code, executed from where it lies:
#   projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py  (PadMultiViewImage :12-62, NormalizeMultiviewImage :65-101)
#   projects/mmdet3d_plugin/datasets/pipelines/loading.py       (LoadOccGTFromFile :7-38)
#   projects/mmdet3d_plugin/datasets/nuscenes_occ.py            (NuSceneOcc.get_data_info :49-126, format_results :189-257)
#   tools/ray_iou/ego_pose_extractor.py                         (EgoPoseDataset: the lidar origins format_results casts from)
# Third-party leaves restated (none of them is under /root/reference): mmcv.impad / impad_to_multiple / imnormalize
# (mmcv/image/geometric.py, photometric.py: bottom/right constant padding; `(img - mean) * (1 / std)` in float32 with
# optional BGR->RGB), mmcv.load (pickle), pyquaternion.Quaternion.rotation_matrix and
# nuscenes.utils.geometry_utils.transform_matrix (oracle/thirdparty.py, independent of occnet_amd.io since round 4),
# NuScenesDataset (attribute holder), tqdm (real), cv2 (unused here).
# nuscenes_occ.py does `from ....tools.ray_iou.ego_pose_extractor import EgoPoseDataset` — a relative import that climbs
# above `projects`; the files are therefore mounted under a synthetic top-level package `occref` = /root/reference.
def install_datasets():
    """-> namespace(PadMultiViewImage, NormalizeMultiviewImage, LoadOccGTFromFile, NuSceneOcc, EgoPoseDataset, files)."""
    import importlib.util
    import pickle
    import numpy as np
    import torch.utils.cpp_extension as cpp_ext
    from oracle import thirdparty as oio         # oracle-owned restatements (not occnet_amd.io)
    from oracle.thirdparty import Registry

    PIPELINES, DATASETS = Registry('pipeline'), Registry('dataset')

    def impad(img, *, shape=None, padding=None, pad_val=0, padding_mode='constant'):
        assert padding_mode == 'constant' and shape is not None
        out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
        out[:img.shape[0], :img.shape[1]] = img
        return out

    def impad_to_multiple(img, divisor, pad_val=0):
        pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
        pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
        return impad(img, shape=(pad_h, pad_w), pad_val=pad_val)

    def imnormalize(img, mean, std, to_rgb=True):
        img = img.copy().astype(np.float32)
        mean64 = np.float64(mean.reshape(1, -1))
        stdinv = 1 / np.float64(std.reshape(1, -1))
        if to_rgb:
            img = img[..., ::-1]
        # cv2.subtract / cv2.multiply on a float32 image with a double scalar: float32 result
        return ((img - mean64.astype(np.float32)) * stdinv.astype(np.float32)).astype(np.float32)

    class Quaternion:
        def __init__(self, *a, **k):
            q = a[0] if len(a) == 1 else a
            self.q = np.asarray(q.q if isinstance(q, Quaternion) else q, dtype=np.float64)

        @property
        def rotation_matrix(self):
            return oio.quaternion_rotation_matrix(self.q)

    class NuScenesDataset:
        def __init__(self, ann_file=None, data_root=None, modality=None, test_mode=False, load_interval=1, **kw):
            self.ann_file, self.data_root, self.test_mode, self.load_interval = ann_file, data_root, test_mode, load_interval
            self.modality = modality or dict(use_camera=True)

    def mmcv_load(path):
        with open(path, 'rb') as f:
            return pickle.load(f)

    saved = {k: sys.modules.get(k) for k in ('mmcv', 'mmdet', 'mmdet3d', 'cv2', 'nuscenes', 'pyquaternion', 'prettytable')}
    _mod('mmcv', impad=impad, impad_to_multiple=impad_to_multiple, imnormalize=imnormalize, load=mmcv_load,
         mkdir_or_exist=lambda d: os.makedirs(d, exist_ok=True))
    _mod('mmcv.parallel', DataContainer=object)
    _mod('mmdet')
    _mod('mmdet.datasets', DATASETS=DATASETS)
    _mod('mmdet.datasets.builder', PIPELINES=PIPELINES)
    _mod('mmdet3d')
    _mod('mmdet3d.datasets', NuScenesDataset=NuScenesDataset)
    _mod('cv2')
    _mod('nuscenes')
    _mod('nuscenes.eval')
    _mod('nuscenes.eval.common')
    _mod('nuscenes.eval.common.utils', quaternion_yaw=None, Quaternion=Quaternion)
    _mod('nuscenes.utils')
    _mod('nuscenes.utils.geometry_utils',
         transform_matrix=lambda translation=np.array([0, 0, 0]), rotation=Quaternion([1, 0, 0, 0]), inverse=False:
         oio.transform_matrix(translation, rotation.q, inverse=inverse))
    _mod('nuscenes.nuscenes', NuScenes=None)
    _mod('pyquaternion', Quaternion=Quaternion)

    class PrettyTable:
        def __init__(self, field_names=None):
            self.field_names, self.rows, self.float_format = field_names, [], ''

        def add_row(self, row, divider=False):
            self.rows.append(list(row))
    _mod('prettytable', PrettyTable=PrettyTable)
    saved_load = cpp_ext.load
    cpp_ext.load = lambda name, sources=None, **kw: _reference_dvr()
    try:
        _pkg('occref', REF_ROOT)
        _pkg('occref.tools', os.path.join(REF_ROOT, 'tools'))
        _pkg('occref.tools.ray_iou', os.path.join(REF_ROOT, 'tools', 'ray_iou'))
        P = os.path.join(REF_ROOT, 'projects')
        _pkg('occref.projects', P)
        _pkg('occref.projects.mmdet3d_plugin', os.path.join(P, 'mmdet3d_plugin'))
        _pkg('occref.projects.mmdet3d_plugin.datasets', os.path.join(P, 'mmdet3d_plugin', 'datasets'))
        _pkg('occref.projects.mmdet3d_plugin.datasets.pipelines', os.path.join(P, 'mmdet3d_plugin', 'datasets', 'pipelines'))
        base = 'occref.projects.mmdet3d_plugin.datasets.'
        t3d = importlib.import_module(base + 'pipelines.transform_3d')
        loading = importlib.import_module(base + 'pipelines.loading')
        nus = importlib.import_module(base + 'nuscenes_occ')
        ego = importlib.import_module('occref.tools.ray_iou.ego_pose_extractor')
    finally:
        cpp_ext.load = saved_load
        for k, v in saved.items():          # leave the stubs of install() / install_metrics() as they were
            if v is not None:
                sys.modules[k] = v
    for m in (t3d, loading, nus, ego):
        assert m.__file__.startswith(REF_ROOT), m.__file__
    return types.SimpleNamespace(
        PadMultiViewImage=t3d.PadMultiViewImage, NormalizeMultiviewImage=t3d.NormalizeMultiviewImage,
        LoadOccGTFromFile=loading.LoadOccGTFromFile, NuSceneOcc=nus.NuSceneOcc, EgoPoseDataset=ego.EgoPoseDataset,
        nuscenes_occ_module=nus, on_host=_cpu_cuda, files=[m.__file__ for m in (t3d, loading, nus, ego)])
