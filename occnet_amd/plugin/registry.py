"""Minimal registry / build_from_cfg with mmcv's call shapes, so the reference's configs
(`dict(type='BEVFormerOcc', ...)`) resolve without mmcv installed.

Registry names mirror the ones the reference registers into:
DETECTORS, HEADS, TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE, TRANSFORMER_LAYER, ATTENTION,
POSITIONAL_ENCODING, FEEDFORWARD_NETWORK plus the third-party ones the base config instantiates
(BACKBONES, NECKS, LOSSES, NORM_LAYERS, CONV_LAYERS, ACTIVATION_LAYERS) and the data/optimizer-side
ones that only need to parse (DATASETS, PIPELINES, SAMPLER, BBOX_ASSIGNERS, MATCH_COST, OPTIMIZERS,
RUNNERS).
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _decorator(cls):
            self._register(cls, name, force)
            return cls
        return _decorator

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f"`cfg` or `default_args` must contain the key \"type\", but got {cfg}")
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f"{obj_cls.__name__}: {e}")


DETECTORS = Registry('detector')
HEADS = Registry('head')
BACKBONES = Registry('backbone')
NECKS = Registry('neck')
LOSSES = Registry('loss')
TRANSFORMER = Registry('Transformer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
TRANSFORMER_LAYER = Registry('transformerLayer')
ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward Network')
POSITIONAL_ENCODING = Registry('position encoding')
NORM_LAYERS = Registry('norm layer')
CONV_LAYERS = Registry('conv layer')
ACTIVATION_LAYERS = Registry('activation layer')
# parse-only registries (data pipeline, optimisation, assignment): names resolve, nothing is built
DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')
SAMPLER = Registry('sampler')
BBOX_ASSIGNERS = Registry('bbox_assigner')
MATCH_COST = Registry('Match Cost')
OPTIMIZERS = Registry('optimizer')
RUNNERS = Registry('runner')


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


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_loss(cfg):
    return build_from_cfg(cfg, LOSSES)


def build_backbone(cfg):
    return build_from_cfg(cfg, BACKBONES)


def build_neck(cfg):
    return build_from_cfg(cfg, NECKS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build_from_cfg(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d.models.build_model call shape (reference: tools/train.py:215-219)."""
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


class _ParseOnly:
    """Placeholder class for names that only need to resolve (data pipeline, assigners, ...)."""

    def __init__(self, **kwargs):
        self.cfg = kwargs


def register_parse_only(registry, names):
    for n in names:
        if n not in registry:
            registry.register_module(name=n, module=type(n, (_ParseOnly,), {}))
