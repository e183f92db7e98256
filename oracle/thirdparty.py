"""The oracle's OWN restatements of the third-party leaves the reference imports — TEST INFRASTRUCTURE.

mmcv-full 1.4 / mmdet 2.14 are not in /root/reference (un-vendored pip dependencies, docs/getting_started.md:3) and not in
this image.  oracle/refshim.py executes the reference's module files in place and needs these names to exist; until
round 3 it borrowed them from the product package (occnet_amd.plugin.bricks), so for these leaves a "reference golden"
compared the product with itself.  This file restates them from their published behaviour
(SURVEY.md Appendix B.3-B.5), independently of occnet_amd — nothing here imports the product:

  Registry / build_from_cfg / ConfigDict   mmcv.utils.registry, mmcv.utils.config (the subset the reference calls)
  BaseModule, xavier_init, constant_init   mmcv.runner.base_module, mmcv.cnn.utils.weight_init
  build_norm_layer, build_activation_layer mmcv.cnn.bricks.norm / activation
  FFN                                      mmcv.cnn.bricks.transformer.FFN (mmcv 1.4: `layers`, `dropout_layer`, `add_identity`)
  ConvModule                               mmcv.cnn.bricks.conv_module.ConvModule, order ('conv', 'norm', 'act')
  LearnedPositionalEncoding                mmdet.models.utils.positional_encoding
  CrossEntropyLoss, L1Loss                 mmdet.models.losses (softmax CE branch; weight_reduce_loss semantics)

tests/test_oracle_golden.py::test_thirdparty_restatements_agree pins these against the product's restatements
(two independent implementations must agree to the bit on seeded inputs).
"""
import inspect

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- registry / config ---------------------------------------------------------------------------------------------
class ConfigDict(dict):
    """Attribute-style dict (mmcv.ConfigDict is addict.Dict based): cfg.key == cfg['key'], nested dicts wrapped."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            keys = [name] if isinstance(name, str) else (name or [cls.__name__])
            for k in keys:
                if k in self._module_dict and not force:
                    raise KeyError(f'{k} is already registered in {self.name}')
                self._module_dict[k] = cls
            return cls
        if module is not None:
            _reg(module)
            return module
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError('cfg must be a dict with the key "type"')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or a class, got {type(obj_type)}')
    return obj_cls(**args)


# ---- base module / init --------------------------------------------------------------------------------------------
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    def init_weights(self):
        for child in self.children():
            if hasattr(child, 'init_weights'):
                child.init_weights()
        self._is_init = True


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ('uniform', 'normal')
    if hasattr(module, 'weight') and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


# ---- norm / activation builders ------------------------------------------------------------------------------------
_NORMS = {'BN': ('bn', nn.BatchNorm2d), 'BN1d': ('bn', nn.BatchNorm1d), 'BN2d': ('bn', nn.BatchNorm2d),
          'BN3d': ('bn', nn.BatchNorm3d), 'LN': ('ln', nn.LayerNorm), 'GN': ('gn', nn.GroupNorm)}
_ACTS = {'ReLU': nn.ReLU, 'LeakyReLU': nn.LeakyReLU, 'GELU': nn.GELU, 'Sigmoid': nn.Sigmoid, 'Tanh': nn.Tanh,
         'Softplus': nn.Softplus}
_CONVS = {None: nn.Conv2d, 'Conv': nn.Conv2d, 'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d, 'Conv3d': nn.Conv3d}


def build_norm_layer(cfg, num_features, postfix=''):
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in _NORMS:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    abbr, cls = _NORMS[layer_type]
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type == 'GN':
        layer = cls(num_channels=num_features, **cfg_)
    else:
        layer = cls(num_features, **cfg_)
    for param in layer.parameters():
        param.requires_grad = requires_grad
    return abbr + str(postfix), layer


def build_activation_layer(cfg):
    cfg_ = dict(cfg)
    return _ACTS[cfg_.pop('type')](**cfg_)


# ---- FFN -----------------------------------------------------------------------------------------------------------
class FFN(BaseModule):
    """mmcv 1.4 FFN: layers = Sequential(Sequential(Linear, act, Dropout) * (num_fcs - 1), Linear, Dropout);
    forward(x, identity=None) = (identity if given else x) + dropout_layer(layers(x))   [add_identity=True]."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2, f'num_fcs should be no less than 2. got {num_fcs}.'
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        self.act_cfg = act_cfg
        self.activate = build_activation_layer(act_cfg)
        stages = []
        width = embed_dims
        for _ in range(num_fcs - 1):
            stages.append(nn.Sequential(nn.Linear(width, feedforward_channels), self.activate, nn.Dropout(ffn_drop)))
            width = feedforward_channels
        stages.append(nn.Linear(feedforward_channels, embed_dims))
        stages.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*stages)
        if dropout_layer:
            assert dropout_layer.get('type') in ('Dropout', None), 'only plain Dropout is restated'
            self.dropout_layer = nn.Dropout(dropout_layer.get('drop_prob', dropout_layer.get('p', 0.5)))
        else:
            self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


# ---- ConvModule ----------------------------------------------------------------------------------------------------
class ConvModule(nn.Module):
    """mmcv ConvModule, default order ('conv', 'norm', 'act'); bias='auto' -> bias iff no norm; sub-modules `conv`,
    `<bn|gn|ln>` (norm_name), `activate`; kaiming-normal init of the conv (fan_out, relu), norm weight 1 / bias 0."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, with_spectral_norm=False,
                 padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        assert tuple(order) == ('conv', 'norm', 'act') and not with_spectral_norm and padding_mode == 'zeros'
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        conv_cls = _CONVS[None if conv_cfg is None else conv_cfg['type']]
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                             groups=groups, bias=bias)
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act = dict(act_cfg)
            if act['type'] not in ('Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish', 'GELU', 'Softplus'):
                act.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm:
            x = self.norm(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x


# ---- positional encoding -------------------------------------------------------------------------------------------
class LearnedPositionalEncoding(BaseModule):
    """mmdet: pos (bs, 2 * num_feats, h, w) = cat(col_embed(arange(w)) over rows, row_embed(arange(h)) over columns)."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=dict(type='Uniform', layer='Embedding')):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats = num_feats
        self.row_num_embed = row_num_embed
        self.col_num_embed = col_num_embed
        nn.init.uniform_(self.row_embed.weight, 0, 1)      # init_cfg Uniform on Embedding layers (a=0, b=1)
        nn.init.uniform_(self.col_embed.weight, 0, 1)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x = torch.arange(w, device=mask.device)
        y = torch.arange(h, device=mask.device)
        x_embed = self.col_embed(x)
        y_embed = self.row_embed(y)
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)), dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)


# ---- losses --------------------------------------------------------------------------------------------------------
def _weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """mmdet.models.losses.utils.weight_reduce_loss."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class CrossEntropyLoss(nn.Module):
    """mmdet CrossEntropyLoss, softmax branch (use_sigmoid=False, use_mask=False)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, ignore_index=None,
                 loss_weight=1.0, **kwargs):
        super().__init__()
        assert not use_sigmoid and not use_mask
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.class_weight = class_weight
        self.ignore_index = ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None,
                **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if ignore_index is None:
            ignore_index = self.ignore_index
        ignore_index = -100 if ignore_index is None else ignore_index
        class_weight = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        loss = F.cross_entropy(cls_score, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * _weight_reduce_loss(loss, weight, reduction, avg_factor)


class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        loss = torch.abs(pred - target)
        return self.loss_weight * _weight_reduce_loss(loss, weight, reduction, avg_factor)


# ---- geometry leaves of the dataset code (pyquaternion, nuscenes-devkit) -----------------------------------------------
def quaternion_rotation_matrix(q):
    """pyquaternion.Quaternion.rotation_matrix: normalise, then the lower-right 3x3 block of Q . conj(Qbar)^T."""
    import numpy as np
    q = np.asarray(q, dtype=np.float64)
    n = np.sqrt(np.dot(q, q))
    if abs(1.0 - n) > 1e-14 and n > 0:
        q = q / n
    a, b, c, d = q
    qm = np.array([[a, -b, -c, -d], [b, a, -d, c], [c, d, a, -b], [d, -c, b, a]])
    qbar = np.array([[a, -b, -c, -d], [b, a, d, -c], [c, -d, a, b], [d, c, -b, a]])
    return np.dot(qm, qbar.conj().transpose())[1:][:, 1:]


def transform_matrix(translation, rotation_q, inverse=False):
    """nuscenes.utils.geometry_utils.transform_matrix with the rotation given as a (w, x, y, z) quaternion."""
    import numpy as np
    tm = np.eye(4)
    rot = quaternion_rotation_matrix(rotation_q)
    if inverse:
        rot_inv = rot.T
        trans = np.transpose(-np.array(translation, dtype=np.float64))
        tm[:3, :3] = rot_inv
        tm[:3, 3] = rot_inv.dot(trans)
    else:
        tm[:3, :3] = rot
        tm[:3, 3] = np.transpose(np.array(translation, dtype=np.float64))
    return tm
