"""Third-party building blocks the reference's configs instantiate by name (mmcv / mmdet), restated
on stock torch.nn so configs resolve without mmcv: BaseModule, FFN, norm / activation builders,
ConvModule, LearnedPositionalEncoding, CrossEntropyLoss, L1Loss.
State-dict key layout matches mmcv's (`layers.0.0.*`, `layers.1.*`, `conv.*`, `bn.*`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import (ACTIVATION_LAYERS, CONV_LAYERS, FEEDFORWARD_NETWORK, LOSSES, NORM_LAYERS,
                       POSITIONAL_ENCODING, build_from_cfg)


def _bump_cache_epoch(*_):
    import occnet_amd
    occnet_amd._CACHE_EPOCH += 1


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule call shape: BaseModule(init_cfg) + init_weights()."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg
        # parameters rewritten through load_state_dict (param.data.copy_: no _version bump) make every derived-weight
        # cache stale: bump the epoch.  train() / eval() do NOT: optimizer steps and BatchNorm statistics update their
        # tensors in place (the _version every cache key carries), and obtain_history_bev flips the mode twice per
        # training step (the bump refolded the backbone prefix plan and repacked every weight each step)
        self.register_load_state_dict_post_hook(_bump_cache_epoch)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


ModuleList = nn.ModuleList
Sequential = nn.Sequential


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if module is None:
        return
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


for _n, _c in (('BN', nn.BatchNorm2d), ('BN1d', nn.BatchNorm1d), ('BN2d', nn.BatchNorm2d),
               ('BN3d', nn.BatchNorm3d), ('LN', nn.LayerNorm), ('GN', nn.GroupNorm)):
    NORM_LAYERS.register_module(name=_n, module=_c)
for _n, _c in (('Conv1d', nn.Conv1d), ('Conv2d', nn.Conv2d), ('Conv3d', nn.Conv3d), ('Conv', nn.Conv2d)):
    CONV_LAYERS.register_module(name=_n, module=_c)
for _n, _c in (('ReLU', nn.ReLU), ('GELU', nn.GELU), ('Sigmoid', nn.Sigmoid), ('Tanh', nn.Tanh),
               ('LeakyReLU', nn.LeakyReLU), ('Softplus', nn.Softplus)):
    ACTIVATION_LAYERS.register_module(name=_n, module=_c)


def build_norm_layer(cfg, num_features, postfix=''):
    """-> (name, layer) like mmcv.cnn.build_norm_layer."""
    cfg = dict(cfg)
    layer_type = cfg.pop('type')
    cls = NORM_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    abbr = {'LN': 'ln', 'GN': 'gn'}.get(layer_type, 'bn')
    return abbr + str(postfix), layer


class X3Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose TRAINING forward and backward run on the bf16x3
    matrix-core kernels (ext.LinearX3Function: forward + dx on linear_bf16x3.hip, dW / db on linear_wgrad.hip)
    instead of ATen's fp32 library GEMMs.  Without autograd, on the host, or for shapes the kernels do not cover it
    is F.linear."""

    def forward(self, x, act=None):
        from .. import ext
        if x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return ext.linear_autograd(x, self.weight, self.bias, act=act)
        y = F.linear(x, self.weight, self.bias)
        return torch.relu(y) if act == 'relu' else y


def build_activation_layer(cfg):
    return build_from_cfg(cfg, ACTIVATION_LAYERS)


@FEEDFORWARD_NETWORK.register_module()
class FFN(BaseModule):
    """mmcv FFN: Sequential(Sequential(Linear, act, Dropout) x (num_fcs-1), Linear, Dropout), output
    added to `identity` (or the input).  Built by the transformer layer from the deprecated
    feedforward_channels / ffn_dropout / ffn_num_fcs kwargs
    (reference: modules/custom_base_transformer_layer.py:74-99,144-160)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2, f'num_fcs should be no less than 2. got {num_fcs}.'
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(X3Linear(in_channels, feedforward_channels),
                                     build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(X3Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward_fused(self, x, identity=None, post_norm=None):
        """Inference form on the MFMA Linear kernel: Linear+ReLU, then Linear + residual + the layer's
        following LayerNorm as one epilogue.  -> tensor, or None when the FFN is not the plain
        2-layer ReLU form / a shape has no fused kernel."""
        from .. import ext
        from .._lib import OccAmdUnsupported
        if not (self.num_fcs == 2 and self.add_identity and isinstance(self.layers[0][1], nn.ReLU)
                and isinstance(self.dropout_layer, nn.Identity)):
            return None
        fc1, fc2 = self.layers[0][0], self.layers[1]
        try:
            h = ext.linear(x.contiguous(), fc1.weight, fc1.bias, act='relu')
            return ext.linear(h, fc2.weight, fc2.bias,
                              residual=(x if identity is None else identity).contiguous(), ln=post_norm)
        except OccAmdUnsupported:
            return None

    def forward(self, x, identity=None, post_norm_train=None):
        """post_norm_train (a LayerNorm): the caller's following norm; -> (output, norm_applied) in that case."""
        if (self.num_fcs == 2 and isinstance(self.layers[0][1], nn.ReLU) and x.is_cuda
                and torch.is_grad_enabled()):
            # training: Linear + ReLU as one kernel (the mask for the backward is the output itself)
            h = self.layers[0][2](self.layers[0][0](x, act='relu'))
            out = self.layers[1](h)
            if post_norm_train is not None and self.add_identity and isinstance(self.dropout_layer, nn.Identity):
                from .. import ext
                res = x if identity is None else identity
                if ext.dropout_add_layernorm_ok(out, res, post_norm_train):
                    # the FFN's last Dropout + identity + the layer's following LayerNorm as one autograd node
                    drop = self.layers[2]
                    return ext.dropout_add_layernorm(out, res, post_norm_train, drop.p, drop.training), True
            out = self.layers[2](out)
        else:
            out = self.layers(x)
        if post_norm_train is not None:
            if not self.add_identity:
                return self.dropout_layer(out), False
            return (x if identity is None else identity) + self.dropout_layer(out), False
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


class ConvModule(nn.Module):
    """mmcv ConvModule in its default order conv -> norm -> act; bias defaults to `norm is None`;
    sub-module names `conv`, `bn`/`gn`/`ln`, `activate` (reference use:
    modules/transformer_occ.py:78-129)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), inplace=True, **kwargs):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        conv_cls = CONV_LAYERS.get((conv_cfg or dict(type='Conv2d'))['type'])
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                             dilation=dilation, groups=groups, bias=bias)
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act_cfg = dict(act_cfg)
            if act_cfg['type'] in ('ReLU', 'LeakyReLU'):
                act_cfg.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act_cfg)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.norm(x)
        if self.with_activation:
            x = self.activate(x)
        return x


@POSITIONAL_ENCODING.register_module()
class LearnedPositionalEncoding(BaseModule):
    """mmdet LearnedPositionalEncoding: pos[:, :F] = col_embed(x), pos[:, F:] = row_embed(y)."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats = num_feats
        self.row_num_embed = row_num_embed
        self.col_num_embed = col_num_embed
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x_embed = self.col_embed.weight[:w]
        y_embed = self.row_embed.weight[:h]
        pos = torch.cat((x_embed.unsqueeze(0).expand(h, w, -1), y_embed.unsqueeze(1).expand(h, w, -1)),
                        dim=-1).permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)
        return pos


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """mmdet CrossEntropyLoss (softmax branch): loss_weight * mean CE, optional avg_factor / weight."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0, **kwargs):
        super().__init__()
        assert not use_sigmoid and not use_mask, "only the softmax CE branch is on the occ path"
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.class_weight = class_weight
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, **kwargs):
        cw = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        loss = F.cross_entropy(cls_score, label, weight=cw, reduction='none',
                               ignore_index=self.ignore_index)
        if weight is not None:
            loss = loss * weight.float()
        if avg_factor is None:
            loss = loss.mean() if self.reduction == 'mean' else loss.sum()
        else:
            loss = loss.sum() / avg_factor
        return self.loss_weight * loss


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, **kwargs):
        loss = (pred - target).abs()
        if weight is not None:
            loss = loss * weight
        if avg_factor is None:
            loss = loss.mean() if self.reduction == 'mean' else loss.sum()
        else:
            loss = loss.sum() / avg_factor
        return self.loss_weight * loss
