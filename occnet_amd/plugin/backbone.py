"""Image backbone + neck named by the occ configs (`ResNet`, `FPN`), restated on stock torch.nn.

Out of the hot path's kernel scope ("backbone/neck stay stock PyTorch-ROCm
(MIOpen), no custom kernels"); they exist so `build_model(cfg.model)` works and end-to-end samples/s
(images -> voxels) can be measured.  Module/parameter names follow mmdet's (`conv1`, `bn1`,
`layer{1..4}.{i}.{conv,bn}{1,2,3}`, `downsample.{0,1}`; `lateral_convs.{i}.conv`, `fpn_convs.{i}.conv`)
so a reference checkpoint's `img_backbone.*` / `img_neck.*` keys load.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import cache_epoch
from .bricks import BaseModule, ConvModule
from .registry import BACKBONES, NECKS


def conv_bn_folded(x, conv, bn):
    """conv -> eval-mode BatchNorm as ONE convolution with autograd intact: y = conv(x, W * s) + t with
    s = gamma / sqrt(running_var + eps), t = beta - running_mean * s.  Identical function and identical gradients for
    W, gamma and beta (they flow through the small weight-side products), but no BatchNorm kernel ever touches the
    activation: the reference trains its ResNet with `norm_eval=True` (projects/configs/bevformer/
    bevformer_base_occ.py:55), and torch's eval-mode BN backward costs two passes over every activation
    (batch_norm_backward_reduce + elementwise: 19 ms of the 137 ms training step on MI355X)."""
    # the statistics are frozen (eval mode): 1/sqrt(var + eps) and mean/sqrt(var + eps) are constants, cached per
    # BatchNorm until a buffer is written (three small launches per convolution instead of six, fewer in backward)
    rstd, mean_rstd = _bn_fold_constants(bn)
    s = bn.weight * rstd
    w = conv.weight * s.view(-1, 1, 1, 1)
    b = torch.addcmul(bn.bias, bn.weight, mean_rstd, value=-1.0)
    if conv.bias is not None:
        b = b + conv.bias * s
    return F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)


def _bn_fold_constants(bn):
    """(1 / sqrt(var + eps), mean / sqrt(var + eps)) of an eval-mode BatchNorm, cached until a buffer is written."""
    key = (bn.running_mean._version, bn.running_var._version, bn.running_var.data_ptr(), cache_epoch())
    cached = getattr(bn, '_occ_fold', None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            rstd = torch.rsqrt(bn.running_var + bn.eps)
            cached = (key, rstd, bn.running_mean * rstd)
        object.__setattr__(bn, '_occ_fold', cached)
    return cached[1], cached[2]


class ConvBNActFunction(torch.autograd.Function):
    """conv -> eval-mode BatchNorm -> (+ residual) -> (ReLU) as ONE autograd node on bf16 channels_last activations: the
    training step's form of the inference plan's fused convolutions (the reference trains its ResNet with `norm_eval=True`,
    projects/configs/bevformer/bevformer_base_occ.py:55, under DDP: P/bevformer/apis/mmdet_train.py:71-79).
    Forward: the BatchNorm is folded into the weights (conv_bn_folded's arithmetic) and the convolution runs on this
    repository's 1x1 / 3x3 NHWC kernels with bias, residual and ReLU in their epilogues (other shapes: MIOpen + ONE fused
    tail launch) — under torch.autocast the same chain costs a convolution, a bias add, a residual add and a clamp, three
    more passes over the activation.  Backward: ONE pass makes the ReLU-masked gradient and the bias gradient
    (ext.bias_act_bwd_nhwc; ATen: threshold_backward + a bf16 column reduction + an add at the residual join), MIOpen's data /
    weight gradients follow, and the fold's chain rule gives the gradients of W, gamma and beta.  Same function and the same
    gradients as conv_bn_folded up to bf16 rounding (the tail adds in fp32 and rounds once instead of three times)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rstd, mean_rstd, conv_bias, residual, stride, padding, relu):
        from .. import ext
        O, I, kh, kw = weight.shape
        cl = torch.channels_last
        w16 = None
        one_launch = (gamma is not None and conv_bias is None and weight.dtype == torch.float32 and weight.is_contiguous()
                      and all(t.dtype == torch.float32 and t.is_contiguous() for t in (gamma, beta, rstd, mean_rstd)))
        if one_launch:
            wf, w16, b = ext.conv_bn_fold_fwd(weight, gamma, beta, rstd, mean_rstd)      # the whole fold: one launch
        elif gamma is not None:
            s = gamma * rstd
            wf = weight * s.view(-1, 1, 1, 1)
            b = torch.addcmul(beta, gamma, mean_rstd, value=-1.0)
            if conv_bias is not None:
                b = b + conv_bias * s
        else:
            wf = weight
            b = conv_bias if conv_bias is not None else weight.new_zeros(O)
        wf, b = wf.float(), b.float().contiguous()
        x16 = x if (x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=cl)) else \
            x.to(torch.bfloat16).contiguous(memory_format=cl)
        r16 = None
        if residual is not None:
            r16 = residual if (residual.dtype == torch.bfloat16 and residual.is_contiguous(memory_format=cl)) else \
                residual.to(torch.bfloat16).contiguous(memory_format=cl)
        if w16 is None:
            w16 = wf.to(torch.bfloat16).contiguous(memory_format=cl)
        stride, padding = tuple(stride), tuple(padding)
        y = None
        if (kh, kw) == (1, 1) and padding == (0, 0) and stride[0] == stride[1] and I % 32 == 0 and O % 32 == 0:
            y = ext.conv1x1_nhwc(x16, ext.conv1x1_pack_weight(wf.reshape(O, I)), b, residual=r16, relu=relu, stride=stride[0])
        elif ((kh, kw) == (3, 3) and padding == (1, 1) and stride in ((1, 1), (2, 2)) and I % 32 == 0 and O % 128 == 0
              and r16 is None):
            y = ext.conv3x3_nhwc(x16, ext.conv3x3_pack_weight(wf.contiguous()), b, O, relu=relu, stride=stride[0])
        if y is None:
            y = torch.ops.aten.convolution(x16, w16, None, stride, padding, (1, 1), False, (0, 0), 1)
            if not y.is_contiguous(memory_format=cl):
                y = y.contiguous(memory_format=cl)
            y = ext.bias_act_nhwc_(y, b, residual=r16, relu=relu)
        ctx.conv = (stride, padding, bool(relu), residual is not None and residual.dtype, one_launch)
        ctx.save_for_backward(x16, w16, y if relu else None, weight, gamma, rstd, mean_rstd, conv_bias)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from .. import ext
        x16, w16, y, weight, gamma, rstd, mean_rstd, conv_bias = ctx.saved_tensors
        stride, padding, relu, res_dtype, one_launch = ctx.conv
        s = None if (gamma is None or one_launch) else gamma * rstd
        cl = torch.channels_last
        if not (gy.dtype == torch.bfloat16 and gy.is_contiguous(memory_format=cl)):
            gy = gy.to(torch.bfloat16).contiguous(memory_format=cl)
        g, gb = ext.bias_act_bwd_nhwc(gy, y, relu=relu)
        need = ctx.needs_input_grad
        gx, gw, _ = torch.ops.aten.convolution_backward(g, x16, w16, None, stride, padding, (1, 1), False, (0, 0), 1,
                                                        (bool(need[0]), bool(need[1] or need[2]), False))
        dW = dgamma = dbeta = dcb = None
        if gw is not None and one_launch and need[1] and need[2]:
            dW, dgamma = ext.conv_bn_fold_bwd(gw, weight, gamma, rstd, mean_rstd, gb)    # the fold's chain rule: one launch
        elif gw is not None:
            gwf = gw.float()
            if s is not None:
                if need[1]:
                    dW = gwf * s.view(-1, 1, 1, 1)
                if need[2]:
                    dot = (gwf * weight).sum((1, 2, 3))
                    dgamma = rstd * dot - mean_rstd * gb
                    if conv_bias is not None:
                        dgamma = dgamma + rstd * conv_bias * gb
            elif need[1]:
                dW = gwf
        if gamma is not None:
            if need[3]:
                dbeta = gb
            if conv_bias is not None and need[6]:
                dcb = s * gb
        elif conv_bias is not None and need[6]:
            dcb = gb
        dres = None
        if need[7]:
            dres = g if res_dtype == torch.bfloat16 else g.to(res_dtype)
        return gx, dW, dgamma, dbeta, None, None, dcb, dres, None, None, None


def _fused_train_ok(x):
    """The fused training nodes serve the bf16-autocast backbone of the training step (device tensors only)."""
    return (Bottleneck.fused_train_nodes and torch.is_grad_enabled() and x.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype('cuda') == torch.bfloat16)


def conv_bn_act(x, conv, bn, relu=False, residual=None):
    """ConvBNActFunction on a Conv2d (+ eval-mode BatchNorm2d or None)."""
    if bn is not None:
        rstd, mean_rstd = _bn_fold_constants(bn)
        return ConvBNActFunction.apply(x, conv.weight, bn.weight, bn.bias, rstd, mean_rstd, conv.bias, residual,
                                       conv.stride, conv.padding, relu)
    return ConvBNActFunction.apply(x, conv.weight, None, None, None, None, conv.bias, residual, conv.stride, conv.padding,
                                   relu)


def _plain_conv(conv):
    return (tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and isinstance(conv.padding, tuple) and conv.weight.shape[0] % 8 == 0 and conv.weight.shape[0] <= 2048)


class Bottleneck(nn.Module):
    expansion = 4
    fold_eval_bn = True     # class-wide switch (OCC_TRAIN_FOLD_BN=0 clears it)
    fused_train_nodes = os.environ.get("OCC_TRAIN_FUSED_CONV", "1") != "0"    # ConvBNActFunction under bf16 autocast

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if (self.fold_eval_bn and torch.is_grad_enabled() and not self.bn1.training and x.is_cuda
                and self.bn1.affine and self.bn1.track_running_stats):
            if _fused_train_ok(x) and all(_plain_conv(c) for c in (self.conv1, self.conv2, self.conv3)) and \
                    (self.downsample is None or _plain_conv(self.downsample[0])):
                identity = x if self.downsample is None else conv_bn_act(x, self.downsample[0], self.downsample[1])
                out = conv_bn_act(x, self.conv1, self.bn1, relu=True)
                out = conv_bn_act(out, self.conv2, self.bn2, relu=True)
                return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=identity)
            identity = x if self.downsample is None else conv_bn_folded(x, self.downsample[0], self.downsample[1])
            out = self.relu(conv_bn_folded(x, self.conv1, self.bn1))
            out = self.relu(conv_bn_folded(out, self.conv2, self.bn2))
            out = conv_bn_folded(out, self.conv3, self.bn3)
            return self.relu(out + identity)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


if os.environ.get("OCC_TRAIN_FOLD_BN", "1") == "0":
    Bottleneck.fold_eval_bn = False


@BACKBONES.register_module()
class ResNet(BaseModule):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                 with_cp=False, pretrained=None, in_channels=3, base_channels=64, init_cfg=None,
                 **kwargs):
        super().__init__(init_cfg)
        if depth not in self.arch:
            raise KeyError(f'invalid depth {depth} for resnet (bottleneck depths only)')
        assert style == 'pytorch'
        self.depth, self.num_stages = depth, num_stages
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(base_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = base_channels
        self.res_layers = []
        for i, n in enumerate(self.arch[depth][:num_stages]):
            planes = base_channels * 2 ** i
            stride = 1 if i == 0 else 2
            blocks = []
            for j in range(n):
                ds = None
                if j == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if j == 0 else 1, ds))
                inplanes = planes * 4
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    use_frozen_prefix_plan = os.environ.get("OCC_TRAIN_FROZEN_PREFIX", "1") != "0"

    def _frozen_prefix(self, x):
        """Training with frozen_stages >= 1: the stem and the frozen stages need no autograd graph, so they run on
        the inference plan's kernels (whole stem + whole-bottleneck launches, bf16 NHWC) instead of MIOpen.
        -> (activation after the last frozen stage, number of stages done) or (None, 0)."""
        from .. import cache_epoch
        n = self.frozen_stages
        if not (self.use_frozen_prefix_plan and self.training and n >= 1 and x.is_cuda and x.dtype == torch.float32
                and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and not any(i < n for i in self.out_indices)):
            return None, 0
        plan = getattr(self, '_prefix_plan', None)
        if plan is None or plan.built_epoch != cache_epoch():
            plan = FusedInferenceBackbone(self, None, dtype=torch.bfloat16, prefix_stages=n)
            plan.built_epoch = cache_epoch()
            object.__setattr__(self, '_prefix_plan', plan)      # not a sub-module: owns folded copies
        if not plan._stem_fused:
            return None, 0
        return plan.forward_prefix(x.contiguous()), n

    def forward(self, x):
        x0, done = self._frozen_prefix(x)
        if x0 is not None:
            x = x0
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            if i < done:
                continue
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self


@NECKS.register_module()
class FPN(BaseModule):
    """1x1 laterals, top-down nearest upsample-add, 3x3 output convs, extra stride-2 levels."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'),
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert isinstance(in_channels, (list, tuple))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        self.backbone_end_level = self.num_ins if end_level in (-1, self.num_ins - 1) else end_level + 1
        self.start_level = start_level
        if add_extra_convs is True:
            add_extra_convs = 'on_input'
        assert add_extra_convs in (False, 'on_input', 'on_lateral', 'on_output')
        self.add_extra_convs = add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1,
                                             conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg,
                                             inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] \
                    if (i == 0 and self.add_extra_convs == 'on_input') else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1,
                                                 conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    @staticmethod
    def _run(cm, x):
        """A bias-only ConvModule of the neck: under the training step's bf16 autocast one fused autograd node
        (ConvBNActFunction: bias in the convolution's epilogue, bias gradient in one pass), otherwise the module itself."""
        if (_fused_train_ok(x) and not getattr(cm, 'with_norm', False) and not getattr(cm, 'with_activation', False)
                and isinstance(getattr(cm, 'conv', None), nn.Conv2d) and _plain_conv(cm.conv)):
            return conv_bn_act(x, cm.conv, None)
        return cm(x)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [self._run(conv, inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        n = len(laterals)
        for i in range(n - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(
                laterals[i], size=laterals[i - 1].shape[2:], **self.upsample_cfg)
        outs = [self._run(self.fpn_convs[i], laterals[i]) for i in range(n)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - n):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                if self.add_extra_convs == 'on_input':
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == 'on_lateral':
                    src = laterals[-1]
                else:
                    src = outs[-1]
                outs.append(self._run(self.fpn_convs[n], src))
                for i in range(n + 1, self.num_outs):
                    src = F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]
                    outs.append(self._run(self.fpn_convs[i], src))
        return tuple(outs)


class FusedInferenceBackbone(nn.Module):
    """Inference-time execution plan for ResNet + FPN: eval-mode BatchNorm folded into the preceding
    convolution (`fuse_conv_bn_weights`), bf16, channels_last (NHWC) memory end to end, so the FPN outputs are
    already in the (Cam, H, W, C) layout the hot path reads.  With `hip_tail` (bf16, default) every layer of the
    configs' ResNet-50 + FPN runs on this repository's kernels: whole stem (ext.stem_conv7x7_pool), whole
    64-mid-channel bottlenecks (ext.bottleneck64_nhwc), 1x1 / 3x3 convolutions with bias, residual and ReLU fused
    (ext.conv1x1_nhwc incl. the FPN top-down step, ext.conv3x3_nhwc); convolutions of other shapes fall back to
    MIOpen + one fused bias/residual/ReLU launch.  `hip_tail=False` keeps stock torch ops (any dtype);
    `fused_ops=True` would issue MIOpen's fused conv+bias+ReLU — measured on MI355X / ROCm 7.2 that falls back to
    MIOpen's naive bf16 NHWC kernel (1.7 s per forward), so it is off.  Built from the live modules' parameters
    (it owns folded COPIES: rebuild after changing weights).  The backbone is outside SURVEY.md §8's hand-written
    scope; these kernels exist because end-to-end samples/s (images -> voxels) is the headline metric."""

    def __init__(self, backbone, neck, dtype=torch.bfloat16, fused_ops=False, hip_tail=True,
                 fused_bottleneck=True, prefix_stages=None):
        super().__init__()
        # prefix_stages = k: fold only the stem and the first k stages, no neck (forward_prefix: the frozen part of
        # a training step)
        self.prefix_stages = prefix_stages
        # hip_tail: bias + (residual) + ReLU after each convolution as ONE in-place HIP launch
        # (occ_bias_act_nhwc_bf16) instead of PyTorch's add / add_ / relu_ launches (bf16 only)
        self.hip_tail = hip_tail and dtype == torch.bfloat16
        self.use_graph = False      # set True to replay the plan as one hipGraph per input shape
        self._graphs = {}
        from torch.nn.utils.fusion import fuse_conv_bn_weights
        assert not backbone.training or backbone.norm_eval, "folding BN needs eval-mode statistics"
        self.dtype, self.fused_ops = dtype, fused_ops
        self.out_indices = backbone.out_indices
        self._convs = []

        def fold(conv, bn):
            w, b = fuse_conv_bn_weights(conv.weight, conv.bias, bn.running_mean, bn.running_var,
                                        bn.eps, bn.weight, bn.bias)
            return self._add(w, b, conv)

        self.stem = fold(backbone.conv1, backbone.bn1)
        # whole stem (7x7/s2 convolution + bias + ReLU + 3x3/s2 max pooling) as one kernel reading the fp32 NCHW
        # images directly
        c1, mp = backbone.conv1, backbone.maxpool
        pool_ok = (mp.kernel_size, mp.stride, mp.padding) in ((3, 2, 1), ((3, 3), (2, 2), (1, 1)))
        sw = getattr(self, f'w{self.stem}')
        self._stem_fused = (self.hip_tail and sw.is_cuda and tuple(sw.shape) == (64, 3, 7, 7) and pool_ok
                            and tuple(c1.stride) == (2, 2) and tuple(c1.padding) == (3, 3)
                            and tuple(c1.dilation) == (1, 1) and c1.groups == 1
                            and getattr(mp, 'dilation', 1) in (1, (1, 1)) and not getattr(mp, 'ceil_mode', False))
        if self._stem_fused:
            from .. import ext
            self.register_buffer('stem_frag', ext.stem_pack_weight(sw), persistent=False)
        self.stages = []
        for name in (backbone.res_layers if prefix_stages is None else backbone.res_layers[:prefix_stages]):
            blocks = []
            for blk in getattr(backbone, name):
                ds = None if blk.downsample is None else fold(blk.downsample[0], blk.downsample[1])
                blocks.append((fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2),
                               fold(blk.conv3, blk.bn3), ds))
            self.stages.append(blocks)
        # whole-bottleneck kernel for the 64-mid-channel stride-1 blocks (ResNet-50 layer1: every layer of those
        # blocks is HBM-bound at stride 4, the fused kernel keeps the 64-channel intermediates in LDS)
        self._bneck = {}
        if fused_bottleneck and self.hip_tail and getattr(self, f'w{self.stem}').is_cuda:
            from .. import ext
            for si, blocks in enumerate(self.stages):
                for bi, (c1, c2, c3, ds) in enumerate(blocks):
                    w1, w2, w3 = (getattr(self, f'w{i}') for i in (c1, c2, c3))
                    ok = (tuple(w2.shape) == (64, 64, 3, 3) and tuple(w3.shape[:2]) == (256, 64)
                          and self._convs[c2][0] == (1, 1) and self._convs[c2][1] == (1, 1)
                          and ((ds is None and w1.shape[1] == 256) or
                               (ds is not None and w1.shape[1] == 64 and self._convs[ds][0] == (1, 1))))
                    if not ok:
                        continue
                    pk = ext.bottleneck64_pack(
                        w1, getattr(self, f'b{c1}'), w2, getattr(self, f'b{c2}'), w3, getattr(self, f'b{c3}'),
                        None if ds is None else getattr(self, f'w{ds}'),
                        None if ds is None else getattr(self, f'b{ds}'))
                    for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3'):
                        self.register_buffer(f'k{si}_{bi}_{k}', pk[k], persistent=False)
                    self._bneck[(si, bi)] = (pk['cin'], pk['ds'])
        self.neck = neck
        if neck is None:
            self.laterals, self.fpn = [], []
            return
        self.laterals = [self._add(m.conv.weight, m.conv.bias, m.conv) for m in neck.lateral_convs]
        self.fpn = [self._add(m.conv.weight, m.conv.bias, m.conv) for m in neck.fpn_convs]
        for m in list(neck.lateral_convs) + list(neck.fpn_convs):
            assert not m.with_norm and not m.with_activation, "FPN ConvModules with norm/act not folded"

    def _add(self, w, b, conv):
        w = w.detach().to(self.dtype).contiguous(memory_format=torch.channels_last)
        if b is None:
            b = torch.zeros(w.shape[0], device=w.device)
        b = b.detach().float() if self.hip_tail else b.detach().to(self.dtype)
        idx = len(self._convs)
        self.register_buffer(f'w{idx}', w, persistent=False)
        self.register_buffer(f'b{idx}', b, persistent=False)
        # 1x1 convolutions become the fused bf16 GEMM (occ_conv1x1_nhwc_bf16): (Cout, Cin) weight matrix
        gemm = (self.hip_tail and tuple(conv.kernel_size) == (1, 1) and tuple(conv.padding) == (0, 0)
                and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.stride[0] == conv.stride[1]
                and w.shape[1] % 32 == 0 and w.shape[0] % 32 == 0 and w.is_cuda)
        if gemm:
            from .. import ext
            self.register_buffer(f'm{idx}', ext.conv1x1_pack_weight(w.reshape(w.shape[0], w.shape[1])),
                                 persistent=False)
        # 3x3 convolutions (stride 1 or 2) with >= 128 output channels: own implicit-GEMM kernel (bias+ReLU fused)
        c3 = (self.hip_tail and tuple(conv.kernel_size) == (3, 3) and tuple(conv.padding) == (1, 1)
              and tuple(conv.stride) in ((1, 1), (2, 2)) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
              and w.shape[1] % 32 == 0 and w.shape[0] % 128 == 0 and w.is_cuda)
        if c3:
            from .. import ext
            self.register_buffer(f'p{idx}', ext.conv3x3_pack_weight(w.float().contiguous()), persistent=False)
        self._convs.append((conv.stride, conv.padding, conv.dilation, conv.groups))
        self._gemm = getattr(self, '_gemm', {})
        self._gemm[idx] = gemm
        self._c3 = getattr(self, '_c3', {})
        self._c3[idx] = c3
        return idx

    def _conv(self, i, x, relu=False, add=None, amax=None):
        """amax (FPN output convolutions only): 8 device words the 3x3 kernel folds max|out| into; any other route clears
        self._amax_ok — the maps then carry no maximum and their consumer measures it (ext.value_range_scale)."""
        w, b = getattr(self, f'w{i}'), getattr(self, f'b{i}')
        if amax is not None and not (self._c3.get(i) and add is None and not self.fused_ops
                                     and x.is_contiguous(memory_format=torch.channels_last)):
            self._amax_ok = False
        s, p, d, g = self._convs[i]
        if self.fused_ops and add is not None:
            return torch.miopen_convolution_add_relu(x, w, add, 1.0, b.to(w.dtype), s, p, d, g)
        if self.fused_ops and relu:
            return torch.miopen_convolution_relu(x, w, b.to(w.dtype), s, p, d, g)
        if self._gemm.get(i) and x.is_contiguous(memory_format=torch.channels_last) and \
                (add is None or add.is_contiguous(memory_format=torch.channels_last)):
            from .. import ext
            return ext.conv1x1_nhwc(x, getattr(self, f'm{i}'), b, residual=add,
                                    relu=relu or add is not None, stride=s[0])
        if self._c3.get(i) and add is None and x.is_contiguous(memory_format=torch.channels_last):
            from .. import ext
            return ext.conv3x3_nhwc(x, getattr(self, f'p{i}'), b, w.shape[0], relu=relu, stride=s[0],
                                    amax=amax if getattr(self, '_amax_ok', False) else None)
        if self.hip_tail and w.shape[0] % 8 == 0:
            from .. import ext
            y = F.conv2d(x, w, None, s, p, d, g)
            if y.is_contiguous(memory_format=torch.channels_last) and \
                    (add is None or add.is_contiguous(memory_format=torch.channels_last)):
                return ext.bias_act_nhwc_(y, b, residual=add, relu=relu or add is not None)
            y = y + b.to(y.dtype).view(1, -1, 1, 1)
        else:
            y = F.conv2d(x, w, b.to(w.dtype), s, p, d, g)
        if add is not None:
            y = y.add_(add)
        return y.relu_() if (relu or add is not None) else y

    @torch.no_grad()
    def forward(self, x):
        """x (N, 3, H, W) any float dtype -> tuple of FPN maps (N, C, h, w), dtype self.dtype, NHWC.
        With `use_graph` the whole plan (≈ 120 short launches) is captured into one hipGraph per input
        shape after two eager warm-up calls (MIOpen's find must not run under capture) and replayed; the
        returned maps are then the graph's static output buffers, valid until the next call."""
        if not getattr(self, 'use_graph', False) or not x.is_cuda:
            return self._forward_eager(x)
        key = (tuple(x.shape), x.dtype, str(x.device))
        st = self._graphs.setdefault(key, dict(calls=0))
        st['calls'] += 1
        if st['calls'] <= 2:
            return self._forward_eager(x)
        if 'graph' not in st:
            st['in'] = x.clone()
            torch.cuda.synchronize(x.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st['out'] = self._forward_eager(st['in'])
            st['graph'] = g
        st['in'].copy_(x)
        st['graph'].replay()
        return st['out']

    def forward_u8(self, x_u8, mean, std, to_rgb=False, size_divisor=32):
        """Raw camera images in: x_u8 (N, Hs, Ws, 3) uint8 HWC on the device; normalise + pad happen inside the stem
        kernel (ext.stem_conv7x7_pool_u8).  -> (FPN maps, padded (H, W)).  Needs the fused HIP stem."""
        from .. import ext
        from .._lib import OccAmdUnsupported
        if not getattr(self, '_stem_fused', False):
            raise OccAmdUnsupported("forward_u8 needs the fused stem kernel (bf16 plan, 7x7/s2 stem + 3x3/s2 pool)")
        x, hw = ext.stem_conv7x7_pool_u8(x_u8, self.stem_frag, getattr(self, f'b{self.stem}'), mean, std,
                                         to_rgb=to_rgb, size_divisor=size_divisor)
        return self._forward_stages(x), hw

    def _forward_eager(self, x):
        sw = getattr(self, f'w{self.stem}')
        if getattr(self, '_stem_fused', False) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
            from .. import ext
            x = ext.stem_conv7x7_pool(x, self.stem_frag, getattr(self, f'b{self.stem}'))
            return self._forward_stages(x)
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        if self.hip_tail and sw.shape[0] % 8 == 0 and x.is_cuda:
            # stem tail (bias + ReLU + 3x3/s2 max pooling) as one pass over the raw convolution output
            from .. import ext
            s, p, d, g = self._convs[self.stem]
            y = F.conv2d(x, sw, None, s, p, d, g)
            if y.is_contiguous(memory_format=torch.channels_last):
                x = ext.bias_relu_maxpool_nhwc(y, getattr(self, f'b{self.stem}'))
            else:
                x = F.max_pool2d((y + getattr(self, f'b{self.stem}').to(y.dtype).view(1, -1, 1, 1)).relu_(),
                                 kernel_size=3, stride=2, padding=1)
        else:
            x = F.max_pool2d(self._conv(self.stem, x, relu=True), kernel_size=3, stride=2, padding=1)
        return self._forward_stages(x)

    def _run_stage(self, si, x):
        for bi, (c1, c2, c3, ds) in enumerate(self.stages[si]):
            if (si, bi) in self._bneck and x.is_contiguous(memory_format=torch.channels_last):
                from .. import ext
                cin, has_ds = self._bneck[(si, bi)]
                pk = {k: getattr(self, f'k{si}_{bi}_{k}') for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')}
                pk.update(cin=cin, ds=has_ds)
                x = ext.bottleneck64_nhwc(x, pk)
                continue
            identity = x if ds is None else self._conv(ds, x)
            y = self._conv(c2, self._conv(c1, x, relu=True), relu=True)
            x = self._conv(c3, y, add=identity)
        return x

    @torch.no_grad()
    def forward_prefix(self, x):
        """x (N, 3, H, W) fp32 contiguous -> activation after the folded stages, (N, C, h, w) bf16 channels_last."""
        from .. import ext
        x = ext.stem_conv7x7_pool(x, self.stem_frag, getattr(self, f'b{self.stem}'))
        for si in range(len(self.stages)):
            x = self._run_stage(si, x)
        return x

    def _forward_stages(self, x):
        feats = []
        for si in range(len(self.stages)):
            x = self._run_stage(si, x)
            if si in self.out_indices:
                feats.append(x)
        nk = self.neck
        inputs = feats
        n = len(self.laterals)
        lat = [None] * n
        nearest = nk.upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in nk.upsample_cfg
        for i in range(n - 1, -1, -1):      # top-down: lateral 1x1 conv + nearest x2 upsample of the coarser level
            xin = inputs[i + nk.start_level]
            li = self.laterals[i]
            up = lat[i + 1] if i + 1 < n else None
            if up is not None and nearest and self._gemm.get(li) and xin.shape[2] == 2 * up.shape[2] \
                    and xin.shape[3] == 2 * up.shape[3] and xin.is_contiguous(memory_format=torch.channels_last):
                from .. import ext       # one launch: the upsampled coarser lateral is the GEMM's residual
                lat[i] = ext.conv1x1_nhwc(xin, getattr(self, f'm{li}'), getattr(self, f'b{li}'), residual=up,
                                          relu=False, residual_upsample2=True)
                continue
            lat[i] = self._conv(li, xin)
            if up is not None:
                lat[i] = lat[i] + F.interpolate(up, size=lat[i].shape[2:], **nk.upsample_cfg)
        # the output convolutions fold max|out| into 8 device words while they store the maps: the fp16 range scale of the SCA
        # value rows needs max|x| over exactly these maps (csrc/value_range.hip), and a separate pass over them costs 52 us
        amax = None
        self._amax_ok = bool(lat[0].is_cuda and self.dtype == torch.bfloat16)
        if self._amax_ok:
            from .. import ext
            amax = ext.new_absmax_words(lat[0].device)
        outs = [self._conv(self.fpn[i], lat[i], amax=amax) for i in range(n)]
        if nk.num_outs > n:
            if not nk.add_extra_convs:
                for _ in range(nk.num_outs - n):                    # a subset of outs[-1]: the maximum still bounds it
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                if nk.add_extra_convs == 'on_input':
                    src = inputs[nk.backbone_end_level - 1]
                elif nk.add_extra_convs == 'on_lateral':
                    src = lat[-1]
                else:
                    src = outs[-1]
                outs.append(self._conv(self.fpn[n], src, amax=amax))
                for i in range(n + 1, nk.num_outs):
                    src = F.relu(outs[-1]) if nk.relu_before_extra_convs else outs[-1]
                    outs.append(self._conv(self.fpn[i], src, amax=amax))
        if self._amax_ok:
            from .. import ext
            for o in outs:
                ext.attach_absmax(o, amax)    # rides on the tensor OBJECTS (with their version counter): a consumer that reshapes them re-attaches it
        return tuple(outs)
