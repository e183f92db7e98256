"""Oracle for the lifter + Conv3d decoder + heads (test infrastructure; see oracle/__init__.py).

Restates, on stock torch CPU ops in the caller's dtype (use float64 for kernel-level checks), what
the reference does at projects/mmdet3d_plugin/bevformer/modules/transformer_occ.py:
  :305-307  lifter: (bs, H*W, C) -> permute -> view (bs, C/Z, Z, H, W)
  :106-126  ConvModule(Conv3d k3 p1 no-bias, BN3d, ReLU) — mmcv order conv -> norm -> act
  :308      permute(0, 4, 3, 2, 1)
  :132-141, :318-319  predicter (Linear, Softplus, Linear), flow_predicter (Linear, ReLU, Linear)
"""
import torch
import torch.nn.functional as F


def lifter(bev, Z, H, W):
    """bev (bs, H*W, C) -> (bs, C//Z, Z, H, W): channel c is feature c // Z at height c % Z."""
    bs = bev.shape[0]
    return bev.permute(0, 2, 1).reshape(bs, -1, H, W).reshape(bs, -1, Z, H, W)


def conv3d_bn_relu(x, weight, bn_weight, bn_bias, running_mean, running_var, eps=1e-5, relu=True,
                   conv_bias=None):
    """x (bs, Cin, Z, H, W) -> (bs, Cout, Z, H, W): Conv3d(k3, s1, p1) -> BatchNorm3d(eval) -> ReLU."""
    y = F.conv3d(x, weight, conv_bias, stride=1, padding=1)
    y = F.batch_norm(y, running_mean, running_var, bn_weight, bn_bias, training=False, eps=eps)
    return F.relu(y) if relu else y


def heads(feat, w1o, b1o, w2o, b2o, w1f, b1f, w2f, b2f):
    """feat (..., C) -> (occ (..., ncls), flow (..., 2))."""
    occ = F.linear(F.softplus(F.linear(feat, w1o, b1o)), w2o, b2o)
    flow = F.linear(F.relu(F.linear(feat, w1f, b1f)), w2f, b2f)
    return occ, flow
