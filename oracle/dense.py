"""Oracle for the encoder's dense ops (test infrastructure; see oracle/__init__.py).

`linear_chain` restates, on stock torch CPU ops in the caller's dtype (use float64), the sequences the
reference runs around its attention kernels:
  temporal_self_attention.py:197-209   query = cat([value[:bs], query + query_pos], -1) -> Linear
  spatial_cross_attention.py:173-175, temporal_self_attention.py:266-272   output_proj(x) + residual
  encoder.py:377-404 with mmcv FFN: Linear -> ReLU -> Linear, + identity,
  each followed by nn.LayerNorm(embed_dims) (operation_order 'norm' entries, post-norm).
"""
import torch
import torch.nn.functional as F


def linear_chain(a, weight, bias=None, a2=None, a2_add=None, act=None, residual=None, ln=None):
    """LayerNorm(residual + act(cat([a, a2 + a2_add], -1) @ weight^T + bias)); ln = (gamma, beta, eps)."""
    x = a
    if a2 is not None:
        x = torch.cat([a, a2 if a2_add is None else a2 + a2_add], -1)
    y = F.linear(x, weight, bias)
    if act == 'relu':
        y = F.relu(y)
    if residual is not None:
        y = y + residual
    if ln is not None:
        g, b, eps = ln
        y = F.layer_norm(y, (y.shape[-1],), g, b, eps)
    return y
