"""Registry names of the reference's *detection* branch.

No occupancy config instantiates them (`transformer_occ.py:21,160-161` only imports two of them for
isinstance checks), but §8(b) lists the names as part of the registry surface, so they resolve by string:

* `CustomMSDeformableAttention` (reference decoder.py:132-345) and `LearnedPositionalEncoding3D`
  (models/utils/positional_encoding.py:10-66) are small and are functional here — the attention runs on the
  same HIP operator as everything else (MultiScaleDeformableAttnFunction_fp32 -> ms_deform_attn_forward).
* `PerceptionTransformer` (transformer.py:26) and `DetectionTransformerDecoder` (decoder.py:52) are the DETR
  detection transformer: out of scope; the names parse (constructor keeps its cfg), calling them raises.
"""
import math
import warnings

import torch
import torch.nn as nn

from .._lib import OccAmdUnsupported
from .bricks import BaseModule, constant_init, xavier_init
from .functions import MultiScaleDeformableAttnFunction_fp32
from .registry import ATTENTION, POSITIONAL_ENCODING, TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE
from .spatial_cross_attention import _require_device


@ATTENTION.register_module()
class CustomMSDeformableAttention(BaseModule):
    """Deformable-DETR attention with output projection and residual (decoder-side variant)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        d = embed_dims // num_heads
        if d & (d - 1):
            warnings.warn("head sizes that are not a power of two run the generic gather kernel")
        self.norm_cfg, self.batch_first, self.fp16_enabled = norm_cfg, batch_first, False
        self.dropout = nn.Dropout(dropout)
        self.im2col_step, self.embed_dims = im2col_step, embed_dims
        self.num_levels, self.num_heads, self.num_points = num_levels, num_heads, num_points
        n = num_heads * num_levels * num_points
        self.sampling_offsets = nn.Linear(embed_dims, n * 2)
        self.attention_weights = nn.Linear(embed_dims, n)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        th = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        g = torch.stack([th.cos(), th.sin()], -1)
        g = (g / g.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        g = g * torch.arange(1, self.num_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        self.sampling_offsets.bias.data = g.reshape(-1)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag='decoder', residual=None, **kwargs):
        """query (num_query, bs, C) [or batch-first]; reference_points (bs, num_query, num_levels, 2 | 4)."""
        if identity is None:
            identity = query if residual is None else residual     # mmcv's deprecated `residual` alias
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        _require_device(value, 'CustomMSDeformableAttention')
        bs, nq, _ = query.shape
        nv = value.shape[1]
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == nv
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        M, L, P = self.num_heads, self.num_levels, self.num_points
        offs = self.sampling_offsets(query).view(bs, nq, M, L, P, 2)
        attn = self.attention_weights(query).view(bs, nq, M, L * P).softmax(-1).view(bs, nq, M, L, P)
        if reference_points.shape[-1] == 2:
            norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            loc = reference_points[:, :, None, :, None, :] + offs / norm[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] \
                + offs / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f'Last dim of reference_points must be 2 or 4, but get '
                             f'{reference_points.shape[-1]} instead.')
        out = MultiScaleDeformableAttnFunction_fp32.apply(value, spatial_shapes, level_start_index, loc,
                                                          attn, self.im2col_step)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


@POSITIONAL_ENCODING.register_module()
class LearnedPositionalEncoding3D(BaseModule):
    """Learned (col, row, height) embeddings -> (bs, 3*num_feats, l, h, w) for a (bs, l, h, w) mask."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, height_num_embed=50,
                 init_cfg=dict(type='Uniform', layer='Embedding')):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.height_embed = nn.Embedding(height_num_embed, num_feats)
        self.num_feats = num_feats
        self.row_num_embed, self.col_num_embed = row_num_embed, col_num_embed
        self.height_num_embed = height_num_embed
        for e in (self.row_embed, self.col_embed, self.height_embed):
            nn.init.uniform_(e.weight)

    def forward(self, mask):
        l, h, w = mask.shape[-3:]
        dev = mask.device
        x = self.col_embed(torch.arange(w, device=dev)).view(1, 1, w, -1).expand(l, h, w, -1)
        y = self.row_embed(torch.arange(h, device=dev)).view(1, h, 1, -1).expand(l, h, w, -1)
        z = self.height_embed(torch.arange(l, device=dev)).view(l, 1, 1, -1).expand(l, h, w, -1)
        pos = torch.cat((x, y, z), -1).permute(3, 0, 1, 2)
        return pos.unsqueeze(0).repeat(mask.shape[0], 1, 1, 1, 1)

    def __repr__(self):
        return (f'{self.__class__.__name__}(num_feats={self.num_feats}, row_num_embed={self.row_num_embed}, '
                f'col_num_embed={self.col_num_embed}, height_num_embed={self.height_num_embed})')


class _DetectionBranchName(nn.Module):
    """A detection-branch registry name: resolves and keeps its cfg; running it is out of scope."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.cfg = dict(kwargs)

    def init_weights(self):
        pass

    def forward(self, *args, **kwargs):
        raise OccAmdUnsupported(
            f"{type(self).__name__} belongs to the reference's DETR detection branch, which no occupancy "
            "config uses; only the name is part of this plugin's registry surface (SURVEY.md §2 row 9)")


@TRANSFORMER.register_module()
class PerceptionTransformer(_DetectionBranchName):
    pass


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetectionTransformerDecoder(_DetectionBranchName):
    pass
