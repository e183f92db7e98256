"""CPU oracle of the OccNet / BEVFormer-occ hot path — TEST INFRASTRUCTURE (see oracle/__init__.py).

Plain PyTorch fp32 restatement of the reference's module graph, un-fused and in the reference's own
order of operations, quirks included.  Each class cites the reference lines it
follows (paths relative to projects/mmdet3d_plugin/bevformer/).  Attribute names reproduce the
reference's state_dict key layout, so a product state_dict loads here
unchanged and vice versa.  Third-party pieces the reference pulls from mmcv/mmdet (FFN, LayerNorm
builder, ConvModule, LearnedPositionalEncoding) are restated from their published behaviour
(SURVEY.md Appendix B.3-B.5).
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .msda import multi_scale_deformable_attn_pytorch


def _xavier_uniform(m, bias=0.0):
    if m is not None and hasattr(m, 'weight') and m.weight is not None:
        nn.init.xavier_uniform_(m.weight, gain=1)
    if m is not None and getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def _constant(m, val, bias=0.0):
    nn.init.constant_(m.weight, val)
    if m.bias is not None:
        nn.init.constant_(m.bias, bias)


def _grid_init(num_heads, num_levels, num_points):
    # modules/spatial_cross_attention.py:256-265 / temporal_self_attention.py:110-121
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2).repeat(
        1, num_levels, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    return grid.view(-1)


class MSDeformableAttention3D(nn.Module):
    """modules/spatial_cross_attention.py:178-400."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.batch_first = batch_first
        self.output_proj = None
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels = embed_dims, num_levels
        self.num_heads, self.num_points = num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):  # :253-271
        _constant(self.sampling_offsets, 0.)
        self.sampling_offsets.bias.data = _grid_init(self.num_heads, self.num_levels, self.num_points)
        _constant(self.attention_weights, 0., 0.)
        _xavier_uniform(self.value_proj)
        _xavier_uniform(self.output_proj)  # None: no-op (quirk 7)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):  # :273-400
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == num_value
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(
            bs, num_query, self.num_heads, self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points)
        assert reference_points.shape[-1] == 2
        offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        bs, num_query, num_Z_anchors, xy = reference_points.shape
        reference_points = reference_points[:, :, None, None, None, :, :]
        sampling_offsets = sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        bs, num_query, num_heads, num_levels, num_all_points, xy = sampling_offsets.shape
        sampling_offsets = sampling_offsets.view(
            bs, num_query, num_heads, num_levels, num_all_points // num_Z_anchors, num_Z_anchors, xy)
        sampling_locations = reference_points + sampling_offsets
        sampling_locations = sampling_locations.view(
            bs, num_query, num_heads, num_levels, num_all_points, xy)
        self.last_sampling_locations = sampling_locations  # oracle-only tap (N_in counting)
        output = multi_scale_deformable_attn_pytorch(value, spatial_shapes, sampling_locations,
                                                     attention_weights)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return output


class SpatialCrossAttention(nn.Module):
    """modules/spatial_cross_attention.py:31-175."""

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False, deformable_attention=None, **kwargs):
        super().__init__()
        cfg = dict(deformable_attention or dict(type='MSDeformableAttention3D', embed_dims=256,
                                                num_levels=4))
        cfg.pop('type', None)
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.deformable_attention = MSDeformableAttention3D(**cfg)
        self.embed_dims, self.num_cams = embed_dims, num_cams
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        _xavier_uniform(self.output_proj)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag='encoder', **kwargs):  # :76-175
        if key is None:
            key = query
        if value is None:
            value = key
        if residual is None:
            inp_residual = query
            slots = torch.zeros_like(query)
        if query_pos is not None:
            query = query + query_pos
        bs, num_query, _ = query.size()
        D = reference_points_cam.size(3)
        indexes = []
        for i, mask_per_img in enumerate(bev_mask):
            indexes.append(mask_per_img[0].sum(-1).nonzero().squeeze(-1))   # batch 0's mask (quirk 1)
        max_len = max([len(each) for each in indexes])
        queries_rebatch = query.new_zeros([bs, self.num_cams, max_len, self.embed_dims])
        reference_points_rebatch = reference_points_cam.new_zeros([bs, self.num_cams, max_len, D, 2])
        for j in range(bs):
            for i, reference_points_per_img in enumerate(reference_points_cam):
                idx = indexes[i]
                queries_rebatch[j, i, :len(idx)] = query[j, idx]
                reference_points_rebatch[j, i, :len(idx)] = reference_points_per_img[j, idx]
        num_cams, l, bs, embed_dims = key.shape
        key = key.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        value = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        queries = self.deformable_attention(
            query=queries_rebatch.view(bs * self.num_cams, max_len, self.embed_dims), key=key,
            value=value,
            reference_points=reference_points_rebatch.view(bs * self.num_cams, max_len, D, 2),
            spatial_shapes=spatial_shapes, level_start_index=level_start_index).view(
                bs, self.num_cams, max_len, self.embed_dims)
        self.last_rows = [len(i) for i in indexes]  # oracle-only tap
        for j in range(bs):
            for i, idx in enumerate(indexes):
                slots[j, idx] += queries[j, i, :len(idx)]
        count = bev_mask.sum(-1) > 0
        count = count.permute(1, 2, 0).sum(-1)
        count = torch.clamp(count, min=1.0)
        slots = slots / count[..., None]
        slots = self.output_proj(slots)
        return self.dropout(slots) + inp_residual


class TemporalSelfAttention(nn.Module):
    """modules/temporal_self_attention.py:25-272."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2,
                 im2col_step=64, dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels = embed_dims, num_levels
        self.num_heads, self.num_points = num_heads, num_points
        self.num_bev_queue = num_bev_queue
        self.sampling_offsets = nn.Linear(embed_dims * num_bev_queue,
                                          num_bev_queue * num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims * num_bev_queue,
                                           num_bev_queue * num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):  # :107-126
        _constant(self.sampling_offsets, 0.)
        self.sampling_offsets.bias.data = _grid_init(
            self.num_heads, self.num_levels * self.num_bev_queue, self.num_points)
        _constant(self.attention_weights, 0., 0.)
        _xavier_uniform(self.value_proj)
        _xavier_uniform(self.output_proj)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag='decoder', **kwargs):  # :128-272
        if value is None:
            assert self.batch_first
            bs, len_bev, c = query.shape
            value = torch.stack([query, query], 1).reshape(bs * 2, len_bev, c)
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, embed_dims = query.shape
        _, num_value, _ = value.shape
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == num_value
        assert self.num_bev_queue == 2
        query = torch.cat([value[:bs], query], -1)      # quirk 5
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.reshape(bs * self.num_bev_queue, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(
            bs, num_query, self.num_heads, self.num_bev_queue, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(
            bs, num_query, self.num_heads, self.num_bev_queue, self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(
            bs, num_query, self.num_heads, self.num_bev_queue, self.num_levels, self.num_points)
        attention_weights = attention_weights.permute(0, 3, 1, 2, 4, 5).reshape(
            bs * self.num_bev_queue, num_query, self.num_heads, self.num_levels,
            self.num_points).contiguous()
        sampling_offsets = sampling_offsets.permute(0, 3, 1, 2, 4, 5, 6).reshape(
            bs * self.num_bev_queue, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        assert reference_points.shape[-1] == 2
        offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        sampling_locations = reference_points[:, :, None, :, None, :] \
            + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        output = multi_scale_deformable_attn_pytorch(value, spatial_shapes, sampling_locations,
                                                     attention_weights)
        output = output.permute(1, 2, 0)
        output = output.view(num_query, embed_dims, bs, self.num_bev_queue)
        output = output.mean(-1)
        output = output.permute(2, 0, 1)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return self.dropout(output) + identity


class FFN(nn.Module):
    """mmcv FFN as the reference builds it (modules/custom_base_transformer_layer.py:74-99,144-160;
    SURVEY.md Appendix B.3): Sequential(Sequential(Linear, ReLU, Dropout), Linear, Dropout) + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2
        layers, in_channels = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_channels),
                                        nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        if identity is None:
            identity = x
        return identity + out


_ATTN = dict(TemporalSelfAttention=TemporalSelfAttention, SpatialCrossAttention=SpatialCrossAttention,
             MSDeformableAttention3D=MSDeformableAttention3D)


class BEVFormerLayer(nn.Module):
    """modules/encoder.py:242-406 over modules/custom_base_transformer_layer.py:37-165."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 batch_first=True, **kwargs):
        super().__init__()
        assert len(operation_order) == 6
        assert set(operation_order) == set(['self_attn', 'norm', 'cross_attn', 'ffn'])
        self.batch_first = batch_first
        self.operation_order = operation_order
        self.pre_norm = operation_order[0] == 'norm'
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.attentions = nn.ModuleList()
        index = 0
        for name in operation_order:
            if name in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(dict(attn_cfgs[index]))
                cfg.setdefault('batch_first', self.batch_first)
                self.attentions.append(_ATTN[cfg.pop('type')](**cfg))
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count('ffn')):
            self.ffns.append(FFN(embed_dims=self.embed_dims, feedforward_channels=feedforward_channels,
                                 num_fcs=ffn_num_fcs, ffn_drop=ffn_dropout, act_cfg=act_cfg))
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(nn.LayerNorm(self.embed_dims))

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, ref_2d=None,
                ref_3d=None, bev_h=None, bev_w=None, reference_points_cam=None, mask=None,
                spatial_shapes=None, level_start_index=None, prev_bev=None, **kwargs):  # :287-406
        norm_index = attn_index = ffn_index = 0
        identity = query
        for layer in self.operation_order:
            if layer == 'self_attn':
                query = self.attentions[attn_index](
                    query, prev_bev, prev_bev, identity if self.pre_norm else None,
                    query_pos=bev_pos, key_pos=bev_pos, key_padding_mask=query_key_padding_mask,
                    reference_points=ref_2d,
                    spatial_shapes=torch.tensor([[bev_h, bev_w]], device=query.device),
                    level_start_index=torch.tensor([0], device=query.device), **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, reference_points=ref_3d,
                    reference_points_cam=reference_points_cam, mask=mask,
                    key_padding_mask=key_padding_mask, spatial_shapes=spatial_shapes,
                    level_start_index=level_start_index, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim='3d', bs=1, device='cpu',
                         dtype=torch.float):
    """modules/encoder.py:50-89."""
    if dim == '3d':
        zs = torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype, device=device
                            ).view(-1, 1, 1).expand(num_points_in_pillar, H, W) / Z
        xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device
                            ).view(1, 1, W).expand(num_points_in_pillar, H, W) / W
        ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device
                            ).view(1, H, 1).expand(num_points_in_pillar, H, W) / H
        ref_3d = torch.stack((xs, ys, zs), -1)
        ref_3d = ref_3d.permute(0, 3, 1, 2).flatten(2).permute(0, 2, 1)
        return ref_3d[None].repeat(bs, 1, 1, 1)
    ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device),
                                  torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device),
                                  indexing='ij')
    ref_y = ref_y.reshape(-1)[None] / H
    ref_x = ref_x.reshape(-1)[None] / W
    ref_2d = torch.stack((ref_x, ref_y), -1)
    return ref_2d.repeat(bs, 1, 1).unsqueeze(2)


def point_sampling(reference_points, pc_range, img_metas):
    """modules/encoder.py:92-151 (fp32)."""
    ego2lidar = img_metas[0]['ego2lidar']
    lidar2img = np.asarray([m['lidar2img'] for m in img_metas])
    lidar2img = reference_points.new_tensor(lidar2img)
    ego2lidar = reference_points.new_tensor(ego2lidar)
    reference_points = reference_points.clone()
    reference_points[..., 0:1] = reference_points[..., 0:1] * (pc_range[3] - pc_range[0]) + pc_range[0]
    reference_points[..., 1:2] = reference_points[..., 1:2] * (pc_range[4] - pc_range[1]) + pc_range[1]
    reference_points[..., 2:3] = reference_points[..., 2:3] * (pc_range[5] - pc_range[2]) + pc_range[2]
    reference_points = torch.cat((reference_points, torch.ones_like(reference_points[..., :1])), -1)
    reference_points = reference_points.permute(1, 0, 2, 3)
    D, B, num_query = reference_points.size()[:3]
    num_cam = lidar2img.size(1)
    reference_points = reference_points.view(D, B, 1, num_query, 4).repeat(1, 1, num_cam, 1, 1).unsqueeze(-1)
    lidar2img = lidar2img.view(1, B, num_cam, 1, 4, 4).repeat(D, 1, 1, num_query, 1, 1)
    ego2lidar = ego2lidar.view(1, 1, 1, 1, 4, 4).repeat(D, 1, num_cam, num_query, 1, 1)
    reference_points_cam = torch.matmul(torch.matmul(lidar2img.to(torch.float32),
                                                     ego2lidar.to(torch.float32)),
                                        reference_points.to(torch.float32)).squeeze(-1)
    eps = 1e-5
    bev_mask = (reference_points_cam[..., 2:3] > eps)
    reference_points_cam = reference_points_cam[..., 0:2] / torch.maximum(
        reference_points_cam[..., 2:3], torch.ones_like(reference_points_cam[..., 2:3]) * eps)
    reference_points_cam[..., 0] /= img_metas[0]['img_shape'][0][1]
    reference_points_cam[..., 1] /= img_metas[0]['img_shape'][0][0]
    bev_mask = (bev_mask & (reference_points_cam[..., 1:2] > 0.0)
                & (reference_points_cam[..., 1:2] < 1.0)
                & (reference_points_cam[..., 0:1] < 1.0)
                & (reference_points_cam[..., 0:1] > 0.0))
    bev_mask = torch.nan_to_num(bev_mask)
    reference_points_cam = reference_points_cam.permute(2, 1, 3, 0, 4)
    bev_mask = bev_mask.permute(2, 1, 3, 0, 4).squeeze(-1)
    return reference_points_cam, bev_mask


class BEVFormerEncoder(nn.Module):
    """modules/encoder.py:28-239."""

    def __init__(self, transformerlayers=None, num_layers=None, pc_range=None,
                 num_points_in_pillar=4, return_intermediate=False, dataset_type='nuscenes',
                 init_cfg=None, **kwargs):
        super().__init__()
        cfg = dict(transformerlayers)
        cfg.pop('type', None)
        self.num_layers = num_layers
        self.layers = nn.ModuleList([BEVFormerLayer(**copy.deepcopy(cfg)) for _ in range(num_layers)])
        self.embed_dims = self.layers[0].embed_dims
        self.return_intermediate = return_intermediate
        self.num_points_in_pillar = num_points_in_pillar
        self.pc_range = pc_range

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                **kwargs):  # :153-239
        output = bev_query
        intermediate = []
        ref_3d = get_reference_points(bev_h, bev_w, self.pc_range[5] - self.pc_range[2],
                                      self.num_points_in_pillar, dim='3d', bs=bev_query.size(1),
                                      device=bev_query.device, dtype=bev_query.dtype)
        ref_2d = get_reference_points(bev_h, bev_w, dim='2d', bs=bev_query.size(1),
                                      device=bev_query.device, dtype=bev_query.dtype)
        reference_points_cam, bev_mask = point_sampling(ref_3d, self.pc_range, kwargs['img_metas'])
        self.last_bev_mask, self.last_ref_cam = bev_mask, reference_points_cam  # oracle-only taps
        shift_ref_2d = ref_2d.clone()   # quirk 6: no ego-motion shift
        bev_query = bev_query.permute(1, 0, 2)
        bev_pos = bev_pos.permute(1, 0, 2)
        bs, len_bev, num_bev_level, _ = ref_2d.shape
        if prev_bev is not None:
            prev_bev = prev_bev.permute(1, 0, 2)
            prev_bev = torch.stack([prev_bev, bev_query], 1).reshape(bs * 2, len_bev, -1)
            hybird_ref_2d = torch.stack([shift_ref_2d, ref_2d], 1).reshape(bs * 2, len_bev, num_bev_level, 2)
        else:
            hybird_ref_2d = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, len_bev, num_bev_level, 2)
        for lid, layer in enumerate(self.layers):
            output = layer(bev_query, key, value, *args, bev_pos=bev_pos, ref_2d=hybird_ref_2d,
                           ref_3d=ref_3d, bev_h=bev_h, bev_w=bev_w, spatial_shapes=spatial_shapes,
                           level_start_index=level_start_index,
                           reference_points_cam=reference_points_cam, bev_mask=bev_mask,
                           prev_bev=prev_bev, **kwargs)
            bev_query = output
            if self.return_intermediate:
                intermediate.append(output)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output


class ConvModule3d(nn.Module):
    """mmcv ConvModule(conv_cfg=Conv3d, norm_cfg=BN3d, act_cfg=ReLU): conv -> bn -> relu, keys
    `conv.*`, `bn.*` (modules/transformer_occ.py:106-129)."""

    def __init__(self, cin, cout, bias=False):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)
        self.bn = nn.BatchNorm3d(cout)
        self.activate = nn.ReLU(inplace=True)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


def rotate_nearest(img, angle, center):
    """torchvision.transforms.functional.rotate(img (C,H,W), angle, center=center), nearest
    interpolation, no expand, fill 0 — restated from torchvision's affine-grid formulation
    (modules/transformer_occ.py:195-205 calls it with center=rotate_center=[100,100])."""
    C, H, W = img.shape
    cx, cy = center[0] - W * 0.5, center[1] - H * 0.5
    rot = math.radians(-angle)
    # inverse affine matrix of a pure rotation about (cx, cy) (torchvision _get_inverse_affine_matrix)
    a, b = math.cos(rot), math.sin(rot)
    m = [a, b, 0.0, -b, a, 0.0]
    m[2] += m[0] * (-cx) + m[1] * (-cy) + cx
    m[5] += m[3] * (-cx) + m[4] * (-cy) + cy
    theta = torch.tensor(m, dtype=img.dtype, device=img.device).view(1, 2, 3)
    d = 0.5
    base = torch.empty(1, H, W, 3, dtype=img.dtype, device=img.device)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + d, W * 0.5 + d - 1, steps=W, device=img.device))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + d, H * 0.5 + d - 1, steps=H, device=img.device).unsqueeze(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=img.dtype, device=img.device)
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2)
    return F.grid_sample(img[None], grid, mode='nearest', padding_mode='zeros', align_corners=False)[0]


class TransformerOcc(nn.Module):
    """modules/transformer_occ.py:26-321 (use_3d=True branch; the other two decoder variants are
    restated for completeness of the constructor contract)."""

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 decoder=None, embed_dims=256, rotate_prev_bev=True, use_shift=True, use_can_bus=True,
                 can_bus_norm=True, use_cams_embeds=True, use_3d=False, use_conv=False,
                 rotate_center=[100, 100], num_classes=18, out_dim=32, pillar_h=16,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BN'),
                 norm_cfg_3d=dict(type='BN3d'), **kwargs):
        super().__init__()
        enc = dict(encoder)
        enc.pop('type', None)
        self.encoder = BEVFormerEncoder(**enc)
        self.embed_dims = embed_dims
        self.num_feature_levels, self.num_cams = num_feature_levels, num_cams
        self.rotate_prev_bev = rotate_prev_bev
        self.use_cams_embeds = use_cams_embeds
        self.use_3d, self.use_conv = use_3d, use_conv
        self.pillar_h, self.out_dim = pillar_h, out_dim
        assert use_3d, "oracle restates the use_3d=True decoder (the only one any config selects)"
        self.middle_dims = embed_dims // pillar_h
        self.decoder = nn.Sequential(ConvModule3d(self.middle_dims, out_dim, bias=norm_cfg_3d is None),
                                     ConvModule3d(out_dim, out_dim, bias=norm_cfg_3d is None))
        self.predicter = nn.Sequential(nn.Linear(out_dim, out_dim * 2), nn.Softplus(),
                                       nn.Linear(out_dim * 2, num_classes))
        self.flow_predicter = nn.Sequential(nn.Linear(out_dim, out_dim * 2), nn.ReLU(),
                                            nn.Linear(out_dim * 2, 2))
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(num_cams, embed_dims))
        self.rotate_center = rotate_center

    def init_weights(self):  # :154-167
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)

    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512],
                         bev_pos=None, prev_bev=None, **kwargs):  # :171-242
        bs = mlvl_feats[0].size(0)
        bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        if prev_bev is not None:
            if prev_bev.shape[1] == bev_h * bev_w:
                prev_bev = prev_bev.permute(1, 0, 2)
            elif len(prev_bev.shape) == 4:
                prev_bev = prev_bev.view(bs, -1, bev_h * bev_w).permute(2, 0, 1)
            if self.rotate_prev_bev:
                prev_bev = prev_bev.clone()
                for i in range(bs):
                    rotation_angle = kwargs['img_metas'][i]['can_bus'][-1]
                    tmp = prev_bev[:, i].reshape(bev_h, bev_w, -1).permute(2, 0, 1)
                    tmp = rotate_nearest(tmp, rotation_angle, center=self.rotate_center)
                    tmp = tmp.permute(1, 2, 0).reshape(bev_h * bev_w, 1, -1)
                    prev_bev[:, i] = tmp[:, 0]
        feat_flatten, spatial_shapes = [], []
        for lvl, feat in enumerate(mlvl_feats):
            bs, num_cam, c, h, w = feat.shape
            feat = feat.flatten(3).permute(1, 0, 3, 2)
            if self.use_cams_embeds:
                feat = feat + self.cams_embeds[:, None, None, :].to(feat.dtype)
            feat = feat + self.level_embeds[None, None, lvl:lvl + 1, :].to(feat.dtype)
            spatial_shapes.append((h, w))
            feat_flatten.append(feat)
        feat_flatten = torch.cat(feat_flatten, 2)
        spatial_shapes = torch.as_tensor(spatial_shapes, dtype=torch.long, device=bev_pos.device)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)),
                                       spatial_shapes.prod(1).cumsum(0)[:-1]))
        feat_flatten = feat_flatten.permute(0, 2, 1, 3)
        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                            bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, prev_bev=prev_bev, **kwargs)

    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_h, bev_w,
                grid_length=[0.512, 0.512], bev_pos=None, reg_branches=None, cls_branches=None,
                prev_bev=None, **kwargs):  # :244-321
        bev_embed = self.get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w,
                                          grid_length=grid_length, bev_pos=bev_pos,
                                          prev_bev=prev_bev, **kwargs)
        bs = mlvl_feats[0].size(0)
        bev_embed = bev_embed.permute(0, 2, 1).view(bs, -1, bev_h, bev_w)
        outputs = self.decoder(bev_embed.view(bs, -1, self.pillar_h, bev_h, bev_w))
        outputs = outputs.permute(0, 4, 3, 2, 1)
        flow_pred = self.flow_predicter(outputs)
        occ_pred = self.predicter(outputs)
        return bev_embed, occ_pred, flow_pred


class LearnedPositionalEncoding(nn.Module):
    """mmdet LearnedPositionalEncoding."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x_embed = self.col_embed(torch.arange(w, device=mask.device))
        y_embed = self.row_embed(torch.arange(h, device=mask.device))
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)),
                        dim=-1).permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)
        return pos


class BEVFormerOccHead(nn.Module):
    """dense_heads/bevformer_occ_head.py:32-216."""

    def __init__(self, *args, with_box_refine=False, as_two_stage=False, transformer=None,
                 bbox_coder=None, num_cls_fcs=2, code_weights=None,
                 pc_range=[-40, -40, -1.0, 40, 40, 5.4], bev_h=30, bev_w=30, loss_occ=None,
                 loss_flow=None, use_mask=False, positional_encoding=None, **kwargs):
        super().__init__()
        self.bev_h, self.bev_w = bev_h, bev_w
        self.num_classes = kwargs['num_classes']
        self.use_mask = use_mask
        self.pc_range = pc_range
        self.real_w = pc_range[3] - pc_range[0]
        self.real_h = pc_range[4] - pc_range[1]
        self.loss_occ_weight = (loss_occ or {}).get('loss_weight', 1.0)
        self.loss_flow_weight = (loss_flow or {}).get('loss_weight', 1.0)
        pe = dict(positional_encoding)
        pe.pop('type', None)
        self.positional_encoding = LearnedPositionalEncoding(**pe)
        tr = dict(transformer)
        tr.pop('type', None)
        self.transformer = TransformerOcc(**tr)
        self.embed_dims = self.transformer.embed_dims
        self.bev_embedding = nn.Embedding(bev_h * bev_w, self.embed_dims)

    def init_weights(self):
        self.transformer.init_weights()

    def forward(self, mlvl_feats, img_metas, prev_bev=None, only_bev=False, test=False):  # :100-160
        bs = mlvl_feats[0].shape[0]
        dtype = mlvl_feats[0].dtype
        bev_queries = self.bev_embedding.weight.to(dtype)
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=bev_queries.device).to(dtype)
        bev_pos = self.positional_encoding(bev_mask).to(dtype)
        grid = (self.real_h / self.bev_h, self.real_w / self.bev_w)
        if only_bev:
            return self.transformer.get_bev_features(mlvl_feats, bev_queries, self.bev_h, self.bev_w,
                                                     grid_length=grid, bev_pos=bev_pos,
                                                     img_metas=img_metas, prev_bev=prev_bev)
        bev_embed, occ_outs, flow_outs = self.transformer(
            mlvl_feats, bev_queries, None, self.bev_h, self.bev_w, grid_length=grid, bev_pos=bev_pos,
            reg_branches=None, cls_branches=None, img_metas=img_metas, prev_bev=prev_bev)
        return {'bev_embed': bev_embed, 'occ': occ_outs, 'flow': flow_outs}

    def loss(self, voxel_semantics, voxel_flow, mask_camera, preds_dicts, gt_bboxes_ignore=None,
             img_metas=None):  # :163-196 (use_mask=False branch; mmdet CrossEntropyLoss / L1Loss, mean)
        occ = preds_dicts['occ'].reshape(-1, self.num_classes)
        flow = preds_dicts['flow'].reshape(-1, 2)
        loss_occ = self.loss_occ_weight * F.cross_entropy(occ, voxel_semantics.long().reshape(-1))
        loss_flow = self.loss_flow_weight * F.l1_loss(flow, voxel_flow.reshape(-1, 2))
        return dict(loss_occ=loss_occ, loss_flow=loss_flow)

    def get_occ(self, preds_dicts, img_metas=None, rescale=False):  # :199-216
        return preds_dicts['occ'].softmax(-1).argmax(-1), preds_dicts['flow']
