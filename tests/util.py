"""Shared builders for the parity tests: product (HIP) and oracle (CPU) models from one config."""
import copy

import numpy as np
import torch

from occnet_amd import synthetic

TOL = 1e-3   # north_star: outputs match the reference CPU path within 1e-3 (fp32)


def small_cfg(num_cams=6, bev=(40, 40), feat_shapes=((29, 50), (15, 25), (8, 13), (4, 7)),
              pillar_h=16, z_anchors=8, num_points=8, num_layers=2, img=(928, 1600)):
    """A reduced BEVFormer-occ geometry that the CPU oracle finishes in seconds."""
    return dict(name='small', num_cams=num_cams, img_h=img[0], img_w=img[1], feat_shapes=feat_shapes,
                bev_h=bev[0], bev_w=bev[1], pillar_h=pillar_h, num_points_in_pillar=z_anchors,
                embed_dims=256, pc_range=(-40.0, -40.0, -1.0, 40.0, 40.0, 5.4),
                num_points=num_points, num_layers=num_layers)


def head_cfg(g, num_classes=17):
    """pts_bbox_head config dict in the reference's format for geometry g."""
    dim = g['embed_dims']
    pcr = list(g['pc_range'])
    return dict(
        type='BEVFormerOccHead', pc_range=pcr, bev_h=g['bev_h'], bev_w=g['bev_w'],
        num_classes=num_classes, in_channels=dim, sync_cls_avg_factor=True, with_box_refine=True,
        as_two_stage=False, use_mask=False,
        loss_occ=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_flow=dict(type='L1Loss', loss_weight=0.25),
        transformer=dict(
            type='TransformerOcc', pillar_h=g['pillar_h'], num_classes=num_classes,
            num_cams=g['num_cams'], num_feature_levels=len(g['feat_shapes']),
            norm_cfg=dict(type='BN'), norm_cfg_3d=dict(type='BN3d'), use_3d=True, use_conv=False,
            rotate_prev_bev=True, use_shift=True, use_can_bus=True, embed_dims=dim,
            rotate_center=[g['bev_w'] // 2, g['bev_h'] // 2],
            encoder=dict(
                type='BEVFormerEncoder', num_layers=g.get('num_layers', 4), pc_range=pcr,
                num_points_in_pillar=g['num_points_in_pillar'], return_intermediate=False,
                transformerlayers=dict(
                    type='BEVFormerLayer',
                    attn_cfgs=[
                        dict(type='TemporalSelfAttention', embed_dims=dim, num_levels=1),
                        dict(type='SpatialCrossAttention', pc_range=pcr, num_cams=g['num_cams'],
                             deformable_attention=dict(type='MSDeformableAttention3D',
                                                       embed_dims=dim,
                                                       num_points=g.get('num_points', 8),
                                                       num_levels=len(g['feat_shapes'])),
                             embed_dims=dim)],
                    feedforward_channels=dim * 2, ffn_dropout=0.1,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))),
        positional_encoding=dict(type='LearnedPositionalEncoding', num_feats=dim // 2,
                                 row_num_embed=g['bev_h'], col_num_embed=g['bev_w']))


def randomize(module, seed=0):
    """Reference init, then make the sampling pattern query dependent and BN stats non-trivial
    (the reference init zeroes the offset/weight Linears)."""
    g = torch.Generator().manual_seed(seed)
    module.init_weights()
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith('sampling_offsets.weight') or name.endswith('attention_weights.weight'):
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
        for name, b in module.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
        for name, p in module.named_parameters():
            if '.bn.' in name and name.endswith('weight'):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            elif '.bn.' in name and name.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def build_pair(g, seed=0, device='cuda'):
    """-> (product head on `device`, oracle head on CPU) with identical weights, both eval()."""
    import oracle.model as om
    from occnet_amd.plugin import build_head
    cfg = head_cfg(g)
    prod = build_head(copy.deepcopy(cfg))
    randomize(prod, seed)
    ocfg = copy.deepcopy(cfg)
    ocfg.pop('type')
    ora = om.BEVFormerOccHead(**ocfg)
    missing, unexpected = ora.load_state_dict(prod.state_dict(), strict=True)
    prod = prod.to(device).eval()
    ora = ora.eval()
    return prod, ora


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
