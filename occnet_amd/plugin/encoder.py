"""BEVFormerEncoder / BEVFormerLayer / MyCustomBaseTransformerLayer on the MI355X path.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/modules/encoder.py and
custom_base_transformer_layer.py: same registry names, constructor kwargs (including the deprecated
feedforward_channels / ffn_dropout / ffn_num_fcs -> ffn_cfgs mapping), module attribute names
(`layers`, `attentions`, `ffns`, `norms`) and forward contracts.

What differs is where the work runs: the pillar projection (`point_sampling`, reference
encoder.py:92-151) is one HIP kernel that also emits the per-query camera-visibility word every
SpatialCrossAttention layer would otherwise rebuild with nonzero(); the encoder computes it once and
hands it (plus a cache-friendly query processing order) to all layers through kwargs.
"""
import copy
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import cache_epoch, ext
from ..synthetic import bev_tile_order
from .bricks import FFN, BaseModule, ModuleList, build_norm_layer
from .registry import (TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_attention,
                       build_feedforward_network, build_transformer_layer)
from .spatial_cross_attention import _require_device


# the SCA value projections of ALL layers depend on the camera features only: they are launched before the first layer
# (LazyFeatures.prefetch: ONE stacked launch, on the caller's stream like everything else the library enqueues)

# (Rounds 4-5 built and measured a ROW PIPELINE on top of the chain kernels — the BEV queries cut into K row bands, every
# layer walked band by band on K streams so that the TSA gather / program A / SCA gather / program B of different bands
# co-run — including a native launcher issuing the whole banded encoder from one C-ABI call.  Same box, round 5: 2.555 /
# 2.548 ms per hot-path step with 2 bands against 2.541 / 2.522 for this path (profiles/r05_c1_rowpipe_ab.txt): the
# kernels slow each other down by what the overlap gains.  Rejected; sources, tests and the hazard it exposed under
# tools_dev/lab/row_pipeline/ (README there))


@TRANSFORMER_LAYER.register_module()
class MyCustomBaseTransformerLayer(BaseModule):
    """Generic transformer layer container: attentions / FFNs / norms built from cfg and applied in
    `operation_order` (reference: custom_base_transformer_layer.py:37-262)."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=True,
                 **kwargs):
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for ori_name, new_name in dict(feedforward_channels='feedforward_channels',
                                       ffn_dropout='ffn_drop', ffn_num_fcs='num_fcs').items():
            if ori_name in kwargs:
                ffn_cfgs[new_name] = kwargs[ori_name]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        known = {'self_attn', 'norm', 'ffn', 'cross_attn'}
        assert set(operation_order) & known == set(operation_order), \
            f'operation_order of {self.__class__.__name__} may only contain {sorted(known)}'
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs), \
                f'{len(attn_cfgs)} attention configs for {num_attn} attentions in {operation_order}'
            attn_cfgs = [copy.deepcopy(dict(c)) for c in attn_cfgs]
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        index = 0
        for operation_name in operation_order:
            if operation_name in ('self_attn', 'cross_attn'):
                if 'batch_first' in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]['batch_first']
                else:
                    attn_cfgs[index]['batch_first'] = self.batch_first
                attention = build_attention(attn_cfgs[index])
                attention.operation_name = operation_name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for ffn_index in range(num_ffns):
            cfg = dict(ffn_cfgs[ffn_index])
            cfg.setdefault('type', 'FFN')
            if 'embed_dims' not in cfg:
                cfg['embed_dims'] = self.embed_dims
            else:
                assert cfg['embed_dims'] == self.embed_dims
            self.ffns.append(build_feedforward_network(cfg))
        self.norms = ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        else:
            assert len(attn_masks) == self.num_attn
        for layer in self.operation_order:
            if layer == 'self_attn':
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@TRANSFORMER_LAYER.register_module()
class BEVFormerLayer(MyCustomBaseTransformerLayer):
    """One encoder layer: TSA -> LN -> SCA -> LN -> FFN -> LN in the base config
    (reference: encoder.py:242-406)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        self.fp16_enabled = False
        self.use_fused = True     # flip to force the op-by-op (reference-shaped) execution
        # training: dropout + residual + LayerNorm after each block as one autograd node (OCC_TRAIN_FUSED_LN=0: ATen ops)
        self.fused_train_tail = os.environ.get("OCC_TRAIN_FUSED_LN", "1") != "0"
        assert len(operation_order) == 6
        assert set(operation_order) == set(['self_attn', 'norm', 'cross_attn', 'ffn'])

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, ref_2d=None,
                ref_3d=None, bev_h=None, bev_w=None, reference_points_cam=None, mask=None,
                spatial_shapes=None, level_start_index=None, prev_bev=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
            warnings.warn(f'Use same attn_mask in all attentions in {self.__class__.__name__} ')
        else:
            assert len(attn_masks) == self.num_attn
        tsa_shapes = kwargs.pop('tsa_spatial_shapes', None)
        tsa_start = kwargs.pop('tsa_level_start_index', None)
        # inference: op + residual + the LayerNorm that follows it run as MFMA-Linear epilogues
        fuse = (self.use_fused and not self.pre_norm and not self.training and query.is_cuda
                and query.dtype == torch.float32 and query_key_padding_mask is None
                and key_padding_mask is None and all(m is None for m in attn_masks)
                and not (torch.is_grad_enabled() and (query.requires_grad or any(
                    p.requires_grad for p in self.parameters()))))
        # training (autograd): dropout + residual + the LayerNorm that follows an attention / FFN block as ONE node
        # (ext.DropoutAddLayerNormFunction); the block reports whether it applied the norm
        fuse_train = (self.fused_train_tail and not fuse and not self.pre_norm and query.is_cuda
                      and query.dtype == torch.float32 and torch.is_grad_enabled())
        ops = self.operation_order
        i = 0
        while i < len(ops):
            layer = ops[i]
            post_norm = None
            if fuse and i + 1 < len(ops) and ops[i + 1] == 'norm' \
                    and isinstance(self.norms[norm_index], nn.LayerNorm):
                post_norm = self.norms[norm_index]
            tail = {}
            if fuse_train and i + 1 < len(ops) and ops[i + 1] == 'norm' and isinstance(self.norms[norm_index], nn.LayerNorm):
                tail = dict(post_norm_train=self.norms[norm_index])
            if layer == 'self_attn':   # temporal self attention: BEV plane is its own single level
                attn = self.attentions[attn_index]
                out = None
                if post_norm is not None and hasattr(attn, 'forward_fused'):
                    out = attn.forward_fused(query, prev_bev, query_pos=bev_pos, reference_points=ref_2d,
                                             bev_h=bev_h, bev_w=bev_w,
                                             bev_order=kwargs.get('bev_order'), post_norm=post_norm)
                if out is not None:
                    query = out
                    norm_index += 1
                    i += 1
                else:
                    if tsa_shapes is None:
                        tsa_shapes = torch.tensor([[bev_h, bev_w]], device=query.device)
                        tsa_start = torch.tensor([0], device=query.device)
                    query = attn(
                        query, prev_bev, prev_bev, identity if self.pre_norm else None,
                        query_pos=bev_pos, key_pos=bev_pos, attn_mask=attn_masks[attn_index],
                        key_padding_mask=query_key_padding_mask, reference_points=ref_2d,
                        spatial_shapes=tsa_shapes, level_start_index=tsa_start, bev_h=bev_h,
                        bev_w=bev_w, **(tail if getattr(attn, 'supports_post_norm_train', False) else {}), **kwargs)
                    if tail and getattr(attn, 'supports_post_norm_train', False):
                        query, normed = query
                        if normed:
                            norm_index += 1
                            i += 1
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':  # spatial cross attention: no positional encoding (query_pos None)
                attn = self.attentions[attn_index]
                out = None
                if post_norm is not None and query_pos is None and hasattr(attn, 'forward_fused') \
                        and value is not None:
                    out = attn.forward_fused(query, value, reference_points_cam=reference_points_cam,
                                             bev_mask=kwargs.get('bev_mask'),
                                             spatial_shapes=spatial_shapes,
                                             level_start_index=level_start_index,
                                             vis_bits=kwargs.get('vis_bits'),
                                             bev_order=kwargs.get('sca_bev_order', kwargs.get('bev_order')),
                                             gather_stats=kwargs.get('gather_stats'),
                                             post_norm=post_norm)
                if out is not None:
                    query = out
                    norm_index += 1
                    i += 1
                else:
                    if hasattr(value, 'materialize'):    # LazyFeatures -> the reference-shaped fp32 tensor
                        key = value = value.materialize()
                    query = attn(
                        query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                        key_pos=key_pos, reference_points=ref_3d,
                        reference_points_cam=reference_points_cam, mask=mask,
                        attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask,
                        spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        **(tail if getattr(attn, 'supports_post_norm_train', False) else {}), **kwargs)
                    if tail and getattr(attn, 'supports_post_norm_train', False):
                        query, normed = query
                        if normed:
                            norm_index += 1
                            i += 1
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                ffn = self.ffns[ffn_index]
                out = None
                if post_norm is not None and hasattr(ffn, 'forward_fused'):
                    out = ffn.forward_fused(query, None, post_norm=post_norm)
                if out is not None:
                    query = out
                    norm_index += 1
                    i += 1
                elif tail and isinstance(ffn, FFN):
                    query, normed = ffn(query, identity if self.pre_norm else None, **tail)
                    if normed:
                        norm_index += 1
                        i += 1
                else:
                    query = ffn(query, identity if self.pre_norm else None)
                ffn_index += 1
            i += 1
        return query


    def forward_chain(self, query, value, bev_pos=None, ref_2d=None, bev_h=None, bev_w=None,
                      reference_points_cam=None, spatial_shapes=None, level_start_index=None, prev_bev=None,
                      tsa_pre=None, next_tsa=None, **kwargs):
        """Inference form of the whole layer on TWO Linear launches (csrc/linear_chain_x3.hip) around the two fused
        gathers: [TSA gather] -> output_proj + LN + the SCA's query Linears -> [SCA gather] -> output_proj + LN + FFN +
        LN (+ the NEXT layer's TSA query Linears and value projection when `next_tsa` is given and there is no
        history BEV).  `tsa_pre` = that tail of the previous layer.  -> (output, tail for the next layer or None);
        raises OccAmdUnsupported when a shape has no chain kernel (the caller then runs forward())."""
        tsa, sca = self.attentions
        ffn = self.ffns[0]
        bs = query.shape[0]
        if not tsa._fusable(ref_2d):
            raise ext.OccAmdUnsupported("forward_chain: TSA shape without a fused gather")
        q = query.contiguous()
        attn = tsa.fused_gather(q, prev_bev, bev_pos, ref_2d, bev_h, bev_w, kwargs.get('bev_order'), pre=tsa_pre)
        wq, bq = sca.query_linear_operands()
        x1, lin = ext.linear_ln_chain(attn, q, tsa.output_proj.weight, tsa.output_proj.bias, self.norms[0], wq, bq)
        slots = sca.fused_gather(lin, value, reference_points_cam, kwargs.get('bev_mask'), spatial_shapes,
                                 level_start_index, kwargs.get('vis_bits'), kwargs.get('sca_bev_order', kwargs.get('bev_order')),
                                 kwargs.get('gather_stats'))
        tail = None
        if next_tsa is not None and prev_bev is None and bs == 1 and bev_pos is not None:
            tail = next_tsa.chain_tail(bev_pos)
        fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
        out, zq, zv = ext.encoder_ffn_chain(slots, x1, sca.output_proj.weight, sca.output_proj.bias, self.norms[1],
                                            fc1.weight, fc1.bias, fc2.weight, fc2.bias, self.norms[2], tail=tail)
        return out, (None if tail is None else (zq, zv))


def _chain_layer_ok(layer):
    """True when `layer` is the base-config BEVFormerLayer the chain kernels cover: post-norm TSA -> LN -> SCA -> LN ->
    FFN -> LN with embed_dims 256 and a plain 256 -> 512 -> 256 ReLU FFN."""
    from .bricks import FFN
    from .spatial_cross_attention import SpatialCrossAttention
    from .temporal_self_attention import TemporalSelfAttention
    if not isinstance(layer, BEVFormerLayer) or not layer.use_fused or layer.embed_dims != 256:
        return False
    if tuple(layer.operation_order) != ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'):
        return False
    tsa, sca = layer.attentions
    ffn = layer.ffns[0]
    return (type(tsa) is TemporalSelfAttention and type(sca) is SpatialCrossAttention and tsa.use_fused
            and sca.use_fused and all(isinstance(n, nn.LayerNorm) for n in layer.norms)
            and isinstance(ffn, FFN) and ffn.num_fcs == 2 and ffn.add_identity
            and isinstance(ffn.layers[0][1], nn.ReLU) and isinstance(ffn.dropout_layer, nn.Identity)
            and ffn.feedforward_channels == 512 and sca.query_linear_operands() is not None)


class TransformerLayerSequence(BaseModule):
    """mmcv TransformerLayerSequence: `layers` = num_layers deep copies of the layer cfg."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class BEVFormerEncoder(TransformerLayerSequence):
    """Builds the pillar / BEV-plane reference points, projects the pillars into every camera and
    runs the layer stack (reference: encoder.py:28-239)."""

    def __init__(self, *args, pc_range=None, num_points_in_pillar=4, return_intermediate=False,
                 dataset_type='nuscenes', **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.num_points_in_pillar = num_points_in_pillar
        self.pc_range = pc_range
        self.fp16_enabled = False
        self._order_cache = {}
        self._ref_cache = {}
        self.last_gather_stats = None

    @staticmethod
    def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim='3d', bs=1, device='cuda',
                             dtype=torch.float):
        """dim='3d': (bs, num_points_in_pillar, H*W, 3) pillar points, normalised; query index
        q = y*W + x.  dim='2d': (bs, H*W, 1, 2) BEV-plane points (reference :50-89).

        The grids are constants of the geometry: they are evaluated once on the host (so they are
        bit-identical to the reference's CPU path — torch.linspace rounds differently on the device)
        and moved to `device`; BEVFormerEncoder.forward caches them."""
        if dim == '3d':
            zs = torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype) / Z
            xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype) / W
            ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype) / H
            ref_3d = torch.stack((xs.view(1, 1, W).expand(num_points_in_pillar, H, W),
                                  ys.view(1, H, 1).expand(num_points_in_pillar, H, W),
                                  zs.view(-1, 1, 1).expand(num_points_in_pillar, H, W)), -1)
            ref_3d = ref_3d.reshape(num_points_in_pillar, H * W, 3)
            return ref_3d[None].repeat(bs, 1, 1, 1).to(device)
        elif dim == '2d':
            ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype) / H
            xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype) / W
            ref_2d = torch.stack((xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W)), -1)
            return ref_2d.reshape(1, H * W, 2).repeat(bs, 1, 1).unsqueeze(2).to(device)
        raise ValueError(f"dim must be '3d' or '2d', got {dim!r}")

    def _reference_grids(self, bev_h, bev_w, bs, device, dtype):
        key = (bev_h, bev_w, bs, str(device), dtype)
        if key not in self._ref_cache:
            ref_3d = self.get_reference_points(bev_h, bev_w, self.pc_range[5] - self.pc_range[2],
                                               self.num_points_in_pillar, dim='3d', bs=bs,
                                               device=device, dtype=dtype)
            ref_2d = self.get_reference_points(bev_h, bev_w, dim='2d', bs=bs, device=device,
                                               dtype=dtype)
            hybrid = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, bev_h * bev_w, 1, 2)
            self._ref_cache = {key: (ref_3d, ref_2d, hybrid.contiguous(),
                                     torch.tensor([[bev_h, bev_w]], device=device),
                                     torch.tensor([0], device=device))}
        return self._ref_cache[key]

    def point_sampling(self, reference_points, pc_range, img_metas, return_vis=False):
        """-> reference_points_cam (num_cam, bs, H*W, Z, 2), bev_mask (num_cam, bs, H*W, Z) bool
        [, vis_bits (bs, H*W) int32].  fp32 (reference :91-92).  `ego2lidar` and `img_shape` are
        read from img_metas[0] for the whole batch, like the reference (:94, :133-134)."""
        _require_device(reference_points, 'BEVFormerEncoder.point_sampling')
        dev = reference_points.device
        lidar2img, ego2lidar = self._upload_matrices(
            np.asarray([m['lidar2img'] for m in img_metas], dtype=np.float32),
            np.asarray(img_metas[0]['ego2lidar'], dtype=np.float32), dev)
        img_h, img_w = img_metas[0]['img_shape'][0][0], img_metas[0]['img_shape'][0][1]
        ref_cam, mask, vis = ext.point_sampling(reference_points.float().contiguous(), lidar2img,
                                                ego2lidar, pc_range, img_h, img_w)
        return (ref_cam, mask, vis) if return_vis else (ref_cam, mask)

    def _upload_matrices(self, lidar2img, ego2lidar, dev):
        """The per-sample camera matrices in ONE asynchronous host-to-device copy out of a small ring of pinned
        staging buffers (two pageable copies per step each stalled the launch queue).  A slot is reused only
        after the copy that read it has completed."""
        n1, n2 = lidar2img.size, ego2lidar.size
        ring = getattr(self, '_mat_ring', None)
        if ring is None or ring['n'] != n1 + n2 or ring['dev'] != dev:
            ring = dict(n=n1 + n2, dev=dev, i=0,
                        host=[torch.empty(n1 + n2, dtype=torch.float32).pin_memory() for _ in range(4)],
                        done=[None] * 4)
            self._mat_ring = ring
        i = ring['i']
        ring['i'] = (i + 1) % 4
        if ring['done'][i] is not None:
            ring['done'][i].synchronize()
        h = ring['host'][i]
        h[:n1].copy_(torch.from_numpy(lidar2img.reshape(-1)))
        h[n1:].copy_(torch.from_numpy(ego2lidar.reshape(-1)))
        d = h.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring['done'][i] = ev
        return d[:n1].view(lidar2img.shape), d[n1:].view(ego2lidar.shape)

    def _query_major_pos(self, bev_pos):
        """(nq, bs, C) positional encoding -> contiguous (bs, nq, C).  The head hands out the same
        tensor while its embedding tables are unchanged (inference): keep the transposed copy."""
        if torch.is_grad_enabled() and bev_pos.requires_grad:
            return bev_pos.permute(1, 0, 2).contiguous()
        key = (bev_pos.data_ptr(), bev_pos._version, tuple(bev_pos.shape), tuple(bev_pos.stride()), cache_epoch())
        if getattr(self, '_pos_key', None) != key:
            # keep `bev_pos` referenced: its address cannot be recycled while this entry is live
            self._pos_key, self._pos_src, self._pos_qm = key, bev_pos, bev_pos.permute(1, 0, 2).contiguous()
        return self._pos_qm

    def _bev_order(self, bev_h, bev_w, device, flat=False):
        """Query processing order of the gather kernels: 8x8 BEV tiles, the tile list dealt over the 8 XCDs in
        4-query blocks (one wave per query, four waves per block).  flat: the plain tile walk — the head-major SCA gather deals
        HEADS to the XCDs, every XCD walks all queries."""
        key = (bev_h, bev_w, str(device), bool(flat))
        if key not in self._order_cache:
            # head-major SCA gather: a wave owns 8 consecutive entries — emitted as 2 x 4 patches of the BEV (0.189 against 0.196 ms
            # per launch for 1 x 8 strips; 8 x 1 0.197, 4 x 2 0.190: profiles/r06_c13_sca_patch_shape.txt)
            self._order_cache[key] = torch.from_numpy(
                bev_tile_order(bev_h, bev_w, n_xcd=1 if flat else 8, patch=(2, 4) if flat else None)).to(device)
        return self._order_cache[key]

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                **kwargs):
        """bev_query (num_query, bs, C); key/value (num_cam, num_value, bs, C);
        -> (bs, num_query, C) (or (num_layers, bs, num_query, C) with return_intermediate)."""
        _require_device(bev_query, 'BEVFormerEncoder')
        intermediate = []
        bs = bev_query.size(1)
        ref_3d, ref_2d, hybrid_same, tsa_shapes, tsa_start = self._reference_grids(
            bev_h, bev_w, bs, bev_query.device, bev_query.dtype)
        reference_points_cam, bev_mask, vis_bits = self.point_sampling(
            ref_3d, self.pc_range, kwargs['img_metas'], return_vis=True)
        # the reference keeps `shift_ref_2d = ref_2d.clone()`: no ego-motion shift in this variant
        shift_ref_2d = ref_2d
        bev_query = bev_query.permute(1, 0, 2)
        bev_pos = self._query_major_pos(bev_pos)          # (bs, nq, C) contiguous, cached while constant
        bs, len_bev, num_bev_level, _ = ref_2d.shape
        if prev_bev is not None:
            prev_bev = prev_bev.permute(1, 0, 2)
            prev_bev = torch.stack([prev_bev, bev_query], 1).reshape(bs * 2, len_bev, -1)
            hybird_ref_2d = torch.stack([shift_ref_2d, ref_2d], 1).reshape(
                bs * 2, len_bev, num_bev_level, 2)
        else:
            hybird_ref_2d = hybrid_same
        extra = dict(vis_bits=vis_bits, bev_order=self._bev_order(bev_h, bev_w, bev_query.device),
                     sca_bev_order=self._bev_order(bev_h, bev_w, bev_query.device, flat=ext.sca_head_major()),
                     tsa_spatial_shapes=tsa_shapes, tsa_level_start_index=tsa_start)
        output = bev_query
        if hasattr(value, 'prefetch') and not torch.is_grad_enabled():
            vps = [getattr(getattr(a, 'deformable_attention', None), 'value_proj', None)
                   for layer in self.layers for a in layer.attentions]
            value.prefetch([vp for vp in vps if vp is not None])
        # inference: the row-local Linear chains of a layer as two launches (BEVFormerLayer.forward_chain)
        # (the chain path takes none of the mask / query_pos arguments of BEVFormerLayer.forward: a caller that passes one
        # gets the unfused layer, never an unmasked result — ADVICE r4)
        chain = (ext.LINEAR_CHAIN and ext.LINEAR_PRECISION == "bf16x3" and not self.training and not args
                 and all(kwargs.get(k) is None for k in ('attn_masks', 'query_key_padding_mask', 'key_padding_mask',
                                                         'query_pos', 'mask'))
                 and bev_query.dtype == torch.float32 and value is not None
                 and not (torch.is_grad_enabled() and (bev_query.requires_grad or any(
                     p.requires_grad for p in self.parameters()))))
        chain_ok = [chain and _chain_layer_ok(layer) for layer in self.layers]
        tsa_pre = None
        try:
            for lid, layer in enumerate(self.layers):
                output = None
                if chain_ok[lid]:
                    nxt = self.layers[lid + 1].attentions[0] if lid + 1 < len(self.layers) and chain_ok[lid + 1] else None
                    try:
                        output, tsa_pre = layer.forward_chain(
                            bev_query, value, bev_pos=bev_pos, ref_2d=hybird_ref_2d, bev_h=bev_h, bev_w=bev_w,
                            reference_points_cam=reference_points_cam, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, prev_bev=prev_bev, tsa_pre=tsa_pre, next_tsa=nxt,
                            bev_mask=bev_mask, gather_stats=kwargs.get('gather_stats'), **extra)
                    except ext.OccAmdUnsupported:
                        output, tsa_pre = None, None
                    except ext.OccAmdError as e:
                        # a launch-side refusal (e.g. the 77.8 KB dynamic-LDS attribute of the chain kernels): the
                        # separate launches of layer() compute the same thing
                        if not getattr(self, '_chain_warned', False):
                            import warnings
                            warnings.warn(f"encoder chain kernels unavailable ({e}): this and later calls fall back to the "
                                          f"per-op launches of BEVFormerLayer.forward")
                            self._chain_warned = True
                        output, tsa_pre = None, None
                if output is None:
                    tsa_pre = None
                    output = layer(bev_query, key, value, *args, bev_pos=bev_pos, ref_2d=hybird_ref_2d,
                                   ref_3d=ref_3d, bev_h=bev_h, bev_w=bev_w, spatial_shapes=spatial_shapes,
                                   level_start_index=level_start_index,
                                   reference_points_cam=reference_points_cam, bev_mask=bev_mask,
                                   prev_bev=prev_bev, **extra, **kwargs)
                bev_query = output
                if self.return_intermediate:
                    intermediate.append(output)
        finally:
            if hasattr(value, 'finish'):
                value.finish()          # join the value-projection side stream, drop unconsumed projections
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output
