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
from .bricks import BaseModule, ModuleList, build_norm_layer
from .registry import (TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_attention,
                       build_feedforward_network, build_transformer_layer)
from .spatial_cross_attention import _require_device


# the SCA value projections of ALL layers depend on the camera features only: they start on a side stream before the
# first layer (LazyFeatures.prefetch) and run under the TSA / Linear kernels instead of serially before each gather
# (6.764 -> 6.734 ms per sample, ABAB on one box; OCC_VPROJ_OVERLAP=0 restores the serial order)
_VPROJ_OVERLAP = os.environ.get("OCC_VPROJ_OVERLAP", "1") == "1"

# EXPERIMENT, off by default (built and measured in the last GPU minutes of round 4):
# OCC_ENCODER_ROW_PIPELINE=K (K >= 2) cuts the BEV queries into K row bands and walks every layer band by band on K HIP
# streams.  Between two TSA gathers everything is ROW-LOCAL — chain program A, the SCA gather (its value operand is the
# camera planes, not the BEV) and chain program B only ever touch their own rows — so band 2's TSA gather / program A can
# run under band 1's SCA gather, and band 1's program B under band 2's SCA gather: kernels bound by different units
# (texture path / HBM writes / matrix cores) share the chip instead of running back to back, each with its own ramp-up
# and drain.  Only the TSA gather of the NEXT layer needs all bands (its value operand is the whole BEV).
# What round 4 measured (profiles/r04_rowpipe_*): the banded launches are correct (one stream: 2.3e-5 against the standard
# path after 4 layers; K streams: the same) but the EAGER pipeline is host-bound — 2 bands double the encoder's launches and
# add stream switches and events: 2.44 ms of unqueued host time per hot-path step against 1.0 ms, so the wall time
# (2.39 ms against 2.28 ms) is the launch cost, not the overlap.  The whole standard hot-path step captures into a
# hipGraph (replay 2.31 ms); capturing the pipelined step crashed inside the runtime, which is where round 5 picks up.
# OCC_ROW_PIPELINE_SERIAL=1 adds events that keep two launches of the SAME kernel from overlapping (band i + 1's stage
# waits for band i's): KNOWN BAD — with them 60-100 of band 2's 19 200 queries come out wrong from the second layer on
# (a device synchronize per layer cures it, stream-to-stream barriers do not: not understood, profiles/r04_rowpipe_debug3.log).
# OCC_ROW_PIPELINE_STREAMS=0 / OCC_ROW_PIPELINE_DEBUG_SYNC=1|2 are the debugging switches of tools_dev/row_pipeline_debug.py.
_ROW_PIPELINE = int(os.environ.get("OCC_ENCODER_ROW_PIPELINE", "0") or 0)
_ROW_PIPELINE_SERIAL = os.environ.get("OCC_ROW_PIPELINE_SERIAL", "0") == "1"
# OCC_ROW_PIPELINE_NATIVE=1: the banded sequence issued by ONE C-ABI call (csrc/encoder_bands.hip, ext.encoder_bands_forward)
# instead of ~35 Python-side launches — written after the GPU budget of round 4 was spent: compiled, host-tested, never run on
# an MI355X.  With OCC_ENCODER_ROW_PIPELINE=1 it is the unbanded chain path from one call.
_ROW_PIPELINE_NATIVE = os.environ.get("OCC_ROW_PIPELINE_NATIVE", "0") == "1"
# OCC_ROW_PIPELINE_FLAGS: scheduling flags of the native launcher (include/occnet_amd.h: 1 = bands one stage apart, 2 =
# band-major submission)
_ROW_PIPELINE_FLAGS = int(os.environ.get("OCC_ROW_PIPELINE_FLAGS", "0") or 0)


def row_bands(bev_h, bev_w, k, tile_h=8):
    """k contiguous bands of BEV rows with boundaries on multiples of tile_h (the gather kernels walk the queries in
    tile_h x 8 tiles), as equal as the tile rows allow -> [(first query, one past the last query, band height)];
    fewer than k bands when there are fewer tile rows."""
    nt = (bev_h + tile_h - 1) // tile_h
    k = max(1, min(int(k), nt))
    ys = [min(bev_h, ((i * nt + k // 2) // k) * tile_h) for i in range(k)] + [bev_h]
    return [(ys[i] * bev_w, ys[i + 1] * bev_w, ys[i + 1] - ys[i]) for i in range(k) if ys[i + 1] > ys[i]]


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
        ops = self.operation_order
        i = 0
        while i < len(ops):
            layer = ops[i]
            post_norm = None
            if fuse and i + 1 < len(ops) and ops[i + 1] == 'norm' \
                    and isinstance(self.norms[norm_index], nn.LayerNorm):
                post_norm = self.norms[norm_index]
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
                        bev_w=bev_w, **kwargs)
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
                                             bev_order=kwargs.get('bev_order'),
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
                        spatial_shapes=spatial_shapes, level_start_index=level_start_index, **kwargs)
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
                                 level_start_index, kwargs.get('vis_bits'), kwargs.get('bev_order'),
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

    def _bev_order(self, bev_h, bev_w, device):
        """Query processing order of the gather kernels: 8x8 BEV tiles, the tile list dealt over the 8 XCDs in
        4-query blocks (one wave per query, four waves per block)."""
        key = (bev_h, bev_w, str(device))
        if key not in self._order_cache:
            self._order_cache[key] = torch.from_numpy(bev_tile_order(bev_h, bev_w, n_xcd=8)).to(device)
        return self._order_cache[key]

    def _row_pipeline_plan(self, bev_h, bev_w, k, hybrid_ref_2d, device):
        """Constant per (geometry, K): the bands, their band-local query orders and TSA reference points, the streams."""
        key = (bev_h, bev_w, k, str(device), hybrid_ref_2d.data_ptr())
        plan = getattr(self, '_row_plan', None)
        if plan is None or plan['key'] != key:
            bands = []
            for m0, m1, h in row_bands(bev_h, bev_w, k):
                order = torch.from_numpy(bev_tile_order(h, bev_w, n_xcd=8)).to(device)
                bands.append(dict(m0=m0, m1=m1, order=order, ref_2d=hybrid_ref_2d[:, m0:m1].float().contiguous()))
            plan = dict(key=key, bands=bands, src=hybrid_ref_2d,
                        streams=[torch.cuda.Stream(device=device) for _ in bands])
            self._row_plan = plan
        return plan

    def _forward_row_pipeline(self, k, bev_query, value, bev_pos, hybrid_ref_2d, bev_h, bev_w, reference_points_cam,
                              spatial_shapes, level_start_index, vis_bits, gather_stats):
        """OCC_ENCODER_ROW_PIPELINE (see the switch above): all layers on the chain kernels, band by band on one stream
        per band.  Same kernels, same per-row arithmetic as forward_chain.  bs = 1, no history BEV.  -> the list of layer
        outputs (bs, nq, C), or None when this call has to take the standard path — the first call after any weight /
        cache change does, so that every derived operand (packed chain weights, folded positional terms, ...) is built
        on the main stream by the standard path before several streams use it."""
        sig = (cache_epoch(), sum(p._version for p in self.parameters()), bev_pos.data_ptr(), bev_pos._version)
        if getattr(self, '_row_sig', None) != sig:
            self._row_sig = sig
            return None
        dev = bev_query.device
        main = torch.cuda.current_stream(dev)
        plan = self._row_pipeline_plan(bev_h, bev_w, k, hybrid_ref_2d, dev)
        bands, streams = plan['bands'], plan['streams']
        if os.environ.get("OCC_ROW_PIPELINE_STREAMS", "1") == "0":     # (debugging) every band on the CALLER's stream —
            streams = [main] * len(bands)                               # resolved per call: it may be a capturing stream
            plan = dict(plan, streams=streams)
        if _ROW_PIPELINE_NATIVE:
            return self._forward_row_pipeline_native(plan, bev_query, value, bev_pos, bev_h, bev_w, reference_points_cam,
                                                     spatial_shapes, level_start_index, vis_bits, gather_stats)
        if len(bands) < 2:
            return None
        nq, nl = bev_h * bev_w, len(self.layers)
        q_full = bev_query.contiguous()
        band = lambda t, b: t[0, b['m0']:b['m1']].unsqueeze(0)          # rows of a (1, nq, n) buffer, as (1, rows, n)
        # ---- main stream, before the fork: band copies of the per-step operands, the first layer's TSA query Linears
        # and value projection (program C on all rows: the first TSA gather needs the whole projected BEV), the buffers
        # the bands share
        ref_cam = reference_points_cam.float()
        per_band = [dict(ref_cam=ref_cam[:, :, b['m0']:b['m1']].contiguous(), vis=vis_bits[:, b['m0']:b['m1']])
                    for b in bands]
        tsa0 = self.layers[0].attentions[0]
        w_sum, pos_term, wv, bv = tsa0.chain_tail(bev_pos)
        zq, zv = ext.linear_pair_chain(q_full, w_sum, pos_term, wv, bv)
        outs = [torch.empty((1, nq, 256), dtype=torch.float32, device=dev) for _ in range(nl)]
        tails = []
        for lid in range(nl - 1):
            nxt = self.layers[lid + 1].attentions[0]
            t = nxt.chain_tail(bev_pos)
            tails.append((t, torch.empty((1, nq, t[0].shape[0]), dtype=torch.float32, device=dev),
                          torch.empty((1, nq, 256), dtype=torch.float32, device=dev)))
        shared = [q_full, zq, zv, ref_cam, vis_bits] + outs + [x for _, a, b_ in tails for x in (a, b_)]
        shared += [d['ref_cam'] for d in per_band]
        fork = torch.cuda.Event()
        fork.record(main)
        for s in streams:
            s.wait_event(fork)
            if s != main:
                for t in shared:
                    t.record_stream(s)
        keep = []                                   # band-private tensors stay referenced until the join
        prev_b = None                               # the previous layer's program-B events, one per band
        q_prev = q_full
        try:
            for lid, layer in enumerate(self.layers):
                tsa, sca = layer.attentions
                ffn = layer.ffns[0]
                fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
                n_off = tsa.sampling_offsets.out_features
                v4 = zv.view(1, nq, tsa.num_heads, -1)
                wq, bq = sca.query_linear_operands()
                plane = value.project_on(sca.deformable_attention.value_proj, streams)
                plane_scale = value.value_scale(sca.deformable_attention.value_proj)
                st = [dict() for _ in bands]
                ev = {name: [None] * len(bands) for name in 'TASB'}

                def stage(name, fn):
                    for i, (b, s) in enumerate(zip(bands, streams)):
                        with torch.cuda.stream(s):
                            if _ROW_PIPELINE_SERIAL and i > 0:
                                s.wait_event(ev[name][i - 1])
                                if os.environ.get("OCC_ROW_PIPELINE_DUMMY") == "1":     # (debugging) a tiny kernel between
                                    torch.zeros(1, device=dev)                          # the waits and the stage's launch
                            fn(i, b, s)
                            e = torch.cuda.Event()
                            e.record(s)
                            ev[name][i] = e

                def t_stage(i, b, s):
                    if prev_b is not None:          # the TSA gather reads the WHOLE projected BEV of the layer before
                        for j, e in enumerate(prev_b):
                            if j != i:
                                s.wait_event(e)
                    lin = band(zq, b)
                    st[i]['attn'] = ext.tsa_fused_forward(
                        v4, lin[..., :n_off], lin[..., n_off:], b['ref_2d'], bev_h, bev_w, tsa.num_heads,
                        tsa.num_points, shared_queue=True, order=b['order'], value_rows=nq)

                def a_stage(i, b, s):
                    st[i]['x1'], st[i]['lin'] = ext.linear_ln_chain(
                        st[i]['attn'], band(q_prev, b), tsa.output_proj.weight, tsa.output_proj.bias, layer.norms[0],
                        wq, bq)

                def s_stage(i, b, s):
                    st[i]['slots'] = sca.gather_projected(
                        st[i]['lin'], plane, per_band[i]['ref_cam'], per_band[i]['vis'], spatial_shapes,
                        level_start_index, b['order'], gather_stats, value_scale=plane_scale)

                def b_stage(i, b, s):
                    tail, out = None, (band(outs[lid], b), None, None)
                    if lid + 1 < nl:
                        (tw, tterm, twv, tbv), nzq, nzv = tails[lid]
                        tail = (tw, band(tterm, b), twv, tbv)
                        out = (out[0], band(nzq, b), band(nzv, b))
                    ext.encoder_ffn_chain(st[i]['slots'], st[i]['x1'], sca.output_proj.weight, sca.output_proj.bias,
                                          layer.norms[1], fc1.weight, fc1.bias, fc2.weight, fc2.bias, layer.norms[2],
                                          tail=tail, out=out)

                stage('T', t_stage)
                stage('A', a_stage)
                stage('S', s_stage)
                stage('B', b_stage)
                keep.append(st)
                dbg = os.environ.get("OCC_ROW_PIPELINE_DEBUG_SYNC")          # (debugging)
                if dbg == "1":                      # a device barrier per layer
                    torch.cuda.synchronize()
                elif dbg == "2":                    # every band stream waits for every other one per layer (device side)
                    for s in streams:
                        for s2 in streams:
                            if s2 != s:
                                s.wait_stream(s2)
                prev_b = ev['B']
                q_prev = outs[lid]
                if lid + 1 < nl:
                    zq, zv = tails[lid][1], tails[lid][2]
        finally:
            for s in streams:                       # join (also when a launch raised): nothing may outlive this call
                if s != main:
                    main.wait_stream(s)
        if os.environ.get("OCC_ROW_PIPELINE_DEBUG_KEEP") == "1":     # (debugging: tools_dev/row_pipeline_bisect.py)
            self._row_debug = dict(keep=keep, tails=tails, outs=outs, bands=bands)
        return outs

    def _forward_row_pipeline_native(self, plan, bev_query, value, bev_pos, bev_h, bev_w, reference_points_cam,
                                     spatial_shapes, level_start_index, vis_bits, gather_stats):
        """The banded sequence of _forward_row_pipeline through ext.encoder_bands_forward: this method only gathers the
        operands (band copies of the per-step reference points, the first layer's TSA Linears, the planes and their
        event, the result buffers); csrc/encoder_bands.hip forks the band streams, issues every launch and joins."""
        dev = bev_query.device
        main = torch.cuda.current_stream(dev)
        bands, streams = plan['bands'], plan['streams']
        if len(bands) == 1:
            streams = [main]
        nq, nl = bev_h * bev_w, len(self.layers)
        sca0 = self.layers[0].attentions[1].deformable_attention
        n_lin = sca0.sampling_offsets.out_features + sca0.attention_weights.out_features
        ref_cam = reference_points_cam.float()
        for b in bands:                              # band scratch: allocated once (main stream), reused every step
            n = b['m1'] - b['m0']
            if 'attn' not in b:
                for name, w in (('attn', 256), ('x1', 256), ('slots', 256), ('lin', n_lin)):
                    b[name] = torch.empty((1, n, w), dtype=torch.float32, device=dev)
                b['ref_cam'] = torch.empty((ref_cam.shape[0], 1, n) + tuple(ref_cam.shape[3:]), dtype=torch.float32,
                                           device=dev)
            b['ref_cam'].copy_(ref_cam[:, :, b['m0']:b['m1']])
        q_full = bev_query.contiguous()
        tsa0 = self.layers[0].attentions[0]
        zq0, zv0 = ext.linear_pair_chain(q_full, *tsa0.chain_tail(bev_pos))
        shared = [q_full, zq0, zv0, vis_bits]
        layers = []
        for lid, layer in enumerate(self.layers):
            tsa, sca = layer.attentions
            ffn = layer.ffns[0]
            wq, bq = sca.query_linear_operands()
            plane, ev = value.take_on(sca.deformable_attention.value_proj, [s for s in streams if s != main])
            y = dict(a=(tsa.output_proj.weight, tsa.output_proj.bias, layer.norms[0], wq, bq),
                     b=(sca.output_proj.weight, sca.output_proj.bias, layer.norms[1], ffn.layers[0][0].weight,
                        ffn.layers[0][0].bias, ffn.layers[1].weight, ffn.layers[1].bias, layer.norms[2]),
                     plane=plane.view(plane.shape[0], plane.shape[1], sca.deformable_attention.num_heads, -1),
                     plane_ready=ev, plane_scale=value.value_scale(sca.deformable_attention.value_proj),
                     stats=gather_stats if gather_stats is not None else sca.gather_stats,
                     out=torch.empty((1, nq, 256), dtype=torch.float32, device=dev))
            shared.append(y['out'])
            if lid + 1 < nl:
                t = self.layers[lid + 1].attentions[0].chain_tail(bev_pos)
                y.update(tail=t, zq=torch.empty((1, nq, t[0].shape[0]), dtype=torch.float32, device=dev),
                         zv=torch.empty((1, nq, 256), dtype=torch.float32, device=dev))
                shared += [y['zq'], y['zv']]
            layers.append(y)
        for s in streams:
            if s != main:
                for t in shared:
                    t.record_stream(s)
        tsa = self.layers[0].attentions[0]
        ext.encoder_bands_forward(q_full, zq0, zv0, layers,
                                  [dict(b, stream=s) for b, s in zip(bands, streams)], spatial_shapes, level_start_index,
                                  vis_bits, bev_h, bev_w, sca0.num_levels, sca0.num_points, tsa.num_points,
                                  flags=_ROW_PIPELINE_FLAGS)
        return [y['out'] for y in layers]

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
                     tsa_spatial_shapes=tsa_shapes, tsa_level_start_index=tsa_start)
        output = bev_query
        if _VPROJ_OVERLAP and hasattr(value, 'prefetch') and not torch.is_grad_enabled():
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
            if ((_ROW_PIPELINE >= 2 or (_ROW_PIPELINE == 1 and _ROW_PIPELINE_NATIVE)) and all(chain_ok) and bs == 1
                    and prev_bev is None and hasattr(value, 'project_on')
                    and bev_pos is not None and not torch.is_grad_enabled()):
                try:
                    piped = self._forward_row_pipeline(
                        _ROW_PIPELINE, bev_query, value, bev_pos, hybird_ref_2d, bev_h, bev_w, reference_points_cam,
                        spatial_shapes, level_start_index, vis_bits, kwargs.get('gather_stats'))
                except ext.OccAmdUnsupported:
                    piped = None            # (streams joined; projections it consumed are redone by the layers below)
                if piped is not None:
                    return torch.stack(piped) if self.return_intermediate else piped[-1]
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
