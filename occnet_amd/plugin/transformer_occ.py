"""TransformerOcc on the MI355X path: camera features -> BEV embedding -> voxel features -> heads.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/modules/transformer_occ.py (registry
name, constructor kwargs incl. the accepted-and-ignored use_shift / use_can_bus / can_bus_norm /
two_stage_num_proposals / decoder, parameter names `level_embeds`, `cams_embeds`, `encoder`,
`decoder.{0,1}.{conv,bn}`, `predicter`, `flow_predicter`).

Data layout: the flattened multi-camera key/value tensor is produced directly as
(bs*num_cams, sum_l H_l*W_l, C) — the (B, Cam, H, W, C) layout the gather kernels want, one pixel's
256 channels contiguous — and handed to the encoder as a permuted VIEW in the reference's
(num_cams, sum HW, bs, C) axis order, so SpatialCrossAttention's permute+reshape back is free.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import cache_epoch, ext
from .._lib import OccAmdUnsupported
from .bricks import BaseModule, ConvModule, X3Linear
from .registry import TRANSFORMER, build_transformer_layer_sequence
from .spatial_cross_attention import MSDeformableAttention3D, _require_device
from .temporal_self_attention import TemporalSelfAttention


_ROT_CACHE = {}


def _rotation_source_index(H, W, angle_deg, center):
    """For every output pixel of a nearest-neighbour rotation by angle_deg (counter-clockwise, degrees) about
    `center` (x, y): the flat index of its source pixel, or -1 outside the map.  Evaluated ON THE HOST in float32 with
    the affine-grid formulation torchvision's `rotate` uses (inverse rotation matrix about the centre -> normalised
    sampling grid -> grid_sample(mode='nearest', align_corners=False)), by rotating an image of pixel indices: which
    neighbour a near-tie coordinate rounds to then matches the reference's CPU path bit for bit.  (Evaluated on the
    device, fused multiply-adds move a handful of the 40 000 coordinates across a rounding boundary: one wrong
    source pixel is an O(1) error in the history BEV.)"""
    import math
    key = (H, W, float(angle_deg), float(center[0]), float(center[1]))
    hit = _ROT_CACHE.get(key)
    if hit is None:
        cx, cy = center[0] - W * 0.5, center[1] - H * 0.5
        rot = math.radians(-float(angle_deg))
        cos, sin = math.cos(rot), math.sin(rot)
        m = [cos, sin, 0.0, -sin, cos, 0.0]
        m[2] += m[0] * (-cx) + m[1] * (-cy) + cx
        m[5] += m[3] * (-cx) + m[4] * (-cy) + cy
        theta = torch.tensor(m, dtype=torch.float32).view(1, 2, 3)
        base = torch.empty(1, H, W, 3, dtype=torch.float32)
        base[..., 0].copy_(torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W))
        base[..., 1].copy_(torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H).unsqueeze(-1))
        base[..., 2].fill_(1)
        grid = base.view(1, H * W, 3).bmm(theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H])).view(1, H, W, 2)
        # pixel indices + 1 (exact in float32 below 2^24), zero padding marks "outside"
        ids = torch.arange(1, H * W + 1, dtype=torch.float32).view(1, 1, H, W)
        src = F.grid_sample(ids, grid, mode='nearest', padding_mode='zeros', align_corners=False).view(-1)
        hit = src.to(torch.int64) - 1
        if torch.cuda.is_available():
            # pinned: the per-frame upload is then an asynchronous copy (pageable memory makes it a synchronous one)
            try:
                hit = hit.pin_memory()
            except RuntimeError:
                pass
        if len(_ROT_CACHE) >= 16:
            _ROT_CACHE.pop(next(iter(_ROT_CACHE)))
        _ROT_CACHE[key] = hit
    return hit


def rotate_bev_nearest(bev, angle_deg, center):
    """Rotate a (C, H, W) BEV map by angle_deg (counter-clockwise, degrees) about `center` (x, y) in
    pixels, nearest-neighbour, zero fill — the operation the reference applies to the history BEV
    with torchvision's `rotate(img, angle, center=rotate_center)` (transformer_occ.py:195-205).  The source-pixel map
    comes from the host (see _rotation_source_index); the device only gathers."""
    C, H, W = bev.shape
    assert H * W < (1 << 24)
    src = _rotation_source_index(H, W, angle_deg, center).to(bev.device, non_blocking=True)
    gathered = bev.reshape(C, H * W).index_select(1, src.clamp(min=0))
    # pixels whose source lies outside the map are ZERO whatever pixel 0 holds (a product with 0 would spread a
    # NaN / Inf at bev[:, 0, 0] to every padded pixel)
    out = torch.where((src >= 0).unsqueeze(0), gathered, gathered.new_zeros(()))
    return out.view(C, H, W)


class LazyFeatures:
    """The SCA `value` input kept as what the backbone emitted — bf16 NHWC FPN maps — plus the (level, camera)
    embedding table, instead of the reference's flattened fp32 tensor (`transformer_occ.py:204-222`).
    `project(value_proj)` runs the projection straight off the maps (ext.value_proj_bf16: the embeddings become
    a per-(level, camera) bias because the projection is linear), so the 189 MB fp32 flatten buffer is neither
    written nor read; `materialize()` builds the reference-shaped tensor for consumers without that kernel."""

    _shape_cache = {}

    def __init__(self, owner, mlvl_feats):
        self.owner = owner
        self.mlvl_feats = mlvl_feats
        self.bs, self.num_cam, self.c = mlvl_feats[0].shape[:3]
        self.hw = [(f.shape[3], f.shape[4]) for f in mlvl_feats]
        self.total = sum(h * w for h, w in self.hw)
        dev = mlvl_feats[0].device
        starts = [0]
        for h, w in self.hw[:-1]:
            starts.append(starts[-1] + h * w)
        self.starts = starts
        # the two index tensors are constant per feature geometry: uploaded once (a pageable host-to-device copy
        # per step stalls the launch queue)
        key = (tuple(self.hw), str(dev))
        hit = self._shape_cache.get(key)
        if hit is None:
            hit = (torch.as_tensor(self.hw, dtype=torch.long, device=dev),
                   torch.as_tensor(starts, dtype=torch.long, device=dev))
            hit[0]._occ_hw = tuple(self.hw)        # host copy of the shapes travels with the tensor object
            if len(self._shape_cache) > 16:
                self._shape_cache.clear()
            self._shape_cache[key] = hit
        self.spatial_shapes, self.level_start_index = hit
        # (bs*num_cam*h*w, C) pixel-major views of the NHWC maps
        self.rows = [f.permute(0, 1, 3, 4, 2).reshape(-1, self.c) for f in mlvl_feats]
        self._flat = None

    @staticmethod
    def eligible(mlvl_feats):
        if torch.is_grad_enabled() or not mlvl_feats:
            return False
        c = mlvl_feats[0].shape[2]
        return all(f.is_cuda and f.dtype == torch.bfloat16 and f.dim() == 5 and f.shape[2] == c and c % 32 == 0
                   and f.permute(0, 1, 3, 4, 2).is_contiguous() for f in mlvl_feats)

    def embeds(self):
        """(L, num_cam, C) fp32: level_embeds[l] (+ cams_embeds[cam])."""
        o = self.owner
        e = o.level_embeds[:len(self.hw), None, :].float()
        if o.use_cams_embeds:
            return (e + o.cams_embeds[None, :, :].float()).contiguous()
        return e.expand(-1, self.num_cam, -1).contiguous()

    def materialize(self):
        """The reference-shaped (num_cam, sum hw, bs, C) fp32 tensor."""
        if self._flat is None:
            flat, _, _ = self.owner.flatten_features(self.mlvl_feats)
            self._flat = flat.view(self.bs, self.num_cam, flat.shape[1], flat.shape[2]).permute(1, 2, 0, 3)
        return self._flat

    # (Round 5 measured three LAYERED schedules — plane l + 1 projected on a side stream from the start of layer l + 1, from
    # layer l's chain program B, or from layer l's gather on — against this one stacked launch, same box: 2.576 / 2.588 /
    # 2.545 ms per hot-path step against 2.49.  Rejected; profiles/r05_c2_vproj_schedule_ab.txt.  Rounds 2-4 ran the stacked
    # launch on a side stream under layer 0's TSA gather (2.255-2.272 against 2.25 ms: nothing gained); that switch,
    # OCC_VPROJ_OVERLAP, left the library in round 6: everything the library enqueues goes to the CALLER's stream.)

    def prefetch(self, value_projs):
        """project() for several layers' value_proj modules AHEAD of the layer stack, on the caller's stream: the projections
        depend on the camera features only, not on the BEV queries, so all of them go in ONE stacked launch (the maps are read
        once)."""
        gbs = [self._group_bias(vp) for vp in value_projs]
        stacked = (len(value_projs) > 1 and all(tuple(vp.weight.shape) == tuple(value_projs[0].weight.shape)
                                                for vp in value_projs) and value_projs[0].weight.shape[0] % 256 == 0)
        if stacked:
            try:
                ext.value_proj_planes_prepare([vp.weight for vp in value_projs])     # stack + pack
            except ext.OccAmdError:
                stacked = False
        if not stacked:
            for vp in value_projs:
                ext.linear_pack_weight_bf16x3(vp.weight)
        self._pending, self._scale_of = {}, {}
        scales = self._scales(value_projs, gbs)              # per-plane fp16 range scales of this call's maps
        if scales is not None:
            for l, vp in enumerate(value_projs):
                self._scale_of[id(vp)] = scales[l:l + 1]
        if stacked:
            # ONE launch for all layers: every block projects its rows with all the layers' weights, the feature
            # maps are read from HBM once (occ_value_proj_bf16_planes); layer l's values are plane l
            n = value_projs[0].weight.shape[0]
            try:
                out = self._alloc(n, planes=len(value_projs))
                ext.value_proj_bf16_planes(self.rows, [vp.weight for vp in value_projs], gbs, out,
                                           rows_per_group=[h * wd for h, wd in self.hw],
                                           out_group_rows=self.group_rows, out_row0=self.starts, out_scale=scales)
                for l, vp in enumerate(value_projs):
                    self._pending[id(vp)] = out[l].view(self.bs * self.num_cam, self.group_rows, n)
            except ext.OccAmdError:          # e.g. the 74 KB LDS attribute refused: one launch per layer instead
                stacked = False
                self._pending = {}
        if not stacked:
            for l, (vp, gb) in enumerate(zip(value_projs, gbs)):
                self._pending[id(vp)] = self._launch(vp, gb, None if scales is None else scales[l:l + 1])

    # fp16 value rows carry 11 significant bits and end at 65 504.  Every plane is therefore stored times a power of two
    # chosen per call ON THE DEVICE from an a-priori bound of its values (ext.value_range_scale: max|x| of this call's maps
    # x the projection's largest absolute row sum + its largest bias), so that no finite feature map can saturate; the gather
    # divides the scale out again (exact).  Round 4 only warned (|v| = 1.8e4 on the benchmarked maps, 3.6x under the limit).
    def _range_terms(self, value_proj, gb):
        """(max_n sum_k |W[n][k]|, max|group bias|) of a projection as a 2-element DEVICE tensor: computed by two small
        reductions on the stream when the weight state changes, never read back (round 6: no host synchronisation, valid
        inside a captured graph, follows the live weights)."""
        w = value_proj.weight
        key = (w.data_ptr(), w._version, gb.data_ptr(), gb._version, cache_epoch())
        hit = getattr(value_proj, '_occ_range_terms', None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                t = torch.stack([w.detach().float().abs().sum(1).amax(), gb.detach().abs().amax()])
            hit = (key, t, w, gb)                                # the sources stay referenced: the key stays unambiguous
            value_proj._occ_range_terms = hit
        return hit[1]

    def _absmax_words(self):
        """The 8 device words of max|x| the producer of the maps accumulated (the backbone plan's FPN output convolutions),
        or None: all levels must carry the SAME words object, and none may have been written in place since (ext.absmax_of)."""
        am = ext.absmax_of(self.mlvl_feats[0])
        if am is None or any(ext.absmax_of(f) is not am for f in self.mlvl_feats):
            return None
        return am

    def _scales(self, value_projs, gbs):
        """The planes' range scales for THIS call's maps, or None (fp32 rows): from the producer's maximum when the maps carry
        one (a 64-thread launch), otherwise measured over the maps (ext.value_range_scale: one pass, 52 us at the base config)."""
        if not ext.sca_rows_16bit():
            return None
        terms = [self._range_terms(vp, gb) for vp, gb in zip(value_projs, gbs)]
        # the stacked (2, P) term vectors are a constant of the weight state too: cached on the first projection (the entry
        # holds the per-projection tensors, so their identity keys it) — no small stack / copy kernels per step
        hit = getattr(value_projs[0], '_occ_range_terms_stacked', None)
        if hit is None or len(hit[0]) != len(terms) or any(a is not b for a, b in zip(hit[0], terms)):
            st = torch.stack(terms).t().contiguous()                                             # (2, P)
            hit = (terms, st)
            value_projs[0]._occ_range_terms_stacked = hit
        row_l1, bias_max = hit[1][0], hit[1][1]
        am = self._absmax_words()
        if am is not None:
            t = ext.value_range_scale_from_amax(am, row_l1, bias_max)
        else:
            t = ext.value_range_scale(self.rows, row_l1, bias_max)
        self.range_report_ = t                                   # 2 P + 1 device floats: scales, max|x|, bounds
        return t[:len(value_projs)]

    def value_scale(self, value_proj):
        """The 1-element device tensor the fused gather undoes plane `value_proj`'s range scale with (None: fp32 rows)."""
        return getattr(self, '_scale_of', {}).get(id(value_proj))

    def finish(self):
        """Drop projections no layer consumed (a layer fell back to the unfused path, an exception unwound the encoder)."""
        self._pending = {}

    def _group_bias(self, value_proj):
        """per-(level, camera) bias = (cams_embeds + level_embeds) . W^T + b: constant while the parameters are,
        cached on the projection module (the entry holds the parameters, so the keys stay unambiguous)"""
        w, b = value_proj.weight, value_proj.bias
        o = self.owner
        srcs = (w, b, o.level_embeds, o.cams_embeds if o.use_cams_embeds else None)
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in srcs) + (len(self.hw), cache_epoch())
        hit = getattr(value_proj, '_occ_group_bias', None)
        if hit is None or hit[0] != key:
            emb = self.embeds()                                                 # (L, cam, C)
            gb = emb.view(-1, self.c) @ w.t().float()
            if b is not None:
                gb = gb + b.float()
            hit = (key, gb.view(len(self.hw), self.num_cam, w.shape[0]).contiguous(), srcs)
            value_proj._occ_group_bias = hit
        return hit[1]

    @property
    def group_rows(self):
        """rows of one camera's block in the projected maps: fp16 maps are stored in pixel PAIRS (ext.sca_pair_layout),
        so an odd pixel count is padded by one (never written, never read: no sampling corner maps to it)"""
        return self.total + (self.total & 1) if ext.sca_rows_16bit() else self.total

    def _alloc(self, n, planes=None):
        rows = self.bs * self.num_cam * self.group_rows
        return torch.empty((rows, n) if planes is None else (planes, rows, n), device=self.rows[0].device,
                           dtype=ext.sca_rows_dtype())

    def _launch(self, value_proj, gb, scale="measure"):
        """One projection on the current stream.  scale: its range scale (1-element device tensor), None (fp32 rows), or
        "measure": derive it here (a projection that was not prefetched)."""
        w = value_proj.weight
        n = w.shape[0]
        if isinstance(scale, str):
            scale = self._scales([value_proj], [gb])
            if not hasattr(self, '_scale_of'):
                self._scale_of = {}
            self._scale_of[id(value_proj)] = scale
        out = self._alloc(n)
        kw = dict(rows_per_group=[h * wd for h, wd in self.hw], out_row0=self.starts)
        try:
            ext.value_proj_bf16(self.rows, w, gb, out, out_group_rows=self.group_rows, out_scale=scale, **kw)
        except ext.OccAmdUnsupported:
            if out.dtype != torch.int16:
                raise
            # q16 rows exist on the activation-resident projection only: other shapes project to fp32 rows and encode
            tmp = torch.empty((self.bs * self.num_cam * self.total, n), dtype=torch.float32, device=out.device)
            ext.value_proj_bf16(self.rows, w, gb, tmp, out_group_rows=self.total, **kw)
            out = ext.sca_rows_encode_q16(tmp.view(self.bs * self.num_cam, self.total, n), scale).view(-1, n)
        return out.view(self.bs * self.num_cam, self.group_rows, n)

    def project(self, value_proj):
        """value_proj(feat + embeds) for every camera pixel -> (bs*num_cam, sum hw, N) fp32 (OCC_SCA_VALUES=f32), or
        (bs*num_cam, sum hw rounded up to even, N) fp16 in the gather's pixel-pair order (ext.sca_pair_layout)."""
        hit = getattr(self, '_pending', {}).pop(id(value_proj), None)
        if hit is not None:
            return hit
        return self._launch(value_proj, self._group_bias(value_proj))



@TRANSFORMER.register_module()
class TransformerOcc(BaseModule):

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 decoder=None, embed_dims=256, rotate_prev_bev=True, use_shift=True, use_can_bus=True,
                 can_bus_norm=True, use_cams_embeds=True, use_3d=False, use_conv=False,
                 rotate_center=[100, 100], num_classes=18, out_dim=32, pillar_h=16,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BN', ),
                 norm_cfg_3d=dict(type='BN3d', ), **kwargs):
        super().__init__(**kwargs)
        self.encoder = build_transformer_layer_sequence(encoder)
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.rotate_prev_bev = rotate_prev_bev
        self.use_shift = use_shift          # accepted, unused: this variant applies no ego shift
        self.use_can_bus = use_can_bus      # accepted, unused: no can-bus MLP in this variant
        self.can_bus_norm = can_bus_norm
        self.use_cams_embeds = use_cams_embeds
        self.use_3d = use_3d
        self.use_conv = use_conv
        self.pillar_h = pillar_h
        self.out_dim = out_dim
        if not use_3d:
            if use_conv:
                use_bias = norm_cfg is None
                self.decoder = nn.Sequential(
                    ConvModule(embed_dims, embed_dims, kernel_size=3, stride=1, padding=1,
                               bias=use_bias, norm_cfg=norm_cfg, act_cfg=act_cfg),
                    ConvModule(embed_dims, embed_dims * 2, kernel_size=3, stride=1, padding=1,
                               bias=use_bias, norm_cfg=norm_cfg, act_cfg=act_cfg))
            else:
                self.decoder = nn.Sequential(nn.Linear(embed_dims, embed_dims * 2), nn.Softplus(),
                                             nn.Linear(embed_dims * 2, embed_dims * 2))
        else:
            use_bias_3d = norm_cfg_3d is None
            self.middle_dims = embed_dims // pillar_h
            self.decoder = nn.Sequential(
                ConvModule(self.middle_dims, out_dim, kernel_size=3, stride=1, padding=1,
                           bias=use_bias_3d, conv_cfg=dict(type='Conv3d'), norm_cfg=norm_cfg_3d,
                           act_cfg=act_cfg),
                ConvModule(out_dim, out_dim, kernel_size=3, stride=1, padding=1, bias=use_bias_3d,
                           conv_cfg=dict(type='Conv3d'), norm_cfg=norm_cfg_3d, act_cfg=act_cfg))
        # X3Linear: nn.Linear whose training forward/backward run on the own kernels (640 000 voxel rows: the weight
        # and bias gradients are the expensive part in ATen)
        self.predicter = nn.Sequential(X3Linear(out_dim, out_dim * 2), nn.Softplus(),
                                       X3Linear(out_dim * 2, num_classes))
        self.flow_predicter = nn.Sequential(X3Linear(out_dim, out_dim * 2), nn.ReLU(),
                                            X3Linear(out_dim * 2, 2))
        self.two_stage_num_proposals = two_stage_num_proposals
        self.init_layers()
        self.rotate_center = rotate_center
        self.use_fused_decoder = True     # flip to force the stock torch (MIOpen) decoder
        self.fuse_heads = os.environ.get("OCC_DECODER_FUSE_HEADS", "1") == "1"   # conv3d_2 + heads + decode in one launch
        self.use_lazy_features = True     # bf16 NHWC maps go straight into the SCA value projection
        # autograd path only: dtype the MIOpen Conv3d decoder runs in under torch.autocast (None = fp32 as the
        # reference).  MIOpen's fp32 Conv3d backward costs 196 ms per step at 200x200x16, bf16 7 ms.
        self.decoder_autocast_dtype = None
        self._dec_key, self._dec_pack = None, None

    def init_layers(self):
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(self.num_cams, self.embed_dims))

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)

    def flatten_features(self, mlvl_feats):
        """list of (bs, num_cam, C, h, w) -> ((bs*num_cam, sum hw, C) with cams/level embeds added,
        spatial_shapes (L,2) int64, level_start_index (L) int64)."""
        bs, num_cam, c = mlvl_feats[0].shape[:3]
        shapes = [(f.shape[3], f.shape[4]) for f in mlvl_feats]
        total = sum(h * w for h, w in shapes)
        # the hot path computes in fp32: a half-precision backbone's maps are widened here, in the same
        # pass that transposes them and adds the (fp32) embeddings
        out = mlvl_feats[0].new_empty((bs * num_cam, total, c), dtype=self.level_embeds.dtype)
        start = 0
        for lvl, feat in enumerate(mlvl_feats):
            h, w = shapes[lvl]
            emb = self.level_embeds[lvl]
            if self.use_cams_embeds:   # (1, num_cam, 1, C) + (C)
                emb = self.cams_embeds[None, :, None, :] + emb
            else:
                emb = emb.view(1, 1, 1, c)
            dst = out[:, start:start + h * w].view(bs, num_cam, h * w, c)
            src = feat.flatten(3).permute(0, 1, 3, 2)
            if torch.is_grad_enabled() and (feat.requires_grad or self.level_embeds.requires_grad):
                dst.copy_(src + emb)            # differentiable (CopySlices)
            else:
                torch.add(src, emb, out=dst)    # transpose + embed add in one pass
            start += h * w
        dev = mlvl_feats[0].device
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=dev)
        spatial_shapes._occ_hw = tuple(tuple(int(v) for v in hw) for hw in shapes)
        starts = [0]
        for h, w in shapes[:-1]:
            starts.append(starts[-1] + h * w)
        level_start_index = torch.as_tensor(starts, dtype=torch.long, device=dev)
        return out, spatial_shapes, level_start_index

    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512],
                         bev_pos=None, prev_bev=None, **kwargs):
        """-> BEV embedding (bs, bev_h*bev_w, C)."""
        _require_device(mlvl_feats[0], 'TransformerOcc')
        bs, num_cam = mlvl_feats[0].shape[:2]
        if bs == 1 and not (torch.is_grad_enabled() and bev_queries.requires_grad):
            bev_queries = bev_queries.unsqueeze(1)      # (nq, 1, C) view: the encoder only reads it
        else:
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
                    tmp = rotate_bev_nearest(tmp, float(rotation_angle), self.rotate_center)
                    prev_bev[:, i] = tmp.permute(1, 2, 0).reshape(bev_h * bev_w, -1)
        if self.use_lazy_features and LazyFeatures.eligible(mlvl_feats):
            # bf16 NHWC maps straight into the SCA value projection (no fp32 flatten buffer)
            feat_flatten = LazyFeatures(self, mlvl_feats)
            spatial_shapes, level_start_index = feat_flatten.spatial_shapes, feat_flatten.level_start_index
        else:
            flat, spatial_shapes, level_start_index = self.flatten_features(mlvl_feats)
            # reference axis order (num_cam, sum hw, bs, C) as a view of the (bs*num_cam, sum hw, C) buffer
            feat_flatten = flat.view(bs, num_cam, flat.shape[1], flat.shape[2]).permute(1, 2, 0, 3)
        out = self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                           bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                           level_start_index=level_start_index, prev_bev=prev_bev, **kwargs)
        # diagnostics (a device tensor, no synchronisation): the fp16 value planes' range scales of this call, max|x| of
        # its maps and the a-priori bounds (ext.value_range_scale) — bench.py's headline_feature_parity reports them
        object.__setattr__(self, 'value_range_report', getattr(feat_flatten, 'range_report_', None))
        return out

    # ------------------------------------------------------------------ fused decoder (inference)
    def _decoder_pack(self):
        """Packed Conv3d weights + eval-mode BatchNorm folded to (scale, shift), cached until a
        parameter or running statistic changes."""
        mods = [self.decoder[0], self.decoder[1]]
        ts = []
        for m in mods:
            ts += [m.conv.weight, m.norm.weight, m.norm.bias, m.norm.running_mean, m.norm.running_var]
            if m.conv.bias is not None:
                ts.append(m.conv.bias)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (cache_epoch(),)
        if key != self._dec_key:
            pack = []
            with torch.no_grad():
                for m in mods:
                    bn = m.norm
                    scale = (bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps))
                    shift = bn.bias.float() - bn.running_mean.float() * scale
                    if m.conv.bias is not None:
                        shift = shift + m.conv.bias.float() * scale
                    pack.append((ext.conv3d_pack_weight(m.conv.weight.detach().float().contiguous()),
                                 scale.contiguous(), shift.contiguous()))
            self._dec_key, self._dec_pack = key, pack
        return self._dec_pack

    def _heads_pack(self):
        p, f = self.predicter, self.flow_predicter
        ts = (p[0].weight, p[0].bias, p[2].weight, p[2].bias, f[0].weight, f[0].bias, f[2].weight, f[2].bias)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (cache_epoch(),)
        if key != getattr(self, '_hp_key', None):
            with torch.no_grad():
                self._hp_key, self._hp_val = key, ext.conv3d_heads_pack(*[t.detach().float().contiguous() for t in ts])
        return self._hp_val

    def _fused_decoder_ok(self, bev):
        if not (self.use_fused_decoder and self.use_3d and bev.is_cuda and bev.dtype == torch.float32):
            return False
        if torch.is_grad_enabled() and (bev.requires_grad or
                                        any(p.requires_grad for p in self.decoder.parameters())):
            return False
        for m in (self.decoder[0], self.decoder[1]):
            bn = m.norm
            if not isinstance(bn, nn.BatchNorm3d) or bn.training or bn.running_mean is None \
                    or not bn.affine or not isinstance(getattr(m, 'activate', None), nn.ReLU):
                return False
            c = m.conv
            if (c.kernel_size, c.stride, c.padding, c.dilation, c.groups) != \
                    ((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1):
                return False
        for head, act in ((self.predicter, nn.Softplus), (self.flow_predicter, nn.ReLU)):
            if len(head) != 3 or not isinstance(head[1], act):
                return False
        sp = self.predicter[1]
        return sp.beta == 1 and sp.threshold == 20

    def _fused_decoder(self, bev, bev_h, bev_w):
        """bev (bs, bev_h*bev_w, C) contiguous -> occ (bs, W, H, Z, num_classes), flow (bs, W, H, Z, 2):
        lifter view + 2x(Conv3d+BN+ReLU) + permute + both MLP heads as three HIP launches."""
        (w1, s1, t1), (w2, s2, t2) = self._decoder_pack()
        Z = self.pillar_h
        x = ext.conv3d_bn_relu(bev, w1, s1, t1, Z, bev_h, bev_w, self.middle_dims, self.out_dim,
                               in_layout=1)
        p, f = self.predicter, self.flow_predicter
        if (self.fuse_heads and Z in (16, 32) and self.out_dim == 32 and w2.dtype == torch.int16
                and ext.HEADS_PRECISION == "bf16x3" and tuple(p[0].weight.shape) == (64, 32)
                and tuple(f[0].weight.shape) == (64, 32) and tuple(p[2].weight.shape)[1] == 64
                and tuple(f[2].weight.shape) == (2, 64)):
            # second convolution + heads + decode in ONE launch: its 82 MB of activations stay on chip
            try:
                occ, flow, cls = ext.conv3d_heads_decode(x, w2, s2, t2, self._heads_pack(), Z, bev_h, bev_w,
                                                         p[2].weight.shape[0])
                occ._occ_cls = cls
                return occ, flow
            except OccAmdUnsupported:
                pass
        x = ext.conv3d_bn_relu(x, w2, s2, t2, Z, bev_h, bev_w, self.out_dim, self.out_dim,
                               in_layout=0, out_xy_major=True)
        occ, flow, cls = ext.occ_heads(x, p[0].weight, p[0].bias, p[2].weight, p[2].bias,
                                       f[0].weight, f[0].bias, f[2].weight, f[2].bias, decode=True)
        # the decoded classes ride on the logits tensor OBJECT (BEVFormerOccHead.get_occ picks them up): the
        # reference's softmax(-1).argmax(-1) would re-read the 43 MB of logits in two more launches
        occ._occ_cls = cls
        return occ, flow

    # training: the two decoder convolutions (forward, dx, dW) on this library's bf16x3 kernels at fp32-class precision
    # (ext.Conv3dX3Function) instead of MIOpen under bf16 autocast; OCC_TRAIN_DECODER=torch keeps the stock modules
    train_decoder_own = os.environ.get("OCC_TRAIN_DECODER", "own") != "torch"

    def _train_decoder_ok(self, bev):
        if not (self.train_decoder_own and self.use_fused_decoder and bev.is_cuda and bev.dtype == torch.float32
                and torch.is_grad_enabled()):
            return False
        for m in (self.decoder[0], self.decoder[1]):
            c = m.conv
            if (not isinstance(m.norm, nn.BatchNorm3d) or not isinstance(getattr(m, 'activate', None), nn.ReLU)
                    or c.bias is not None or (c.kernel_size, c.stride, c.padding, c.dilation, c.groups) !=
                    ((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1) or c.out_channels != 32):
                return False
            # ext.conv3d_autograd's own shape conditions, checked for BOTH convolutions before the first one runs: nothing
            # may raise OccAmdUnsupported after the first BatchNorm has updated its running statistics (the stock-decoder
            # fallback would update them a second time)
            if not (c.in_channels % 16 == 0 or c.in_channels == 8):
                return False
        if self.pillar_h not in (4, 8, 16, 32) or ext.CONV3D_PRECISION != "bf16x3":
            return False
        return True

    @staticmethod
    def _bn_rows(bn, x2d):
        """nn.BatchNorm3d.forward on channels-last rows (N, C): statistics over N = every voxel of the batch."""
        eaf = 0.0 if bn.momentum is None else bn.momentum
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
            eaf = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        keep = not bn.training or bn.track_running_stats
        return F.batch_norm(x2d, bn.running_mean if keep else None, bn.running_var if keep else None, bn.weight,
                            bn.bias, use_batch, eaf, bn.eps)

    def _train_decoder(self, bev, bev_h, bev_w):
        """bev (bs, bev_h*bev_w, C) -> decoder features (bs, W, H, Z, out_dim), differentiable: lifter view + 2 x
        (Conv3d -> BatchNorm3d -> ReLU) + permute (reference transformer_occ.py:305-308)."""
        Z, bs = self.pillar_h, bev.shape[0]
        x = ext.conv3d_autograd(bev, self.decoder[0].conv.weight, Z, bev_h, bev_w, in_layout=1)
        x = torch.relu_(self._bn_rows(self.decoder[0].norm, x.view(-1, x.shape[-1]))).view(bs, bev_h, bev_w, Z, -1)
        x = ext.conv3d_autograd(x, self.decoder[1].conv.weight, Z, bev_h, bev_w, in_layout=0)
        x = torch.relu_(self._bn_rows(self.decoder[1].norm, x.view(-1, x.shape[-1]))).view(bs, bev_h, bev_w, Z, -1)
        return x.permute(0, 2, 1, 3, 4).contiguous()               # (bs, W, H, Z, C)

    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_h, bev_w,
                grid_length=[0.512, 0.512], bev_pos=None, reg_branches=None, cls_branches=None,
                prev_bev=None, **kwargs):
        """-> (bev_embed (bs, C, bev_h, bev_w), occ (bs, W, H, Z, num_classes), flow (bs, W, H, Z, 2))."""
        bev_embed = self.get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w,
                                          grid_length=grid_length, bev_pos=bev_pos,
                                          prev_bev=prev_bev, **kwargs)
        bs = mlvl_feats[0].size(0)
        if self._fused_decoder_ok(bev_embed):
            try:
                occ_pred, flow_pred = self._fused_decoder(bev_embed.contiguous(), bev_h, bev_w)
                return bev_embed.permute(0, 2, 1).view(bs, -1, bev_h, bev_w), occ_pred, flow_pred
            except OccAmdUnsupported:
                pass
        if self.use_3d and self._train_decoder_ok(bev_embed):
            try:
                outputs = self._train_decoder(bev_embed.contiguous(), bev_h, bev_w)
                flow_pred = self.flow_predicter(outputs)
                occ_pred = self.predicter(outputs)
                return bev_embed.permute(0, 2, 1).view(bs, -1, bev_h, bev_w), occ_pred, flow_pred
            except OccAmdUnsupported:
                pass
        bev_embed = bev_embed.permute(0, 2, 1).view(bs, -1, bev_h, bev_w)
        if self.use_3d:
            # lifter: channel c -> (feature c // pillar_h, height c % pillar_h): a free view
            lifted = bev_embed.view(bs, -1, self.pillar_h, bev_h, bev_w)
            if self.decoder_autocast_dtype is not None and lifted.is_cuda:
                with torch.autocast('cuda', dtype=self.decoder_autocast_dtype):
                    outputs = self.decoder(lifted)
                outputs = outputs.float()
            else:
                outputs = self.decoder(lifted)
            outputs = outputs.permute(0, 4, 3, 2, 1)
        elif self.use_conv:
            outputs = self.decoder(bev_embed)
            outputs = outputs.view(bs, -1, self.pillar_h, bev_h, bev_w).permute(0, 3, 4, 2, 1)
        else:
            outputs = self.decoder(bev_embed.permute(0, 2, 3, 1))
            outputs = outputs.view(bs, bev_h, bev_w, self.pillar_h, self.out_dim)
        flow_pred = self.flow_predicter(outputs)
        occ_pred = self.predicter(outputs)
        return bev_embed, occ_pred, flow_pred
