"""SpatialCrossAttention / MSDeformableAttention3D on the MI355X kernels.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/modules/spatial_cross_attention.py:
same registry names, constructor kwargs, parameter names (state_dict layout) and forward() call
contract.  Two execution paths, both HIP:

* fused (inference, the hot path): the camera-independent query Linears run ONCE per BEV query
  (the reference recomputes them on every padded per-camera copy of the query), the gather kernel
  `occ_sca_fused_forward_f32` replaces rebatch + softmax + location arithmetic + deformable attention
  + scatter-add + camera-count divide (reference :136-173, :338-396).
* unfused (autograd / shapes without a fused kernel): the reference's own decomposition —
  rebatch, MSDeformableAttention3D, scatter — with the deformable attention going through
  MultiScaleDeformableAttnFunction_fp32 (the operator boundary).

There is no CPU branch: host tensors raise (the reference's CPU selector at :386-396 is restated
only in oracle/).
"""
import math
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import cache_epoch, ext
from .._lib import OccAmdError, OccAmdUnsupported
from .bricks import BaseModule, X3Linear, constant_init, xavier_init
from .functions import MultiScaleDeformableAttnFunction_fp32
from .registry import ATTENTION, build_attention


def _require_device(t, who):
    if not t.is_cuda:
        raise OccAmdError(f"{who}: host tensor given — the MI355X path has no CPU fallback "
                          "(the CPU restatement lives in oracle/ and is test infrastructure)")


class _CatLinearCache:
    """Concatenated (weight, bias) of several nn.Linear layers, rebuilt when a parameter changes."""

    def __init__(self):
        self._key, self._w, self._b = None, None, None

    def get(self, linears):
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version)
                    for l in linears) + (cache_epoch(),)
        if key != self._key:
            self._w = torch.cat([l.weight.detach() for l in linears], 0).contiguous()
            self._b = torch.cat([l.bias.detach() for l in linears], 0).contiguous()
            self._key = key
        return self._w, self._b


@ATTENTION.register_module()
class MSDeformableAttention3D(BaseModule):
    """Deformable attention over the z-anchor reference points of each BEV pillar
    (reference :178-400).  No output projection, no residual (output_proj=None, :221)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn("the fused gfx950 gather kernels need 32 channels per head; other head "
                          "sizes run the generic kernel")
        self.norm_cfg = norm_cfg
        self.batch_first = batch_first
        self.output_proj = None
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = X3Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = X3Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = X3Linear(embed_dims, embed_dims)
        self._qcat = _CatLinearCache()
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid_init.view(-1)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)  # None -> no-op, as in mmcv
        self._is_init = True

    def query_linears(self, query):
        """Both query-side Linears as one GEMM -> (offsets, logits) column views."""
        w, b = self._qcat.get((self.sampling_offsets, self.attention_weights))
        out = F.linear(query, w, b)
        n_off = self.sampling_offsets.out_features
        return out[..., :n_off], out[..., n_off:]

    def query_linears_autograd(self, query):
        """Training form of query_linears: sampling_offsets and attention_weights as ONE differentiable GEMM
        (the concatenation is part of the graph, so both layers receive their gradients) -> (…, n_off + n_att)."""
        w = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0)
        b = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0)
        w._occ_no_cache = True                  # rebuilt every forward: its packed form must not pile up in the cache
        return ext.linear_autograd(query, w, b)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, query_proj=None, sampling=None, **kwargs):
        """Unfused form, reference semantics: query (bs, num_query, C), value (bs, num_value, C),
        reference_points (bs, num_query, Z, 2) -> (bs, num_query, C).  `query_proj` (bs, num_query, n_off + n_att):
        the two query-side Linears already applied (SpatialCrossAttention computes them once per BEV query and
        rebatches the RESULT per camera instead of the input — a Linear is row-wise, so that is the same function
        with 40 000 instead of 6 x 9 900 GEMM rows); `query` is then ignored.  `sampling` = (sampling_locations,
        attention_weights) already prepared (ext.SCAPrepFunction): only the value projection and the operator run."""
        if sampling is not None:
            if not self.batch_first:
                value = value.permute(1, 0, 2)
            _require_device(value, 'MSDeformableAttention3D')
            bs, num_value, _ = value.shape
            value = self.value_proj(value)
            if key_padding_mask is not None:
                value = value.masked_fill(key_padding_mask[..., None], 0.0)
            value = value.view(bs, num_value, self.num_heads, -1)
            output = MultiScaleDeformableAttnFunction_fp32.apply(
                value, spatial_shapes, level_start_index, sampling[0], sampling[1], self.im2col_step)
            return output if self.batch_first else output.permute(1, 0, 2)
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None and query_proj is None:
            query = query + query_pos
        if not self.batch_first:
            query = None if query_proj is not None else query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        _require_device(value, 'MSDeformableAttention3D')
        bs, num_query, _ = (query_proj if query_proj is not None else query).shape
        bs, num_value, _ = value.shape
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        if query_proj is not None:
            n_off = self.sampling_offsets.out_features
            sampling_offsets = query_proj[..., :n_off].reshape(
                bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
            attention_weights = query_proj[..., n_off:].reshape(
                bs, num_query, self.num_heads, self.num_levels * self.num_points)
        else:
            sampling_offsets = self.sampling_offsets(query).view(
                bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
            attention_weights = self.attention_weights(query).view(
                bs, num_query, self.num_heads, self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points)
        if reference_points.shape[-1] != 2:
            raise ValueError(f'Last dim of reference_points must be 2, '
                             f'but get {reference_points.shape[-1]} instead.')
        # every query owns Z anchors; point p of a level samples around anchor p % Z
        offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        Z = reference_points.shape[2]
        if self.num_points % Z:
            raise ValueError(f'num_points ({self.num_points}) must be a multiple of the number of '
                             f'z-anchors ({Z})')
        sampling_offsets = sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        sampling_offsets = sampling_offsets.view(bs, num_query, self.num_heads, self.num_levels,
                                                 self.num_points // Z, Z, 2)
        sampling_locations = reference_points[:, :, None, None, None, :, :] + sampling_offsets
        sampling_locations = sampling_locations.view(bs, num_query, self.num_heads, self.num_levels,
                                                     self.num_points, 2)
        output = MultiScaleDeformableAttnFunction_fp32.apply(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            self.im2col_step)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return output


def pack_vis_bits(bev_mask):
    """bev_mask (num_cams, bs, Nq, Z) bool -> (bs, Nq) int32, bit c = query visible in camera c."""
    any_z = bev_mask.any(-1)                                         # (NC, bs, Nq)
    w = (1 << torch.arange(any_z.shape[0], device=any_z.device, dtype=torch.int64)).view(-1, 1, 1)
    return (any_z.to(torch.int64) * w).sum(0).to(torch.int32).contiguous()


@ATTENTION.register_module()
class SpatialCrossAttention(BaseModule):
    """Each BEV query attends, through MSDeformableAttention3D, to the cameras its pillar projects
    into; the per-camera results are averaged over the visible cameras (reference :31-175)."""

    supports_post_norm_train = True     # forward(post_norm_train=LayerNorm) -> (output, norm applied?) (encoder.py)

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False,
                 deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=256,
                                           num_levels=4),
                 **kwargs):
        super().__init__(init_cfg)
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = X3Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.use_fused = True          # flip to force the unfused (reference-shaped) HIP path
        # optional int64[2] device tensor: the fused gather adds (visible rows, in-map corners) of every launch to it
        # (bench.py's roofline leg, tests) — an attribute, so it reaches the gather on every fused path
        self.gather_stats = None
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    # ------------------------------------------------------------------ fused inference path
    def _fused_slots(self, query, value, reference_points_cam, bev_mask, spatial_shapes,
                     level_start_index, vis_bits=None, order=None, stats=None):
        da = self.deformable_attention
        num_cams, l, bs, _ = value.shape
        # (num_cams, l, bs, C) -> (bs*num_cams, l, C): a view when the producer laid it out that way
        v = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        v = da.value_proj(v.float()).view(bs * self.num_cams, l, da.num_heads, -1)
        offs, logits = da.query_linears(query.float())
        if vis_bits is None:
            vis_bits = pack_vis_bits(bev_mask)
        if stats is None:
            stats = self.gather_stats
        return ext.sca_fused_forward(v, spatial_shapes, level_start_index, offs, logits,
                                     reference_points_cam.float().contiguous(), vis_bits,
                                     da.num_heads, da.num_levels, da.num_points, order=order,
                                     stats=stats)

    def query_linear_operands(self):
        """(weight (n_off + n_att, C), bias) of the two query-side Linears as one GEMM, or None when the deformable
        attention is not the MSDeformableAttention3D the fused gather implements."""
        da = self.deformable_attention
        if not isinstance(da, MSDeformableAttention3D):
            return None
        return da._qcat.get((da.sampling_offsets, da.attention_weights))

    def fused_gather(self, lin, value, reference_points_cam=None, bev_mask=None, spatial_shapes=None,
                     level_start_index=None, vis_bits=None, bev_order=None, gather_stats=None):
        """Value projection + fused SCA gather for the query-side Linear outputs `lin` (bs, nq, n_off + n_att)
        -> slots (bs, nq, C) BEFORE output_proj.  Raises OccAmdUnsupported."""
        da = self.deformable_attention
        if gather_stats is None:
            gather_stats = self.gather_stats
        layout, vscale = "rows", None
        if hasattr(value, 'project'):      # LazyFeatures: bf16 NHWC maps, projected level by level
            bs = value.bs
            v = value.project(da.value_proj)
            vscale = value.value_scale(da.value_proj)                   # fp16 rows: the plane's range scale (exact)
            layout = "pairs" if v.dtype in (torch.float16, torch.int16) else "rows"    # the 16-bit epilogues write pairs
        else:
            num_cams, l, bs, _ = value.shape
            v = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
            v = ext.linear(v, da.value_proj.weight, da.value_proj.bias)
            if ext.SCA_VALUES == "f16":
                v, vscale = ext.f16_range_scaled(v)                     # power-of-two scale: no fp16 saturation
            elif ext.SCA_VALUES == "q16":
                v, vscale = ext.q16_range_scaled(v)                     # block floating point, pixel-pair order
                layout = "pairs"
        v = v.view(bs * self.num_cams, v.shape[1], da.num_heads, -1)
        n_off = da.sampling_offsets.out_features
        if vis_bits is None:
            vis_bits = pack_vis_bits(bev_mask)
        return ext.sca_fused_forward(v, spatial_shapes, level_start_index, lin[..., :n_off],
                                     lin[..., n_off:], reference_points_cam.float().contiguous(),
                                     vis_bits, da.num_heads, da.num_levels, da.num_points,
                                     order=bev_order, stats=gather_stats, value_layout=layout, value_scale=vscale)

    def forward_fused(self, query, value, reference_points_cam=None, bev_mask=None,
                      spatial_shapes=None, level_start_index=None, vis_bits=None, bev_order=None,
                      gather_stats=None, post_norm=None):
        """Inference form with every dense op on the MFMA Linear kernel (value_proj over all camera
        pixels, the query-side Linears, output_proj + residual + the layer's following LayerNorm as one
        epilogue) around the fused gather.  -> LayerNorm(output_proj(slots) + query), or None when a
        shape has no fused kernel."""
        wb = self.query_linear_operands()
        if wb is None:
            return None
        try:
            lin = ext.linear(query.contiguous(), wb[0], wb[1])
            slots = self.fused_gather(lin, value, reference_points_cam, bev_mask, spatial_shapes,
                                      level_start_index, vis_bits, bev_order, gather_stats)
            return ext.linear(slots, self.output_proj.weight, self.output_proj.bias,
                              residual=query.contiguous(), ln=post_norm)
        except OccAmdUnsupported:
            return None

    # ------------------------------------------------------------------ unfused path
    # False (OCC_SCA_TRAIN_REBATCH=reference): the reference's literal order (rebatch the queries, then the Linears)
    rebatch_projected = os.environ.get("OCC_SCA_TRAIN_REBATCH", "projected") != "reference"
    # the two row shuffles of that path on ext.rows_gather_sum (OCC_SCA_TRAIN_ROWS=torch: index_select / index_add_)
    rebatch_kernel = os.environ.get("OCC_SCA_TRAIN_ROWS", "kernel") != "torch"
    # rebatch + softmax + offset normalisation + anchor add as one kernel (OCC_SCA_TRAIN_PREP=torch: ATen ops)
    prep_kernel = os.environ.get("OCC_SCA_TRAIN_PREP", "kernel") != "torch"

    def _rebatch_plan(self, bev_mask, reference_points_cam):
        """Visible-query lists of every camera from batch element 0's mask (reference :138-140) as ONE padded index
        tensor, built once per mask TENSOR (the encoder hands the same bev_mask to all layers; nonzero() is a host
        sync): idx (num_cams * max_len,) BEV query per padded row (clamped), valid (num_cams * max_len, 1) 0/1,
        ref (bs, num_cams, max_len, Z, 2) the rebatched reference points (zeros on padded rows, as the reference)."""
        plan = getattr(bev_mask, '_occ_rebatch', None)
        if plan is not None:
            return plan
        indexes = getattr(bev_mask, '_occ_indexes', None)
        if indexes is None:
            indexes = [m[0].sum(-1).nonzero().squeeze(-1) for m in bev_mask]
        max_len = max(len(each) for each in indexes)
        nc = len(indexes)
        dev = bev_mask.device
        idx = torch.zeros(nc, max_len, dtype=torch.long, device=dev)
        valid = torch.zeros(nc, max_len, dtype=torch.float32, device=dev)
        for i, each in enumerate(indexes):
            idx[i, :len(each)] = each
            valid[i, :len(each)] = 1.0
        cam = torch.arange(nc, device=dev).view(nc, 1).expand(nc, max_len)
        # reference_points_cam (num_cams, bs, Q, Z, 2) -> (bs, num_cams, max_len, Z, 2)
        ref = reference_points_cam[cam, :, idx].permute(2, 0, 1, 3, 4) * valid.view(1, nc, max_len, 1, 1)
        # gather maps for ext.RowsGatherSumFunction: row_to_query (rows, 1) with -1 on padded rows, and its inverse
        # query_to_rows (Q, Kmax): the <= Kmax padded rows that hold a BEV query (-1 = none), in camera order
        num_query = bev_mask.shape[2]
        rows = torch.nonzero(valid.view(-1) > 0).squeeze(-1)                 # valid padded rows, ascending
        q_of = idx.view(-1)[rows]
        cnt = torch.bincount(q_of, minlength=num_query)
        kmax = max(int(cnt.max()), 1)
        order = torch.argsort(q_of, stable=True)                             # camera order inside a query
        sq = q_of[order]
        start = torch.cumsum(cnt, 0) - cnt
        pos = torch.arange(sq.numel(), device=dev) - start[sq]
        query_to_rows = torch.full((num_query, kmax), -1, dtype=torch.long, device=dev)
        query_to_rows[sq, pos] = rows[order]
        row_to_query = torch.where(valid.view(-1) > 0, idx.view(-1), torch.full_like(idx.view(-1), -1)).view(-1, 1)
        plan = dict(indexes=indexes, max_len=max_len, idx=idx.view(-1), valid=valid.view(-1, 1),
                    ref=ref.contiguous(), row_to_query=row_to_query.contiguous(),
                    query_to_rows=query_to_rows.contiguous())
        try:
            bev_mask._occ_indexes = indexes
            bev_mask._occ_rebatch = plan
        except AttributeError:
            pass
        return plan

    def _unfused_slots(self, query, key, value, reference_points_cam, bev_mask, spatial_shapes,
                       level_start_index):
        bs, num_query, _ = query.size()
        D = reference_points_cam.size(3)
        plan = self._rebatch_plan(bev_mask, reference_points_cam)
        indexes, max_len = plan['indexes'], plan['max_len']
        num_cams, l, bs, embed_dims = key.shape
        key = key.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        value = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        da = self.deformable_attention
        if self.rebatch_projected and hasattr(da, 'query_linears_autograd') and query.is_cuda:
            # the two query-side Linears once per BEV query; their OUTPUT rows are then dealt to the cameras
            # (index_select + padding mask: two launches instead of 12 indexed copies, and a third fewer GEMM rows)
            proj = da.query_linears_autograd(query)                                     # (bs, Q, n_off + n_att)
            gather = self.rebatch_kernel and proj.dtype == torch.float32
            prep = (gather and self.prep_kernel and (da.num_heads, da.num_levels, da.num_points) == (8, 4, 8)
                    and da.num_points % D == 0 and True)
            if prep:
                # rebatch + softmax + offset normalisation + anchor add as ONE kernel (and one in backward)
                loc, att = ext.SCAPrepFunction.apply(proj.contiguous(), plan['row_to_query'], plan['query_to_rows'],
                                                     plan['ref'].view(bs, self.num_cams * max_len, D, 2),
                                                     spatial_shapes, da.num_heads, da.num_levels, da.num_points)
                shp = (bs * self.num_cams, max_len, da.num_heads, da.num_levels, da.num_points)
                queries = da(query=None, key=key, value=value, sampling=(loc.view(*shp, 2), att.view(*shp)),
                             spatial_shapes=spatial_shapes, level_start_index=level_start_index)
            else:
                if gather:  # one copy kernel; its gradient is the gather-sum over the inverse map (no float atomics)
                    proj_rb = ext.RowsGatherSumFunction.apply(proj, plan['row_to_query'], plan['query_to_rows'])
                else:
                    proj_rb = proj.index_select(1, plan['idx']) * plan['valid']         # (bs, cams * max_len, .)
                queries = da(query=None, key=key, value=value,
                             query_proj=proj_rb.view(bs * self.num_cams, max_len, proj.shape[-1]),
                             reference_points=plan['ref'].view(bs * self.num_cams, max_len, D, 2),
                             spatial_shapes=spatial_shapes, level_start_index=level_start_index)
            if gather and queries.dtype == torch.float32:
                slots = ext.RowsGatherSumFunction.apply(
                    queries.view(bs, self.num_cams * max_len, self.embed_dims), plan['query_to_rows'],
                    plan['row_to_query'])
            else:
                slots = query.new_zeros(bs, num_query, self.embed_dims).index_add_(
                    1, plan['idx'], queries.view(bs, self.num_cams * max_len, self.embed_dims) * plan['valid'])
        else:
            slots = torch.zeros_like(query)
            queries_rebatch = query.new_zeros([bs, self.num_cams, max_len, self.embed_dims])
            reference_points_rebatch = reference_points_cam.new_zeros([bs, self.num_cams, max_len, D, 2])
            for i, idx in enumerate(indexes):
                queries_rebatch[:, i, :len(idx)] = query[:, idx]
                reference_points_rebatch[:, i, :len(idx)] = reference_points_cam[i][:, idx]
            queries = da(
                query=queries_rebatch.view(bs * self.num_cams, max_len, self.embed_dims), key=key,
                value=value,
                reference_points=reference_points_rebatch.view(bs * self.num_cams, max_len, D, 2),
                spatial_shapes=spatial_shapes, level_start_index=level_start_index).view(
                    bs, self.num_cams, max_len, self.embed_dims)
            for i, idx in enumerate(indexes):
                slots[:, idx] += queries[:, i, :len(idx)]
        count = (bev_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1)
        count = torch.clamp(count, min=1.0)
        return slots / count[..., None]

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag='encoder', **kwargs):
        """query (bs, num_query, C) [batch-first, as the encoder calls it]; key/value
        (num_cams, num_value, bs, C); reference_points_cam (num_cams, bs, num_query, Z, 2);
        bev_mask (num_cams, bs, num_query, Z) -> (bs, num_query, C)."""
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        _require_device(query, 'SpatialCrossAttention')
        needs_grad = torch.is_grad_enabled() and (
            query.requires_grad or value.requires_grad or
            any(p.requires_grad for p in self.deformable_attention.parameters()))
        slots = None
        if self.use_fused and not needs_grad and key_padding_mask is None:
            try:
                slots = self._fused_slots(query, value, reference_points_cam, bev_mask,
                                          spatial_shapes, level_start_index,
                                          vis_bits=kwargs.get('vis_bits'),
                                          order=kwargs.get('sca_bev_order', kwargs.get('bev_order')),
                                          stats=kwargs.get('gather_stats'))
            except OccAmdUnsupported:
                slots = None
        if slots is None:
            slots = self._unfused_slots(query, key, value, reference_points_cam, bev_mask,
                                        spatial_shapes, level_start_index)
        slots = self.output_proj(slots)
        norm = kwargs.get('post_norm_train')
        if norm is not None and ext.dropout_add_layernorm_ok(slots, inp_residual, norm):
            # training: dropout + residual + the layer's following LayerNorm as one autograd node (the layer skips its norm)
            return ext.dropout_add_layernorm(slots, inp_residual, norm, self.dropout.p, self.training), True
        out = self.dropout(slots) + inp_residual
        return (out, False) if norm is not None else out
