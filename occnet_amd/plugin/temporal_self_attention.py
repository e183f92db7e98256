"""TemporalSelfAttention on the MI355X kernels.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/modules/temporal_self_attention.py
(registry name, constructor kwargs, parameter names, forward contract).  BEV self-attention over a
2-deep queue {history BEV (or the current BEV again), current BEV}, one level, `num_points` samples
per head and queue entry; the two queue results are averaged, projected and added to the identity.

* fused path (inference): both query Linears as one GEMM over cat([value[:bs], query+pos]),
  `occ_tsa_fused_forward_f32` does softmax + location arithmetic + gather + queue mean
  (reference :206-262).  Without history the reference stacks the same tensor twice and projects it
  twice (:177, :198); here it is projected once and both queue entries alias it.
* unfused path (autograd / unsupported shapes): the reference's decomposition through
  MultiScaleDeformableAttnFunction_fp32.
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import cache_epoch, ext
from .._lib import OccAmdUnsupported
from .bricks import BaseModule, X3Linear, constant_init, xavier_init
from .functions import MultiScaleDeformableAttnFunction_fp32
from .registry import ATTENTION
from .spatial_cross_attention import _CatLinearCache, _require_device


@ATTENTION.register_module()
class TemporalSelfAttention(BaseModule):

    supports_post_norm_train = True     # forward(post_norm_train=LayerNorm) -> (output, norm applied?) (encoder.py)

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2,
                 im2col_step=64, dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn("the fused gfx950 gather kernels need 32 channels per head; other head "
                          "sizes run the generic kernel")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_bev_queue = num_bev_queue
        self.sampling_offsets = X3Linear(
            embed_dims * num_bev_queue, num_bev_queue * num_heads * num_levels * num_points * 2)
        self.attention_weights = X3Linear(
            embed_dims * num_bev_queue, num_bev_queue * num_heads * num_levels * num_points)
        self.value_proj = X3Linear(embed_dims, embed_dims)
        self.output_proj = X3Linear(embed_dims, embed_dims)
        self._qcat = _CatLinearCache()
        self.use_fused = True
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels * self.num_bev_queue,
                                            self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid_init.view(-1)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def _fused(self, query_cat, value, shared, reference_points, bev_h, bev_w, order):
        bs, num_query, _ = query_cat.shape
        w, b = self._qcat.get((self.sampling_offsets, self.attention_weights))
        lin = F.linear(query_cat, w, b)
        n_off = self.sampling_offsets.out_features
        v = self.value_proj(value)
        v = v.view(v.shape[0], num_query, self.num_heads, -1)
        return ext.tsa_fused_forward(v, lin[..., :n_off], lin[..., n_off:],
                                     reference_points.float().contiguous(), bev_h, bev_w,
                                     self.num_heads, self.num_points, shared_queue=shared,
                                     order=order)

    def _folded_query_weights(self, w, b, query_pos):
        c = self.embed_dims
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version, query_pos.data_ptr(),
               query_pos._version, tuple(query_pos.shape), cache_epoch())
        if getattr(self, '_fold_key', None) != key:
            w_sum = (w[:, :c] + w[:, c:]).contiguous()
            pos_term = ext.linear(query_pos.contiguous(), w[:, c:].contiguous(), b)
            # the keyed tensors stay referenced so their addresses cannot be recycled under the cache
            self._fold_key, self._fold_src, self._fold_val = key, (w, b, query_pos), (w_sum, pos_term)
        return self._fold_val

    def _fusable(self, reference_points):
        return (self.batch_first and self.num_levels == 1 and self.num_bev_queue == 2
                and reference_points is not None and reference_points.shape[-1] == 2)

    def fused_gather(self, query, value=None, query_pos=None, reference_points=None, bev_h=None, bev_w=None,
                     bev_order=None, pre=None):
        """The gather half of forward_fused: both query Linears, the value projection and the fused TSA gather
        -> (bs, num_query, C) BEFORE output_proj.  `pre` = (lin, v): the two Linear outputs already computed by the
        previous layer's chain kernel (ext.encoder_ffn_chain tail; no-history case only).  Raises OccAmdUnsupported."""
        bs, num_query, c = query.shape
        shared = value is None
        if shared and bs > 1:   # interleaved (b0,b0,b1,b1,..) stack, as in forward()
            value = torch.stack([query, query], 1).reshape(bs * 2, num_query, c)
            shared = False
        n_off = self.sampling_offsets.out_features
        if pre is not None and shared:
            lin, v = pre
        else:
            value_first = query if shared else value[:bs]
            w, b = self._qcat.get((self.sampling_offsets, self.attention_weights))
            if shared and query_pos is not None:
                # no history: cat([q, q + pos]) @ W^T + b = q @ (Wa + Wb)^T + (pos @ Wb^T + b); the
                # position term is constant while weights and positional encoding are (inference)
                w_sum, pos_term = self._folded_query_weights(w, b, query_pos)
                if ext.LINEAR_CHAIN and ext.LINEAR_PRECISION == "bf16x3" and bs == 1:
                    try:        # both Linears of the same rows in one launch (program C of csrc/linear_chain_x3.hip)
                        lin, v = ext.linear_pair_chain(query.contiguous(), w_sum, pos_term, self.value_proj.weight,
                                                       self.value_proj.bias)
                    except OccAmdUnsupported:
                        lin = None
                    if lin is not None:
                        v = v.view(v.shape[0], num_query, self.num_heads, -1)
                        return ext.tsa_fused_forward(v, lin[..., :n_off], lin[..., n_off:],
                                                     reference_points.float().contiguous(), bev_h, bev_w,
                                                     self.num_heads, self.num_points, shared_queue=shared,
                                                     order=bev_order)
                lin = ext.linear(query.contiguous(), w_sum, None, residual=pos_term)
            else:
                lin = ext.linear(value_first.contiguous(), w, b, a2=query.contiguous(),
                                 a2_add=None if query_pos is None else query_pos.contiguous())
            vsrc = (value_first if shared else value).contiguous()
            v = ext.linear(vsrc, self.value_proj.weight, self.value_proj.bias)
        v = v.view(v.shape[0], num_query, self.num_heads, -1)
        return ext.tsa_fused_forward(v, lin[..., :n_off], lin[..., n_off:],
                                     reference_points.float().contiguous(), bev_h, bev_w,
                                     self.num_heads, self.num_points, shared_queue=shared,
                                     order=bev_order)

    def chain_tail(self, query_pos):
        """Operands with which the PREVIOUS layer's chain kernel computes this layer's query Linears and value
        projection (no history, bs = 1): (w_sum (n, C), pos_term (bs, nq, n), value_proj.weight, value_proj.bias)."""
        w, b = self._qcat.get((self.sampling_offsets, self.attention_weights))
        w_sum, pos_term = self._folded_query_weights(w, b, query_pos)
        return w_sum, pos_term, self.value_proj.weight, self.value_proj.bias

    def forward_fused(self, query, value=None, query_pos=None, reference_points=None, bev_h=None,
                      bev_w=None, bev_order=None, post_norm=None):
        """Inference form with every dense op on the MFMA Linear kernel: the cat([value, query+pos])
        feeding the offset/weight Linears is read as two K segments, output_proj + residual + the
        layer's following LayerNorm are one epilogue.  Same arguments/semantics as forward() (value
        None = no history).  -> LayerNorm(output_proj(attn) + query), or None when a shape has no
        fused kernel (the caller then takes forward())."""
        if not self._fusable(reference_points):
            return None
        try:
            out = self.fused_gather(query, value, query_pos, reference_points, bev_h, bev_w, bev_order)
            return ext.linear(out, self.output_proj.weight, self.output_proj.bias,
                              residual=query.contiguous(), ln=post_norm)
        except OccAmdUnsupported:
            return None

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag='decoder', **kwargs):
        """query (bs, num_query, C); value None (no history: the current BEV twice) or
        (bs*2, num_query, C) = stack([prev_bev, bev_query], 1); reference_points
        (bs*2, num_query, num_levels, 2) -> (bs, num_query, C)."""
        _require_device(query, 'TemporalSelfAttention')
        shared = value is None
        if shared:
            assert self.batch_first
            bs, len_bev, c = query.shape
            if bs > 1:   # interleaved (b0,b0,b1,b1,..) stack: value[:bs] below is NOT "all prev_bevs"
                value = torch.stack([query, query], 1).reshape(bs * 2, len_bev, c)
                shared = False
        if identity is None:
            identity = query
        value_first = query if shared else None
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            if value is not None:
                value = value.permute(1, 0, 2)
        bs, num_query, embed_dims = query.shape
        assert self.num_bev_queue == 2
        if value_first is None:
            value_first = value[:bs]
        query = torch.cat([value_first, query], -1)

        needs_grad = torch.is_grad_enabled() and (
            query.requires_grad or any(p.requires_grad for p in self.parameters()))
        output = None
        if (self.use_fused and not needs_grad and key_padding_mask is None and
                self.num_levels == 1 and reference_points.shape[-1] == 2):
            bev_h, bev_w = kwargs.get('bev_h'), kwargs.get('bev_w')
            if bev_h is None or bev_w is None:
                bev_h, bev_w = [int(v) for v in spatial_shapes[0].tolist()]   # device sync
            try:
                output = self._fused(query, value_first if shared else value, shared,
                                     reference_points, bev_h, bev_w, kwargs.get('bev_order'))
            except OccAmdUnsupported:
                output = None
        if output is None:
            if shared:
                value = torch.stack([value_first, value_first], 1).reshape(bs * 2, num_query, -1)
            output = self._unfused(query, value, key_padding_mask, reference_points,
                                   spatial_shapes, level_start_index)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        norm = kwargs.get('post_norm_train')
        if norm is not None and ext.dropout_add_layernorm_ok(output, identity, norm):
            # training: dropout + residual + the layer's following LayerNorm as one autograd node (the layer skips its norm)
            return ext.dropout_add_layernorm(output, identity, norm, self.dropout.p, self.training), True
        out = self.dropout(output) + identity
        return (out, False) if norm is not None else out

    def _unfused(self, query, value, key_padding_mask, reference_points, spatial_shapes,
                 level_start_index):
        bs, num_query, _ = query.shape
        embed_dims = self.embed_dims
        num_value = value.shape[1]
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.reshape(bs * self.num_bev_queue, num_value, self.num_heads, -1)
        if query.is_cuda and torch.is_grad_enabled():
            # training: both query-side Linears as ONE differentiable GEMM (forward, dx and dW each once instead of
            # twice); the concatenation is part of the graph, so both layers receive their gradients
            n_off = self.sampling_offsets.out_features
            wcat = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0)
            wcat._occ_no_cache = True           # rebuilt every forward: its packed form must not pile up in the cache
            proj = ext.linear_autograd(
                query, wcat, torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0))
            sampling_offsets = proj[..., :n_off].reshape(
                bs, num_query, self.num_heads, self.num_bev_queue, self.num_levels, self.num_points, 2)
            attention_weights = proj[..., n_off:].reshape(
                bs, num_query, self.num_heads, self.num_bev_queue, self.num_levels * self.num_points)
        else:
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
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f'Last dim of reference_points must be 2 or 4, '
                             f'but get {reference_points.shape[-1]} instead.')
        output = MultiScaleDeformableAttnFunction_fp32.apply(
            value, spatial_shapes, level_start_index, sampling_locations.contiguous(),
            attention_weights, self.im2col_step)
        # (bs*queue, num_query, C) -> mean over the queue -> (bs, num_query, C)
        output = output.view(bs, self.num_bev_queue, num_query, embed_dims).mean(1)
        return output
