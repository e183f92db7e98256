"""Oracle for multi-scale deformable attention (test infrastructure; see oracle/__init__.py).

`multi_scale_deformable_attn_pytorch` restates mmcv-full 1.4.x
`mmcv/ops/multi_scale_deform_attn.py::multi_scale_deformable_attn_pytorch` — the CPU branch the
reference takes at projects/mmdet3d_plugin/bevformer/modules/spatial_cross_attention.py:395-396 and
temporal_self_attention.py:252-253 — from its published algorithm.
`msda_scalar_f64` is an independent re-derivation of the CUDA kernel's per-sample arithmetic
(Appendix B.2), used to cross-check the former.  tests/test_oracle_golden.py additionally holds the restatement bit for
bit against the Hugging Face transformers port of the same function (Deformable-DETR's ms_deform_attn_core_pytorch, which
mmcv vendored) — an outside implementation, though still not a reference-owned vector.
"""
import numpy as np
import torch
import torch.nn.functional as F


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
                                        attention_weights):
    """value (bs, S, M, D); value_spatial_shapes (L, 2) (h, w); sampling_locations
    (bs, Lq, M, L, P, 2) in [0,1] (x, y); attention_weights (bs, Lq, M, L, P) -> (bs, Lq, M*D)."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes.tolist()]
    value_list = value.split([h * w for h, w in shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (H_, W_) in enumerate(shapes):
        # (bs, H*W, M, D) -> (bs*M, D, H, W)
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(
            bs * num_heads, embed_dims, H_, W_)
        # (bs, Lq, M, P, 2) -> (bs*M, Lq, P, 2)
        sampling_grid_l_ = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_l_ = F.grid_sample(value_l_, sampling_grid_l_, mode='bilinear',
                                          padding_mode='zeros', align_corners=False)
        sampling_value_list.append(sampling_value_l_)           # (bs*M, D, Lq, P)
    attention_weights = attention_weights.transpose(1, 2).reshape(
        bs * num_heads, 1, num_queries, num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1).view(
        bs, num_heads * embed_dims, num_queries)
    return output.transpose(1, 2).contiguous()


def msda_scalar_f64(value, spatial_shapes, level_start_index, sampling_locations,
                    attention_weights):
    """Scalar fp64 restatement of ms_deformable_im2col's arithmetic (numpy loops; small cases only).
    Also returns the number of bilinear corners that fall inside their map (N_in)."""
    v = np.asarray(value, dtype=np.float64)
    loc = np.asarray(sampling_locations, dtype=np.float64)
    aw = np.asarray(attention_weights, dtype=np.float64)
    B, S, M, D = v.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((B, Lq, M, D))
    n_in = 0
    for b in range(B):
        for q in range(Lq):
            for m in range(M):
                col = np.zeros(D)
                for l in range(L):
                    H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    st = int(level_start_index[l])
                    for p in range(P):
                        loc_w, loc_h = loc[b, q, m, l, p]
                        h_im = loc_h * H - 0.5
                        w_im = loc_w * W - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        h_low, w_low = int(np.floor(h_im)), int(np.floor(w_im))
                        h_high, w_high = h_low + 1, w_low + 1
                        lh, lw = h_im - h_low, w_im - w_low
                        hh, hw = 1 - lh, 1 - lw
                        val = np.zeros(D)
                        if h_low >= 0 and w_low >= 0:
                            val += hh * hw * v[b, st + h_low * W + w_low, m]; n_in += 1
                        if h_low >= 0 and w_high <= W - 1:
                            val += hh * lw * v[b, st + h_low * W + w_high, m]; n_in += 1
                        if h_high <= H - 1 and w_low >= 0:
                            val += lh * hw * v[b, st + h_high * W + w_low, m]; n_in += 1
                        if h_high <= H - 1 and w_high <= W - 1:
                            val += lh * lw * v[b, st + h_high * W + w_high, m]; n_in += 1
                        col += val * aw[b, q, m, l, p]
                out[b, q, m] = col
    return out.reshape(B, Lq, M * D), n_in


def count_inbounds_corners(spatial_shapes, sampling_locations):
    """Vectorised N_in: corners inside their map, fp32 arithmetic as the kernel."""
    loc = sampling_locations.float()
    n = 0
    for l, (H, W) in enumerate([(int(h), int(w)) for h, w in spatial_shapes.tolist()]):
        x = loc[:, :, :, l, :, 0] * W - 0.5
        y = loc[:, :, :, l, :, 1] * H - 0.5
        ok = (y > -1) & (x > -1) & (y < H) & (x < W)
        xl, yl = torch.floor(x), torch.floor(y)
        t, btm = yl >= 0, (yl + 1) <= H - 1
        lft, rgt = xl >= 0, (xl + 1) <= W - 1
        n += int((ok & t & lft).sum() + (ok & t & rgt).sum() + (ok & btm & lft).sum() +
                 (ok & btm & rgt).sum())
    return n


def msda_backward_autograd(value, value_spatial_shapes, sampling_locations, attention_weights,
                           grad_output):
    """Oracle for `ms_deform_attn_backward` (reference call sites:
    multi_scale_deformable_attn_function.py:74-84,150-160): torch.autograd through the restated
    forward above.  SURVEY.md Appendix B.9 records that mmcv's col2im formulas equal this derivative.
    -> (grad_value, grad_sampling_loc, grad_attn_weight) in the inputs' dtype (use float64)."""
    v = value.detach().clone().requires_grad_(True)
    loc = sampling_locations.detach().clone().requires_grad_(True)
    aw = attention_weights.detach().clone().requires_grad_(True)
    out = multi_scale_deformable_attn_pytorch(v, value_spatial_shapes, loc, aw)
    out.backward(grad_output.to(out.dtype))
    return v.grad, loc.grad, aw.grad
