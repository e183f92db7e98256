"""Autograd Functions at the operator boundary — mirror of the reference's
projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:15-163:
same class names, same `.apply(value, value_spatial_shapes, value_level_start_index,
sampling_locations, attention_weights, im2col_step)` signature, same cast behaviour
(`custom_fwd(cast_inputs=float32 / float16)`), backward returning
(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None), once_differentiable.
The extension module comes from `ext_loader.load_ext('_ext', [...])` exactly as in the reference,
but resolves to the MI355X library (occnet_amd.ext -> libocc_amd.so).
"""
import torch
from torch.autograd.function import Function, once_differentiable

from .. import ext_loader

ext_module = ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])

try:  # torch >= 2.4 spelling
    from torch.amp import custom_bwd as _custom_bwd, custom_fwd as _custom_fwd

    def custom_fwd(cast_inputs):
        return _custom_fwd(device_type='cuda', cast_inputs=cast_inputs)

    custom_bwd = _custom_bwd(device_type='cuda')
except ImportError:  # pragma: no cover
    from torch.cuda.amp import custom_bwd, custom_fwd


def _forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
             attention_weights, im2col_step, compute_dtype):
    ctx.im2col_step = im2col_step
    in_dtype = value.dtype
    # the gfx950 kernels compute in fp32; a half-precision caller gets its dtype back
    output = ext_module.ms_deform_attn_forward(
        value.float().contiguous(), value_spatial_shapes, value_level_start_index,
        sampling_locations.float().contiguous(), attention_weights.float().contiguous(),
        im2col_step=ctx.im2col_step)
    ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                          sampling_locations, attention_weights)
    return output.to(in_dtype)


def _backward(ctx, grad_output):
    value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = \
        ctx.saved_tensors
    v32 = value.float().contiguous()
    l32 = sampling_locations.float().contiguous()
    a32 = attention_weights.float().contiguous()
    grad_value = torch.zeros_like(v32)
    grad_sampling_loc = torch.zeros_like(l32)
    grad_attn_weight = torch.zeros_like(a32)
    ext_module.ms_deform_attn_backward(
        v32, value_spatial_shapes, value_level_start_index, l32, a32,
        grad_output.float().contiguous(), grad_value, grad_sampling_loc, grad_attn_weight,
        im2col_step=ctx.im2col_step)
    return (grad_value.to(value.dtype), None, None, grad_sampling_loc.to(sampling_locations.dtype),
            grad_attn_weight.to(attention_weights.dtype), None)


class MultiScaleDeformableAttnFunction_fp16(Function):

    @staticmethod
    @custom_fwd(cast_inputs=torch.float16)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        return _forward(ctx, value, value_spatial_shapes, value_level_start_index,
                        sampling_locations, attention_weights, im2col_step, torch.float16)

    @staticmethod
    @once_differentiable
    @custom_bwd
    def backward(ctx, grad_output):
        return _backward(ctx, grad_output)


class MultiScaleDeformableAttnFunction_fp32(Function):

    @staticmethod
    @custom_fwd(cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        return _forward(ctx, value, value_spatial_shapes, value_level_start_index,
                        sampling_locations, attention_weights, im2col_step, torch.float32)

    @staticmethod
    @once_differentiable
    @custom_bwd
    def backward(ctx, grad_output):
        return _backward(ctx, grad_output)
