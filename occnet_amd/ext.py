"""The operator module: mirror of the reference's `mmcv._ext` surface for this path.

`ms_deform_attn_forward` / `ms_deform_attn_backward` keep mmcv's exact Python signatures
(reference call sites: projects/mmdet3d_plugin/bevformer/modules/
multi_scale_deformable_attn_function.py:42-48, 74-84, 118-124, 150-160): tensors are torch-owned and
device resident, `im2col_step` is passed as a keyword, the op runs on the current stream without
synchronising, failures raise Python exceptions.  Underneath, each call hands raw device pointers to
the C ABI (include/occnet_amd.h) of libocc_amd.so.

The remaining functions expose the fused MI355X kernels (no counterpart in mmcv._ext).
"""
import ctypes
import os

import torch

from . import _lib, cache_epoch
from ._lib import OccAmdError, OccAmdUnsupported, f32, i32, i64, ptr, stream_ptr


_TIMING = None   # None, or {kernel name: [(start_event, end_event), ...]} (bench.py roofline leg)
_TIMING_ONLY = None   # None = every instrumented launch, or the set of kernel names still timed


def kernel_timing(enable=True):
    """Turn HIP-event timing of the instrumented launches on/off.  Events are recorded on the
    stream the kernel is launched on (torch's current stream).  Returns the record dict."""
    global _TIMING, _TIMING_ONLY
    _TIMING = {} if enable else None
    _TIMING_ONLY = None
    return _TIMING


def kernel_timing_only(names):
    """Restrict the timing to the given kernel names (None = all again): two event records per launch cost
    ~4 us of queue time each, 0.3 ms per step when every Linear is timed."""
    global _TIMING_ONLY
    _TIMING_ONLY = None if names is None else set(names)


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = _TIMING is not None and (_TIMING_ONLY is None or self.name in _TIMING_ONLY)
        if self.on:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.on:
            self.ev[1].record()
            _TIMING.setdefault(self.name, []).append(self.ev)
        return False


def _note_flops(name, flops):
    """bench.py's backbone roofline: the FLOPs of an instrumented launch, next to its event pair."""
    if _TIMING is not None and (_TIMING_ONLY is None or name in _TIMING_ONLY):
        _TIMING.setdefault(name + '_flops', []).append(float(flops))


def kernel_times_ms(record):
    """{name: [ms per launch]} from a kernel_timing() record (call after a device synchronize); entries
    that are plain numbers (e.g. 'linear_flops') pass through."""
    return {k: [a.elapsed_time(b) for a, b in v] if v and isinstance(v[0], tuple) else list(v)
            for k, v in record.items()}


def _need_cuda_f32(name, t, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise OccAmdError(f"{name} must be a device (HIP) tensor: the MI355X path has no CPU fallback")
    if t.dtype != torch.float32:
        raise OccAmdError(f"{name} must be float32, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise OccAmdError(f"{name} must be contiguous")


def _need_cuda_i64(name, t):
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
        raise OccAmdError(f"{name} must be a contiguous int64 device tensor")


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                           sampling_locations, attention_weights, im2col_step=64):
    """-> Tensor (B, Lq, M*D).  Same contract as mmcv._ext.ms_deform_attn_forward."""
    for n, t in (("value", value), ("sampling_locations", sampling_locations),
                 ("attention_weights", attention_weights)):
        _need_cuda_f32(n, t)
    _need_cuda_i64("value_spatial_shapes", value_spatial_shapes)
    _need_cuda_i64("value_level_start_index", value_level_start_index)
    if value.dim() != 4 or sampling_locations.dim() != 6 or attention_weights.dim() != 5:
        raise OccAmdError("ms_deform_attn_forward: expected value (B,S,M,D), sampling_locations "
                          "(B,Lq,M,L,P,2), attention_weights (B,Lq,M,L,P)")
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_locations.shape
    if (M2 != M or two != 2 or sampling_locations.shape[0] != B or
            tuple(attention_weights.shape) != (B, Lq, M, L, P) or
            tuple(value_spatial_shapes.shape) != (L, 2) or value_level_start_index.numel() != L):
        raise OccAmdError("ms_deform_attn_forward: inconsistent shapes")
    out = torch.empty((B, Lq, M * D), dtype=torch.float32, device=value.device)
    with torch.cuda.device(value.device):
        rc = _lib.lib().occ_ms_deform_attn_forward_f32(
            ptr(value), ptr(value_spatial_shapes), ptr(value_level_start_index),
            ptr(sampling_locations), ptr(attention_weights), ptr(out), i32(B), i32(S), i32(M),
            i32(D), i32(L), i32(Lq), i32(P), i32(int(im2col_step)), stream_ptr(value.device))
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
                            sampling_locations, attention_weights, grad_output, grad_value,
                            grad_sampling_loc, grad_attn_weight, im2col_step=64):
    """Writes into the three pre-zeroed grad tensors; returns None (mmcv contract)."""
    for n, t in (("value", value), ("sampling_locations", sampling_locations),
                 ("attention_weights", attention_weights), ("grad_output", grad_output),
                 ("grad_value", grad_value), ("grad_sampling_loc", grad_sampling_loc),
                 ("grad_attn_weight", grad_attn_weight)):
        _need_cuda_f32(n, t)
    _need_cuda_i64("value_spatial_shapes", value_spatial_shapes)
    _need_cuda_i64("value_level_start_index", value_level_start_index)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    if (grad_value.shape != value.shape or grad_sampling_loc.shape != sampling_locations.shape or
            grad_attn_weight.shape != attention_weights.shape or
            grad_output.numel() != B * Lq * M * D):
        raise OccAmdError("ms_deform_attn_backward: inconsistent shapes")
    lib = _lib.lib()
    lib.occ_ms_deform_attn_backward_workspace_bytes.restype = ctypes.c_int64
    # scratch of the atomic-free grad_value path from torch's caching allocator (not hipMallocAsync behind its back);
    # freed back to the pool when this call returns — stream-ordered, the kernels are already enqueued
    need = int(lib.occ_ms_deform_attn_backward_workspace_bytes(i32(B), i32(S), i32(M), i32(D), i32(L), i32(Lq),
                                                               i32(P)))
    ws = torch.empty(need, dtype=torch.uint8, device=value.device) if need > 0 else None
    with torch.cuda.device(value.device):
        rc = lib.occ_ms_deform_attn_backward_ws_f32(
            ptr(value), ptr(value_spatial_shapes), ptr(value_level_start_index),
            ptr(sampling_locations), ptr(attention_weights), ptr(grad_output), ptr(grad_value),
            ptr(grad_sampling_loc), ptr(grad_attn_weight), i32(B), i32(S), i32(M), i32(D), i32(L),
            i32(Lq), i32(P), i32(int(im2col_step)), ptr(ws), i64(need), stream_ptr(value.device))
    _lib.check(rc, "ms_deform_attn_backward")


def point_sampling(ref_3d, lidar2img, ego2lidar, pc_range, img_h, img_w):
    """ref_3d (B,Z,Nq,3), lidar2img (B,NC,4,4), ego2lidar (4,4) ->
    ref_cam (NC,B,Nq,Z,2) f32, bev_mask (NC,B,Nq,Z) bool, vis_bits (B,Nq) int32 bit field."""
    for n, t in (("ref_3d", ref_3d), ("lidar2img", lidar2img), ("ego2lidar", ego2lidar)):
        _need_cuda_f32(n, t)
    B, Z, Nq, three = ref_3d.shape
    NC = lidar2img.shape[1]
    if three != 3 or tuple(lidar2img.shape) != (B, NC, 4, 4) or tuple(ego2lidar.shape) != (4, 4):
        raise OccAmdError("point_sampling: inconsistent shapes")
    dev = ref_3d.device
    ref_cam = torch.empty((NC, B, Nq, Z, 2), dtype=torch.float32, device=dev)
    mask = torch.empty((NC, B, Nq, Z), dtype=torch.uint8, device=dev)
    vis = torch.empty((B, Nq), dtype=torch.int32, device=dev)
    pcr = (f32 * 6)(*[float(v) for v in pc_range])
    with torch.cuda.device(dev):
        rc = _lib.lib().occ_point_sampling_f32(
            ptr(ref_3d), ptr(lidar2img), ptr(ego2lidar), pcr, f32(float(img_h)), f32(float(img_w)),
            ptr(ref_cam), ptr(mask), ptr(vis), i32(B), i32(NC), i32(Nq), i32(Z), stream_ptr(dev))
    _lib.check(rc, "point_sampling")
    return ref_cam, mask.view(torch.bool), vis


# Storage type of the projected value maps the fused SCA gather reads (the reference keeps them in fp32:
# spatial_cross_attention.py:75,387-390).  All three keep sampling arithmetic, attention weights and accumulation in fp32.
#   'q16' (default since round 6): block floating point, 16 bits per element — one head row of a pixel = 64 bytes = 4 lanes x
#         16 bytes, so a wave load fetches 16 rows instead of 8 (csrc/sca_fused.hip); every 16-byte piece of 8 channels holds
#         int16 mantissas under one shared 4-bit exponent (csrc/common.h fma8q): the elements that dominate the gather's sums
#         are rounded to 2^-16 relative.  Against the CPU oracle: 7e-5 at the synthetic feature scale, < 6e-4 at ANY scale
#         (tests/test_gpu_value_range.py);
#   'f16' (OCC_SCA_VALUES=f16, the default of rounds 3-5): fp16 rows of the same geometry, 2^-12 relative on every element:
#         2.2e-4 at the synthetic scale, up to 1.4e-3 when the camera term dominates the residual; 2 % faster per step
#         (the q16 decode costs 17 VALU instructions per 16-byte load against 9, its encode 10 us per projection launch);
#   'f32' (OCC_SCA_VALUES=f32): the reference's storage, 8 lanes per 128-byte row (round-1/2 kernel): 3e-5, +15 % per step.
SCA_VALUES = os.environ.get("OCC_SCA_VALUES", "q16")
if SCA_VALUES not in ("f16", "f32", "q16"):
    raise OccAmdError(f"OCC_SCA_VALUES={SCA_VALUES!r}: expected f16, f32 or q16")


def sca_head_major():
    """The 16-bit-row SCA gather runs on the head-major kernel (one wave = 8 queries x one head, heads dealt to the XCDs;
    csrc/sca_fused.hip — the default since round 6); OCC_SCA_HEAD_MAJOR=0 selects the query-major kernel.  Read per call, like
    the library does.  (fp32 rows always take the query-major fp32 kernel.)"""
    return os.environ.get("OCC_SCA_HEAD_MAJOR", "1") != "0" and sca_rows_16bit()


def sca_rows_16bit():
    """True when the fused gather's value maps are 16-bit rows in the pixel-pair layout with a range scale (f16, q16)."""
    return SCA_VALUES in ("f16", "q16")


def sca_rows_dtype():
    return {"f16": torch.float16, "q16": torch.int16, "f32": torch.float32}[SCA_VALUES]


def sca_variant_name():
    hm = sca_head_major()
    return {"f16": "sca_fused_hm_kernel<4,8> (head-major, fp16 value rows)" if hm else "sca_fused_h_kernel<4,8> (query-major, fp16 value rows)",
            "q16": ("sca_fused_hm_kernel<4,8,Q> (head-major, q16 block-floating-point value rows)" if hm
                    else "sca_fused_h_kernel<4,8,Q> (query-major, q16 block-floating-point value rows)"),
            "f32": "sca_fused_kernel<4,8> (query-major, fp32 value rows)"}[SCA_VALUES]


def sca_value_bytes():
    return 2 if sca_rows_16bit() else 4


def sca_pair_layout(value):
    """(B*NC, S, M, D) fp16 value maps in row order -> the fused gather's pixel-PAIR order: a contiguous
    (B*NC, S_pad, M, D) tensor, S_pad = S rounded up to even, whose memory is [b][pix >> 1][head][pix & 1][D] — the two
    x-neighbours (2k, 2k+1) of one head share one 128-byte line (csrc/sca_fused.hip).  The value projection's fp16
    outputs are written in this order directly; this helper is for tests and for maps that were projected in fp32."""
    BN, S, M, D = value.shape
    if S & 1:
        value = torch.cat([value, value.new_zeros(BN, 1, M, D)], 1)
    return value.reshape(BN, -1, 2, M, D).permute(0, 1, 3, 2, 4).contiguous().view(BN, -1, M, D)


def sca_unpair_layout(value_pairs, S=None):
    """Inverse of sca_pair_layout: pixel-pair-ordered (B*NC, S_pad, M, D) [or (B*NC * S_pad, M*D) with M = cols / 32]
    -> row order, first S rows."""
    BN, S_pad, M, D = value_pairs.shape
    v = value_pairs.reshape(BN, S_pad // 2, M, 2, D).permute(0, 1, 3, 2, 4).reshape(BN, S_pad, M, D)
    return v if S is None else v[:, :S]


def sca_fused_forward(value, spatial_shapes, level_start_index, offs, logits, ref_cam, vis_bits,
                      num_heads, num_levels, num_points, order=None, stats=None, value_layout="rows", value_scale=None):
    """Fused SCA gather.  value (B*NC, S, M, D) float32 or float16; offs (B, Nq, M*L*P*2) / logits (B, Nq, M*L*P) may
    be column slices of one wider Linear output (last dim contiguous); ref_cam (NC,B,Nq,Z,2);
    vis_bits (B,Nq) int32.  -> slots (B, Nq, M*D) float32.
    fp16 maps reach the kernel in pixel-pair order (sca_pair_layout): value_layout="pairs" says the tensor already is
    (what value_proj_bf16 / value_proj_bf16_planes write into an fp16 output); "rows" (default) converts a row-ordered
    fp16 tensor first (a copy: tests and the fp32-projection path only).
    value_scale (fp16 maps only): 1-element float32 device tensor s, a power of two — the maps hold s * value
    (value_range_scale / f16_range_scaled), the kernel divides its fp32 sums by count * s: the same result, whatever s."""
    q16 = value.dtype == torch.int16
    half = value.dtype == torch.float16 or q16
    if q16 and value_layout != "pairs":
        raise OccAmdError("sca_fused_forward: q16 value maps exist in the pixel-pair layout only (sca_rows_encode_q16)")
    if value_scale is not None:
        if not half:
            raise OccAmdError("sca_fused_forward: value_scale goes with 16-bit value maps")
        if not (value_scale.is_cuda and value_scale.dtype == torch.float32 and value_scale.numel() == 1
                and value_scale.device == value.device):
            raise OccAmdError("sca_fused_forward: value_scale must be a 1-element float32 tensor on the value's device")
    if half:
        if value_layout not in ("rows", "pairs"):
            raise OccAmdError(f"sca_fused_forward: unknown value_layout {value_layout!r}")
        if not value.is_cuda:
            raise OccAmdError("sca_fused_forward: fp16 value must be a contiguous device tensor")
        if value_layout == "rows":
            value = sca_pair_layout(value)
        if not value.is_contiguous() or value.shape[1] % 2:
            raise OccAmdError("sca_fused_forward: pixel-pair fp16 value must be contiguous with an even number of rows")
    else:
        _need_cuda_f32("value", value)
    _need_cuda_f32("ref_cam", ref_cam)
    _need_cuda_f32("offs", offs, contiguous=False)
    _need_cuda_f32("logits", logits, contiguous=False)
    _need_cuda_i64("spatial_shapes", spatial_shapes)
    _need_cuda_i64("level_start_index", level_start_index)
    NC, B, Nq, Z, _ = ref_cam.shape
    BN, S, M, D = value.shape
    L, P = int(num_levels), int(num_points)
    if BN != B * NC or M != num_heads:
        raise OccAmdError("sca_fused_forward: value batch must equal B*num_cams")
    for n, t, w in (("offs", offs, M * L * P * 2), ("logits", logits, M * L * P)):
        if tuple(t.shape[:2]) != (B, Nq) or t.shape[-1] != w or t.stride(-1) != 1 \
                or t.stride(0) != Nq * t.stride(1):
            raise OccAmdError(f"sca_fused_forward: {n} must be (B,Nq,{w}) with unit inner stride")
    if vis_bits.dtype != torch.int32 or tuple(vis_bits.shape) != (B, Nq) or not vis_bits.is_contiguous():
        raise OccAmdError("sca_fused_forward: vis_bits must be contiguous int32 (B,Nq)")
    if order is not None and (order.dtype != torch.int32 or order.numel() != Nq):
        raise OccAmdError("sca_fused_forward: order must be int32 (Nq)")
    slots = torch.empty((B, Nq, M * D), dtype=torch.float32, device=value.device)
    fn = (_lib.lib().occ_sca_fused_forward_q16v if q16 else _lib.lib().occ_sca_fused_forward_f16v if half
          else _lib.lib().occ_sca_fused_forward_f32)
    tail = (ptr(value_scale), stream_ptr(value.device)) if half else (stream_ptr(value.device),)
    with torch.cuda.device(value.device), _timed('sca_fused_forward'):
        rc = fn(ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(offs), i64(offs.stride(1)), ptr(logits),
                i64(logits.stride(1)), ptr(ref_cam), ptr(vis_bits), ptr(order), ptr(slots), ptr(stats), i32(B),
                i32(NC), i32(S), i32(M), i32(D), i32(L), i32(P), i32(Z), i32(Nq), *tail)
    _lib.check(rc, "sca_fused_forward")
    return slots


def tsa_fused_forward(value, offs, logits, ref_2d, bev_h, bev_w, num_heads, num_points,
                      shared_queue=False, order=None, value_rows=None):
    """Fused TSA gather.  value: (B*2, Nq, M, D), or (B, Nq, M, D) with shared_queue=True when both
    queue entries are the same projected BEV (no history).  offs (B,Nq,M*2*P*2), logits
    (B,Nq,M*2*P), ref_2d (B*2,Nq,1,2).  -> (B, Nq, M*D).
    value_rows (B == 1 only): the queries are a ROW BAND of the BEV — offs / logits / ref_2d / order / the result cover
    Nq < bev_h*bev_w queries (order holds band-local indices) while `value` is the whole (bev_h*bev_w = value_rows)-pixel
    map: the encoder's row pipeline (plugin/encoder.py)."""
    _need_cuda_f32("value", value)
    _need_cuda_f32("ref_2d", ref_2d)
    _need_cuda_f32("offs", offs, contiguous=False)
    _need_cuda_f32("logits", logits, contiguous=False)
    B, Nq = offs.shape[:2]
    M, D, P = int(num_heads), value.shape[-1], int(num_points)
    Nv = Nq if value_rows is None else int(value_rows)
    if Nv != bev_h * bev_w or (value_rows is not None and B != 1):
        raise OccAmdError("tsa_fused_forward: the value map must have bev_h*bev_w rows (a query band needs B == 1)")
    per = Nv * M * D
    if shared_queue:
        if value.numel() != B * per:
            raise OccAmdError("tsa_fused_forward: shared value must be (B,Nq,M,D)")
        if B != 1:
            # entry (b,t) lives at (b*2+t)*stride: aliasing both entries needs stride 0 -> B == 1
            value = torch.stack([value.view(B, Nq, M, D)] * 2, 1).reshape(B * 2, Nq, M, D).contiguous()
            stride = per
        else:
            stride = 0
    else:
        if value.numel() != 2 * B * per:
            raise OccAmdError("tsa_fused_forward: value must be (B*2,Nq,M,D)")
        stride = per
    if tuple(ref_2d.shape) != (B * 2, Nq, 1, 2):
        raise OccAmdError("tsa_fused_forward: ref_2d must be (B*2,Nq,1,2)")
    for n, t, w in (("offs", offs, M * 2 * P * 2), ("logits", logits, M * 2 * P)):
        if t.shape[-1] != w or t.stride(-1) != 1 or t.stride(0) != Nq * t.stride(1):
            raise OccAmdError(f"tsa_fused_forward: {n} must be (B,Nq,{w}) with unit inner stride")
    out = torch.empty((B, Nq, M * D), dtype=torch.float32, device=value.device)
    with torch.cuda.device(value.device), _timed('tsa_fused_forward'):
        rc = _lib.lib().occ_tsa_fused_forward_f32(
            ptr(value), i64(stride), ptr(offs), i64(offs.stride(1)), ptr(logits),
            i64(logits.stride(1)), ptr(ref_2d), ptr(order), ptr(out), i32(B), i32(Nq), i32(bev_h),
            i32(bev_w), i32(M), i32(D), i32(P), stream_ptr(value.device))
    _lib.check(rc, "tsa_fused_forward")
    return out


def conv3d_channel_block(cin):
    """Input channels contracted per LDS phase by the MFMA Conv3d kernel (0 = unsupported)."""
    return int(_lib.lib().occ_conv3d_channel_block(i32(int(cin))))


CONV3D_PRECISION = os.environ.get("OCC_CONV3D_PRECISION", "bf16x3")   # 'bf16x3' (default) or 'f32'


def conv3d_pack_weight(weight, precision=None):
    """torch Conv3d weight (Cout, Cin, 3, 3, 3) f32 -> packed MFMA B-fragment order (flat tensor): float32 for
    the exact-f32 kernel, int16 (bf16 hi/lo planes) for the bf16x3 kernel (Cin % 16 == 0, else f32)."""
    _need_cuda_f32("weight", weight)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise OccAmdError("conv3d_pack_weight: expected a (Cout, Cin, 3, 3, 3) weight")
    cout, cin = weight.shape[:2]
    precision = precision or CONV3D_PRECISION
    if precision == 'bf16x3' and (cin % 16 == 0 or (cin == 8 and cout == 32)):
        # Cin = 8: two taps per MFMA step, 14 steps (the 28th tap is zero weights) -> 14 * 2 * 2 * 32 * 8 bf16
        n16 = 14 * 2 * 2 * 32 * 8 if cin == 8 else weight.numel() * 2
        packed = torch.empty(n16, dtype=torch.int16, device=weight.device)
        with torch.cuda.device(weight.device):
            rc = _lib.lib().occ_conv3d_pack_weight_bf16x3(ptr(weight), ptr(packed), i32(cin), i32(cout),
                                                          stream_ptr(weight.device))
        _lib.check(rc, "conv3d_pack_weight")
        return packed
    packed = torch.empty(weight.numel(), dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.lib().occ_conv3d_pack_weight_f32(ptr(weight), ptr(packed), i32(cin), i32(cout),
                                                   stream_ptr(weight.device))
    _lib.check(rc, "conv3d_pack_weight")
    return packed


def conv3d_bn_relu(x, w_packed, scale, shift, Z, Y, X, cin, cout, in_layout, out_xy_major=False,
                   relu=True):
    """Lifter/Conv3d(k3,p1)+BN(eval)+ReLU on the matrix cores; the kernel (exact f32, or bf16x3: operands split
    into hi + lo bf16, f32 accumulation) follows the dtype of `w_packed` (conv3d_pack_weight).

    x: in_layout 0 -> (B, Y, X, Z, cin); in_layout 1 -> (B, Y*X, cin*Z) BEV embedding (lifter view).
    -> (B, Y, X, Z, cout), or (B, X, Y, Z, cout) with out_xy_major (the reference's
    permute(0,4,3,2,1) order, transformer_occ.py:308)."""
    for n, t in (("x", x), ("scale", scale), ("shift", shift)):
        _need_cuda_f32(n, t)
    x3 = w_packed.dtype == torch.int16
    if not (w_packed.is_cuda and w_packed.is_contiguous() and (x3 or w_packed.dtype == torch.float32)):
        raise OccAmdError("conv3d_bn_relu: w_packed must come from conv3d_pack_weight")
    B = x.shape[0]
    n_w = (14 * 2 * 2 * 32 * 8 if cin == 8 else cout * cin * 27 * 2) if x3 else cout * cin * 27
    if x.numel() != B * Y * X * Z * cin or w_packed.numel() != n_w \
            or scale.numel() != cout or shift.numel() != cout:
        raise OccAmdError("conv3d_bn_relu: inconsistent shapes")
    if out_xy_major:
        out = torch.empty((B, X, Y, Z, cout), dtype=torch.float32, device=x.device)
        sy, sx = Z * cout, Y * Z * cout
    else:
        out = torch.empty((B, Y, X, Z, cout), dtype=torch.float32, device=x.device)
        sy, sx = X * Z * cout, Z * cout
    fn = _lib.lib().occ_conv3d_bn_relu_bf16x3_f32 if x3 else _lib.lib().occ_conv3d_bn_relu_f32
    with torch.cuda.device(x.device), _timed('conv3d_bn_relu'):
        rc = fn(ptr(x), ptr(w_packed), ptr(scale), ptr(shift), ptr(out), i32(B), i32(Z), i32(Y), i32(X),
                i32(cin), i32(cout), i32(int(in_layout)), i64(Y * X * Z * cout), i64(sy), i64(sx),
                i32(1 if relu else 0), stream_ptr(x.device))
    _lib.check(rc, "conv3d_bn_relu")
    return out


def conv3d_heads_pack(w1_occ, b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow):
    """The eight head tensors (C = 32, hidden = 64) -> the operand buffer of conv3d_heads_decode (bf16 hi/lo MFMA
    fragments + biases; uint8 tensor)."""
    ts = (w1_occ, b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow)
    for n, t in zip(("w1_occ", "b1_occ", "w2_occ", "b2_occ", "w1_flow", "b1_flow", "w2_flow", "b2_flow"), ts):
        _need_cuda_f32(n, t)
    hidden, C = w1_occ.shape
    ncls = w2_occ.shape[0]
    if (tuple(w1_flow.shape) != (hidden, C) or tuple(w2_occ.shape) != (ncls, hidden) or tuple(w2_flow.shape) != (2, hidden)
            or b1_occ.numel() != hidden or b1_flow.numel() != hidden or b2_occ.numel() != ncls or b2_flow.numel() != 2):
        raise OccAmdError("conv3d_heads_pack: inconsistent weight shapes")
    lib = _lib.lib()
    lib.occ_conv3d_heads_pack_bytes.restype = ctypes.c_int64
    packed = torch.empty(int(lib.occ_conv3d_heads_pack_bytes()), dtype=torch.uint8, device=w1_occ.device)
    with torch.cuda.device(w1_occ.device):
        rc = lib.occ_conv3d_heads_pack(*[ptr(t) for t in ts], ptr(packed), i32(C), i32(hidden), i32(ncls),
                                       stream_ptr(w1_occ.device))
    _lib.check(rc, "conv3d_heads_pack")
    packed._occ_heads_meta = (int(C), int(hidden), int(ncls))     # the fragment layout bakes these in: checked at use
    return packed


def conv3d_heads_decode(x, w_packed, scale, shift, heads_packed, Z, Y, X, num_classes):
    """Second decoder convolution + BN(eval) + ReLU + both occupancy heads + argmax decode in one launch: x
    (B, Y, X, Z, 32) fp32 -> occ (B, X, Y, Z, num_classes), flow (B, X, Y, Z, 2), occ_cls (B, X, Y, Z) int64 — what
    conv3d_bn_relu(out_xy_major=True) + occ_heads(decode=True) produce, without the 82 MB round trip of the
    convolution's output.  w_packed: conv3d_pack_weight (bf16x3); heads_packed: conv3d_heads_pack."""
    for n, t in (("x", x), ("scale", scale), ("shift", shift)):
        _need_cuda_f32(n, t)
    if w_packed.dtype != torch.int16:
        raise OccAmdUnsupported("conv3d_heads_decode: needs the bf16x3 packed weight")
    meta = getattr(heads_packed, '_occ_heads_meta', None)
    if meta is not None and meta[2] != int(num_classes):
        raise OccAmdError(f"conv3d_heads_decode: heads were packed for {meta[2]} classes, called with {num_classes}")
    B = x.shape[0]
    cin = x.numel() // (B * Y * X * Z)
    if x.numel() != B * Y * X * Z * cin or w_packed.numel() != 32 * cin * 27 * 2:
        raise OccAmdError("conv3d_heads_decode: inconsistent shapes")
    occ = torch.empty((B, X, Y, Z, num_classes), dtype=torch.float32, device=x.device)
    flow = torch.empty((B, X, Y, Z, 2), dtype=torch.float32, device=x.device)
    cls = torch.empty((B, X, Y, Z), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device), _timed('conv3d_heads'):
        rc = _lib.lib().occ_conv3d_heads_decode_bf16x3_f32(
            ptr(x), ptr(w_packed), ptr(scale), ptr(shift), ptr(heads_packed), ptr(occ), ptr(flow), ptr(cls), i32(B),
            i32(Z), i32(Y), i32(X), i32(cin), i32(num_classes), stream_ptr(x.device))
    _lib.check(rc, "conv3d_heads_decode")
    return occ, flow, cls


HEADS_PRECISION = os.environ.get("OCC_HEADS_PRECISION", "bf16x3")    # 'bf16x3' (default) or 'f32'


def occ_heads(feat, w1_occ, b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow, decode=False,
              precision=None):
    """feat (..., C) -> occ (..., num_classes), flow (..., 2): both decoder MLP heads in one kernel.  With `decode`
    the same pass also writes argmax_c occ (int64, first index on ties): -> (occ, flow, occ_cls)."""
    ts = (feat, w1_occ, b1_occ, w2_occ, b2_occ, w1_flow, b1_flow, w2_flow, b2_flow)
    for n, t in zip(("feat", "w1_occ", "b1_occ", "w2_occ", "b2_occ", "w1_flow", "b1_flow", "w2_flow",
                     "b2_flow"), ts):
        _need_cuda_f32(n, t)
    C = feat.shape[-1]
    hidden, ncls = w1_occ.shape[0], w2_occ.shape[0]
    if (tuple(w1_occ.shape) != (hidden, C) or tuple(w1_flow.shape) != (hidden, C) or
            tuple(w2_occ.shape) != (ncls, hidden) or tuple(w2_flow.shape) != (2, hidden) or
            b1_occ.numel() != hidden or b1_flow.numel() != hidden or b2_occ.numel() != ncls or
            b2_flow.numel() != 2):
        raise OccAmdError("occ_heads: inconsistent weight shapes")
    n_rows = feat.numel() // C
    occ = torch.empty(feat.shape[:-1] + (ncls,), dtype=torch.float32, device=feat.device)
    flow = torch.empty(feat.shape[:-1] + (2,), dtype=torch.float32, device=feat.device)
    cls = torch.empty(feat.shape[:-1], dtype=torch.int64, device=feat.device) if decode else None
    with torch.cuda.device(feat.device), _timed('occ_heads'):
        rc = _lib.lib().occ_occ_heads_decode_f32(
            ptr(feat), ptr(w1_occ), ptr(b1_occ), ptr(w2_occ), ptr(b2_occ), ptr(w1_flow), ptr(b1_flow),
            ptr(w2_flow), ptr(b2_flow), ptr(occ), ptr(flow), ptr(cls), i64(n_rows), i32(C), i32(hidden),
            i32(ncls), i32(1 if (precision or HEADS_PRECISION) == "f32" else 0), stream_ptr(feat.device))
    _lib.check(rc, "occ_heads")
    return (occ, flow, cls) if decode else (occ, flow)


def _rows2d(name, t, k=None):
    """(…, K) tensor -> (data tensor, M, K, row stride) for the Linear kernel: rows must be uniformly
    strided, unit inner stride, 16-byte aligned."""
    _need_cuda_f32(name, t, contiguous=False)
    K = t.shape[-1]
    if k is not None and K != k:
        raise OccAmdError(f"linear: {name} has {K} columns, expected {k}")
    if t.dim() == 2 and t.stride(1) == 1:
        ld = t.stride(0)
    elif t.is_contiguous():
        ld = K
    else:
        raise OccAmdError(f"linear: {name} must be contiguous or a 2-D row-strided view")
    M = t.numel() // K
    if t.data_ptr() % 16 or ld % 4:
        raise OccAmdUnsupported(f"linear: {name} rows are not 16-byte aligned")
    return t, M, K, ld


# Precision of the MFMA Linear kernel: 'f32' = v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf
# chain), 'bf16x3' = three bf16 MFMAs on hi/lo-split operands with f32 accumulation (relative error of a
# product <= 2^-16; 2-3x faster).  The default can be overridden with OCC_LINEAR_PRECISION.
LINEAR_PRECISION = os.environ.get("OCC_LINEAR_PRECISION", "bf16x3")
_PACKED_W = {}          # (data_ptr, version, shape) -> packed hi/lo bf16 weight (small LRU)


def linear_pack_weight_bf16x3(weight):
    """(N, K) float32 Linear weight -> bf16 hi/lo split in MFMA fragment order
    packed[K/16][ceil(N/32)][hi, lo][lane][8] (int16 storage, columns zero-padded to a multiple of 32), cached."""
    # temporaries (a transposed weight in a backward pass, a concatenation rebuilt every forward) are marked by their
    # producer and packed without entering the cache: an entry keeps its tensor alive and can never hit again
    no_cache = getattr(weight, '_occ_no_cache', False)
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), str(weight.device), cache_epoch())
    hit = None if no_cache else _PACKED_W.get(key)
    if hit is not None:
        return hit[1]
    _need_cuda_f32("weight", weight)
    N, K = weight.shape
    packed = torch.empty((N + 31) // 32 * 32 * K * 2, dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.lib().occ_linear_pack_weight_bf16x3(ptr(weight), ptr(packed), i32(N), i32(K),
                                                      stream_ptr(weight.device))
    _lib.check(rc, "linear_pack_weight_bf16x3")
    if no_cache:
        return packed
    if len(_PACKED_W) >= 256:
        _PACKED_W.pop(next(iter(_PACKED_W)))
    # the entry keeps the weight tensor alive: while it is cached its address cannot be recycled by
    # another tensor, so (data_ptr, version) identifies the contents
    _PACKED_W[key] = (weight, packed)
    return packed


def _check_out_scale(what, out_scale, planes, out):
    if out_scale is None:
        return
    if not (out_scale.is_cuda and out_scale.dtype == torch.float32 and out_scale.dim() == 1 and out_scale.is_contiguous()
            and out_scale.numel() >= planes and out_scale.device == out.device):
        raise OccAmdError(f"{what}: out_scale must be a contiguous float32 device vector of >= {planes} entries")


def _range_terms_dev(row_l1, bias_max, dev):
    """(P,) float32 device vectors of the two weight-side terms: device tensors pass through, host floats are uploaded."""
    def one(v):
        if isinstance(v, torch.Tensor):
            t = v.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        else:
            t = torch.tensor([float(x) for x in v], dtype=torch.float32, device=dev)
        return t
    r, b = one(row_l1), one(bias_max)
    if r.numel() != b.numel() or r.numel() == 0:
        raise OccAmdError("value_range_scale: row_l1 and bias_max must have one entry per plane")
    return r, b


def value_range_scale(a_list, row_l1, bias_max):
    """Per-plane power-of-two scales for fp16 storage of the projections of the bf16 feature rows `a_list` (csrc/value_range.hip):
    -> float32 device vector t of 2 P + 1 entries, t[:P] = s_p with  (max|x| * row_l1[p] * (1 + 2^-8) + bias_max[p]) * s_p <= 2^15,
    t[P] = max|x| over every map, t[P + 1 + p] = the bound of plane p (diagnostics).  row_l1[p] = max_n sum_k |W_p[n][k]|,
    bias_max[p] = max |group bias of p| — (P,) float32 DEVICE tensors (round 6; host sequences are uploaded: tests).  No host
    synchronisation; a memset node + one launch on the current stream.  Pass t[:P] as value_proj_bf16*(out_scale=) and
    t[p:p+1] as sca_fused_forward(value_scale=)."""
    if isinstance(a_list, torch.Tensor):
        a_list = [a_list]
    S = len(a_list)
    K = a_list[0].shape[1]
    for a in a_list:
        if not (a.is_cuda and a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == K):
            raise OccAmdUnsupported("value_range_scale: every a must be a (M, K) bfloat16 device matrix with unit column stride")
    dev = a_list[0].device
    r, b = _range_terms_dev(row_l1, bias_max, dev)
    P = r.numel()
    # 2 P + 1 result floats + the kernel's two work words in ONE allocation of the caller's pool (the launcher zeroes the
    # words: nothing is cached across calls, streams or graph captures)
    buf = torch.empty(2 * P + 3, dtype=torch.float32, device=dev)
    out, work = buf[:2 * P + 1], buf[2 * P + 1:]
    arr64 = lambda v: (ctypes.c_int64 * S)(*[int(x) for x in v])
    a_ptrs = (ctypes.c_void_p * S)(*[a.data_ptr() for a in a_list])
    with torch.cuda.device(dev), _timed('value_range'):
        rc = _lib.lib().occ_value_range_scale_bf16(
            i32(S), a_ptrs, arr64([a.stride(0) for a in a_list]), arr64([a.shape[0] for a in a_list]), i32(K), i32(P),
            ptr(r), ptr(b), ptr(out), ptr(work), stream_ptr(dev))
    _lib.check(rc, "value_range_scale")
    return out


def new_absmax_words(device):
    """8 zeroed device words for conv3x3_nhwc(amax=...): the producer side of value_range_scale_from_amax."""
    return torch.zeros(8, dtype=torch.int32, device=device)


def attach_absmax(t, words):
    """Hang the producer's max|x| words on a feature-map tensor (or a view of one) together with the tensor's version counter:
    an in-place write to the map afterwards invalidates the side band (absmax_of returns None, the consumer measures)."""
    t._occ_absmax = words
    t._occ_absmax_version = t._version
    return t


def absmax_of(t):
    """The words attach_absmax hung on `t`, or None (none attached, or the map was written in place since)."""
    w = getattr(t, '_occ_absmax', None)
    if w is None or getattr(t, '_occ_absmax_version', None) != t._version:
        return None
    return w


def feature_absmax_words(a_list):
    """The 8 words a producer would have accumulated, for maps that did not come from this library's backbone plan (bench
    set-up, tests): max of the sign-stripped bf16 patterns of every element, replicated."""
    if isinstance(a_list, torch.Tensor):
        a_list = [a_list]
    m = torch.stack([(a.detach().contiguous().view(torch.int16).to(torch.int32) & 0x7fff).amax() for a in a_list]).amax()
    return m.to(torch.int32).reshape(1).repeat(8)


def value_range_scale_from_amax(amax8, row_l1, bias_max):
    """value_range_scale when the producer of the maps accumulated max|x| itself (conv3x3_nhwc(amax=...) of the backbone plan's
    FPN output convolutions): one 64-thread launch instead of a pass over the maps.  Same result vector."""
    if not (isinstance(amax8, torch.Tensor) and amax8.is_cuda and amax8.dtype == torch.int32 and amax8.numel() == 8
            and amax8.is_contiguous()):
        raise OccAmdError("value_range_scale_from_amax: amax8 must be 8 contiguous int32 device words")
    dev = amax8.device
    r, b = _range_terms_dev(row_l1, bias_max, dev)
    P = r.numel()
    out = torch.empty(2 * P + 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed('value_range'):
        rc = _lib.lib().occ_value_range_scale_from_amax(ptr(amax8), i32(P), ptr(r), ptr(b), ptr(out), stream_ptr(dev))
    _lib.check(rc, "value_range_scale_from_amax")
    return out


def f16_range_scaled(v):
    """fp32 value rows -> (fp16 rows holding s * v, s as a 1-element float32 device tensor), s the power of two that puts
    max|v| into [2^14, 2^15]: the ATen counterpart of value_range_scale for value tensors that were projected in fp32 (the
    non-LazyFeatures inputs of the fused SCA gather).  No host synchronisation; Inf / NaN / all-zero rows: s = 1."""
    amax = v.detach().abs().amax().float()
    _, e = torch.frexp(amax)
    ok = torch.isfinite(amax) & (amax > 0)
    s = torch.where(ok, torch.ldexp(torch.ones_like(amax), (15 - e).clamp(-100, 100)), torch.ones_like(amax))
    return (v * s).clamp(-65504.0, 65504.0).half(), s.reshape(1)


def sca_rows_encode_q16(v, scale=None):
    """fp32 value rows (BN, S, C) [or (BN, S, M, D)], C = heads * 32 -> q16 pixel pairs: int16 (BN, S + (S & 1), C) in the fused
    gather's pair order, each 16-byte piece of 8 channels holding int16 mantissas of scale * v under one 4-bit exponent
    (csrc/common.h; numpy restatement: tests/q16_ref.py).  scale: 1-element float32 device tensor, a power of two with
    max|v| * scale <= 2^15 (q16_range_scaled derives it), None = 1."""
    _need_cuda_f32("v", v)
    BN, S = v.shape[0], v.shape[1]
    C = v.numel() // (BN * S)
    out = torch.empty((BN, S + (S & 1), C), dtype=torch.int16, device=v.device)
    if S & 1:                                    # the pad pixel (second slot of the last pair, every head): never sampled, kept defined
        out.view(BN, (S + 1) // 2, C // 32, 2, 32)[:, -1, :, 1, :].zero_()
    with torch.cuda.device(v.device):
        rc = _lib.lib().occ_sca_rows_encode_q16(ptr(v), ptr(out), ptr(scale), i64(BN), i32(S), i32(C), stream_ptr(v.device))
    _lib.check(rc, "sca_rows_encode_q16")
    return out


def q16_range_scaled(v):
    """fp32 value rows -> (q16 pixel pairs, s): the q16 counterpart of f16_range_scaled (same power-of-two rule for s)."""
    amax = v.detach().abs().amax().float()
    _, e = torch.frexp(amax)
    ok = torch.isfinite(amax) & (amax > 0)
    s = torch.where(ok, torch.ldexp(torch.ones_like(amax), (15 - e).clamp(-100, 100)), torch.ones_like(amax)).reshape(1)
    return sca_rows_encode_q16(v.contiguous(), s), s


def value_proj_bf16(a_list, weight, group_bias, out, rows_per_group, out_group_rows, out_row0, out_scale=None):
    """For every segment s (FPN level) and row m = g*rows_per_group[s] + i of a_list[s]:
        out[g*out_group_rows + out_row0[s] + i] = a_list[s][m] @ weight.T + group_bias[s][g % G]   (fp32 out),
    all segments in one launch.  An fp16 `out` is the fused SCA gather's operand and is written in its pixel-PAIR
    order (sca_pair_layout: row r of a group's block lands at [r >> 1][head][r & 1][32]; out_group_rows must be even,
    N a multiple of 32, rows contiguous) — read it back in row order with sca_unpair_layout.  a_list[s] (M_s, K) bf16 with unit column stride (an NHWC feature map seen as
    pixels x channels); weight (N, K) fp32 Linear weight (packed hi/lo once, cached); group_bias (S, G, N) fp32
    contiguous or None; out fp32 2-D (rows, N); rows_per_group / out_row0: one int per segment.
    out_scale (fp16 out only): float32 device vector, out = fp16(out_scale[0] * (...)) — value_range_scale()."""
    if isinstance(a_list, torch.Tensor):
        a_list, rows_per_group, out_row0 = [a_list], [rows_per_group], [out_row0]
        if group_bias is not None:
            group_bias = group_bias.unsqueeze(0)
    S = len(a_list)
    out_q16 = out.dtype == torch.int16
    out_half = out.dtype == torch.float16 or out_q16
    if not out_half:
        _need_cuda_f32("out", out)
    if not out.is_cuda or out.dim() != 2 or out.stride(1) != 1:
        raise OccAmdError("value_proj_bf16: out must be a 2-D fp32 (or fp16) device matrix with unit column stride")
    N, K = weight.shape
    if out.shape[1] != N or len(rows_per_group) != S or len(out_row0) != S:
        raise OccAmdError("value_proj_bf16: inconsistent shapes")
    for a, rpg, r0 in zip(a_list, rows_per_group, out_row0):
        if not (a.is_cuda and a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1
                and a.shape[1] == K):
            raise OccAmdUnsupported("value_proj_bf16: every a must be a (M, K) bfloat16 device matrix with unit "
                                    "column stride")
        groups = (a.shape[0] + rpg - 1) // rpg
        if (groups - 1) * out_group_rows + r0 + min(rpg, a.shape[0]) > out.shape[0]:
            raise OccAmdError("value_proj_bf16: output rows out of range")
    G = 0
    gb_ptrs = None
    if group_bias is not None:
        _need_cuda_f32("group_bias", group_bias)
        if group_bias.dim() != 3 or group_bias.shape[0] != S or group_bias.shape[2] != N \
                or not group_bias.is_contiguous():
            raise OccAmdError("value_proj_bf16: group_bias must be a contiguous (S, G, N) tensor")
        G = group_bias.shape[1]
        gb_ptrs = (ctypes.c_void_p * S)(*[group_bias[s].data_ptr() for s in range(S)])
    if out_scale is not None and not out_half:
        raise OccAmdError("value_proj_bf16: out_scale goes with an fp16 out")
    _check_out_scale("value_proj_bf16", out_scale, 1, out)
    packed = linear_pack_weight_bf16x3(weight)
    arr64 = lambda v: (ctypes.c_int64 * S)(*[int(x) for x in v])
    a_ptrs = (ctypes.c_void_p * S)(*[a.data_ptr() for a in a_list])
    fn = (_lib.lib().occ_value_proj_bf16_q16pairs if out_q16 else _lib.lib().occ_value_proj_bf16_f16pairs if out_half
          else _lib.lib().occ_value_proj_bf16_f32)
    tail = (ptr(out_scale), stream_ptr(out.device)) if out_half else (stream_ptr(out.device),)
    with torch.cuda.device(out.device), _timed('value_proj'):
        rc = fn(
            i32(S), a_ptrs, arr64([a.stride(0) for a in a_list]), arr64([a.shape[0] for a in a_list]),
            arr64(rows_per_group), arr64(out_row0), gb_ptrs, i32(G), ptr(packed), ptr(out), i64(out.stride(0)),
            i32(K), i32(N), i64(out_group_rows), *tail)
    _lib.check(rc, "value_proj_bf16")
    return out


_STACKED_VP = {}        # (weights' (ptr, version) ..., epoch) -> (stacked weight, its pack) of value_proj_bf16_planes
_STACKED_GB = {}        # (bias tables' (ptr, version) ..., epoch) -> (their concatenation, the sources kept alive)


def value_proj_planes_prepare(weights):
    """The stacked + packed weight of value_proj_bf16_planes for this list of projections (cached on the weights'
    identities): built on the CURRENT stream.  A caller that launches value_proj_bf16_planes on a side stream calls this
    on the main stream first, so that no main-stream reader of the cache entry can race its construction."""
    key = tuple((w.data_ptr(), w._version) for w in weights) + (str(weights[0].device), cache_epoch())
    hit = _STACKED_VP.get(key)
    if hit is None:
        stacked = torch.cat([w.detach() for w in weights], 0).contiguous()
        stacked._occ_no_cache = True
        if len(_STACKED_VP) >= 8:
            _STACKED_VP.pop(next(iter(_STACKED_VP)))
        hit = _STACKED_VP[key] = (stacked, linear_pack_weight_bf16x3(stacked), list(weights))
    return hit


def value_proj_bf16_planes(a_list, weights, group_biases, out, rows_per_group, out_group_rows, out_row0, out_scale=None):
    """The same rows through SEVERAL projections in one launch (the encoder layers' SCA value projections):
    out[p] = value_proj_bf16(a_list, weights[p], group_biases[p], ...) for every p, feature rows read from HBM once.
    weights: list of P (N, K) fp32 Linear weights (N % 256 == 0); group_biases: list of P (S, G, N) fp32 or None;
    out (P, rows, N) fp32 or fp16, contiguous.  out_scale: float32 device vector of P per-plane factors
    (value_range_scale()), plane p = out_scale[p] * (...)."""
    P, S = len(weights), len(a_list)
    N, K = weights[0].shape
    if any(tuple(w.shape) != (N, K) for w in weights) or N % 256:
        raise OccAmdUnsupported("value_proj_bf16_planes: equal (N, K) weights with N % 256 == 0 needed")
    out_q16 = out.dtype == torch.int16
    out_half = out.dtype == torch.float16 or out_q16
    if not (out.is_cuda and out.dim() == 3 and out.shape[0] == P and out.shape[2] == N and out.is_contiguous()
            and (out_half or out.dtype == torch.float32)):
        raise OccAmdError("value_proj_bf16_planes: out must be a contiguous (P, rows, N) fp32 / fp16 / int16 (q16) device tensor")
    for a, rpg, r0 in zip(a_list, rows_per_group, out_row0):
        if not (a.is_cuda and a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == K):
            raise OccAmdUnsupported("value_proj_bf16_planes: every a must be a (M, K) bfloat16 device matrix with unit "
                                    "column stride")
        groups = (a.shape[0] + rpg - 1) // rpg
        if (groups - 1) * out_group_rows + r0 + min(rpg, a.shape[0]) > out.shape[1]:
            raise OccAmdError("value_proj_bf16_planes: output rows out of range")
    _check_out_scale("value_proj_bf16_planes", out_scale, P, out)
    hit = value_proj_planes_prepare(weights)
    gb_ptrs, G, gb_keep = None, 0, None
    if group_biases is not None and group_biases[0] is not None:
        # (S, G, P*N), cached on the per-projection bias tables' identities: it was a cat kernel per step in front of the
        # side stream's projection launch
        gkey = tuple((g.data_ptr(), g._version) for g in group_biases) + (str(out.device), cache_epoch())
        ghit = _STACKED_GB.get(gkey)
        if ghit is None:
            if len(_STACKED_GB) >= 8:
                _STACKED_GB.pop(next(iter(_STACKED_GB)))
            ghit = _STACKED_GB[gkey] = (torch.cat(list(group_biases), 2).contiguous(), list(group_biases))
        gb_keep = ghit[0]
        G = gb_keep.shape[1]
        gb_ptrs = (ctypes.c_void_p * S)(*[gb_keep[s].data_ptr() for s in range(S)])
    arr64 = lambda v: (ctypes.c_int64 * S)(*[int(x) for x in v])
    a_ptrs = (ctypes.c_void_p * S)(*[a.data_ptr() for a in a_list])
    with torch.cuda.device(out.device), _timed('value_proj'):
        rc = _lib.lib().occ_value_proj_bf16_planes(
            i32(S), a_ptrs, arr64([a.stride(0) for a in a_list]), arr64([a.shape[0] for a in a_list]),
            arr64(rows_per_group), arr64(out_row0), gb_ptrs, i32(G), ptr(hit[1]), ptr(out),
            i32(2 if out_q16 else 1 if out_half else 0),
            i64(N), i32(K), i32(P), i32(N), i64(out.stride(0)), i64(out_group_rows), ptr(out_scale),
            stream_ptr(out.device))
    _lib.check(rc, "value_proj_bf16_planes")
    return out


def linear(a, weight, bias=None, a2=None, a2_add=None, act=None, residual=None, ln=None,
           precision=None):
    """out = LayerNorm(residual + act([a | a2 (+ a2_add)] @ weight^T + bias)) on the f32 matrix cores.

    a (…, K1); a2 / a2_add (…, K2) optional second K segment (+ addend); weight (N, K1+K2) and bias (N)
    in torch Linear layout; act None | 'relu'; residual (…, N); ln = (gamma, beta, eps) or an
    nn.LayerNorm; precision 'f32' | 'bf16x3' (None = LINEAR_PRECISION).  -> (…, N) float32.  Raises
    OccAmdUnsupported for shapes without an MFMA kernel."""
    precision = precision or LINEAR_PRECISION
    if precision not in ("f32", "bf16x3"):
        raise OccAmdError(f"linear: unknown precision {precision!r}")
    a_, M, K1, lda1 = _rows2d("a", a)
    K2, lda2 = 0, 0
    if a2 is not None:
        a2_, M2, K2, lda2 = _rows2d("a2", a2)
        if M2 != M:
            raise OccAmdError("linear: a and a2 differ in rows")
        if a2_add is not None:
            _, M3, _, lda3 = _rows2d("a2_add", a2_add, K2)
            if M3 != M or lda3 != lda2:
                raise OccAmdError("linear: a2_add must match a2's shape and row stride")
    elif a2_add is not None:
        raise OccAmdError("linear: a2_add without a2")
    _need_cuda_f32("weight", weight)
    N = weight.shape[0]
    if weight.dim() != 2 or weight.shape[1] != K1 + K2:
        raise OccAmdError("linear: weight must be (N, K1+K2)")
    if bias is not None:
        _need_cuda_f32("bias", bias)
        if bias.numel() != N:
            raise OccAmdError("linear: bias must have N entries")
    ldres = 0
    if residual is not None:
        _, Mr, _, ldres = _rows2d("residual", residual, N)
        if Mr != M:
            raise OccAmdError("linear: residual differs in rows")
    g = b = None
    eps = 0.0
    if ln is not None:
        if isinstance(ln, torch.nn.LayerNorm):
            if tuple(ln.normalized_shape) != (N,) or ln.weight is None or ln.bias is None:
                raise OccAmdUnsupported("linear: LayerNorm must be affine over the N outputs")
            g, b, eps = ln.weight, ln.bias, ln.eps
        else:
            g, b, eps = ln
        _need_cuda_f32("ln_gamma", g)
        _need_cuda_f32("ln_beta", b)
    if act not in (None, 'relu'):
        raise OccAmdError("linear: act must be None or 'relu'")
    if not weight.is_contiguous():
        raise OccAmdError("linear: weight must be contiguous")
    wdev = linear_pack_weight_bf16x3(weight) if precision == "bf16x3" and (K1 + K2) % 16 == 0 else weight
    out = torch.empty(a.shape[:-1] + (N,), dtype=torch.float32, device=a.device)
    if _TIMING is not None and (_TIMING_ONLY is None or 'linear' in _TIMING_ONLY):
        _TIMING.setdefault('linear_flops', []).append(2.0 * M * N * (K1 + K2))
    fn = _lib.lib().occ_linear_f32 if wdev is weight else _lib.lib().occ_linear_bf16x3_f32
    with torch.cuda.device(a.device), _timed('linear'):
        rc = fn(ptr(a_), i64(lda1), i32(K1), ptr(a2), ptr(a2_add), i64(lda2), i32(K2), ptr(wdev),
                ptr(bias), i32(1 if act == 'relu' else 0), ptr(residual), i64(ldres), ptr(g), ptr(b),
                f32(float(eps)), ptr(out), i64(N), i32(M), i32(N), stream_ptr(a.device))
    _lib.check(rc, "linear")
    return out


# ---- row-local Linear chains (csrc/linear_chain_x3.hip) ---------------------------------------------------------------
# OCC_LINEAR_CHAIN=0 keeps the encoder on one launch per Linear (the round-3 path); the chain kernels need
# embed_dims = 256 and the bf16x3 precision mode
LINEAR_CHAIN = os.environ.get("OCC_LINEAR_CHAIN", "1") != "0"
_CHAIN_PACKS = {}       # identity key of the stage weights / biases -> (sources kept alive, packed buffer)


def _ident(t):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape))


def linear_chain_pack(weights):
    """[(N_i, K_i) float32 weights in consumption order] -> one int16 buffer: every weight packed by
    occ_linear_chain_pack_bf16x3 (256-row groups, 16 KB per k-step, zero rows beyond N), concatenated.  Cached on the
    weights' identities (address, version) and the cache epoch."""
    key = ("w",) + tuple(_ident(w) for w in weights) + (str(weights[0].device), cache_epoch())
    hit = _CHAIN_PACKS.get(key)
    if hit is not None:
        return hit[1]
    lib = _lib.lib()
    lib.occ_linear_chain_packed_bytes.restype = ctypes.c_int64
    sizes = []
    for w in weights:
        _need_cuda_f32("weight", w)
        if w.dim() != 2 or w.shape[1] % 16:
            raise OccAmdUnsupported("linear_chain_pack: weights must be (N, K) with K % 16 == 0")
        sizes.append(int(lib.occ_linear_chain_packed_bytes(i32(w.shape[0]), i32(w.shape[1]))) // 2)
    packed = torch.empty(sum(sizes), dtype=torch.int16, device=weights[0].device)
    off = 0
    with torch.cuda.device(packed.device):
        for w, n in zip(weights, sizes):
            rc = lib.occ_linear_chain_pack_bf16x3(ptr(w), ctypes.c_void_p(packed.data_ptr() + off * 2), i32(w.shape[0]),
                                                  i32(w.shape[1]), stream_ptr(packed.device))
            _lib.check(rc, "linear_chain_pack_bf16x3")
            off += n
    if len(_CHAIN_PACKS) >= 128:
        _CHAIN_PACKS.pop(next(iter(_CHAIN_PACKS)))
    _CHAIN_PACKS[key] = (list(weights), packed)      # the sources stay referenced: their addresses cannot be recycled
    return packed


def _chain_bias(parts, device):
    """[(bias tensor or None, padded length)] -> one float32 vector (zeros where a bias is None / beyond its length)."""
    key = ("b",) + tuple((_ident(b), n) for b, n in parts) + (str(device), cache_epoch())
    hit = _CHAIN_PACKS.get(key)
    if hit is not None:
        return hit[1]
    out = torch.zeros(sum(n for _, n in parts), dtype=torch.float32, device=device)
    off = 0
    for b, n in parts:
        if b is not None:
            _need_cuda_f32("bias", b)
            out[off:off + b.numel()] = b.detach().reshape(-1)
        off += n
    if len(_CHAIN_PACKS) >= 128:
        _CHAIN_PACKS.pop(next(iter(_CHAIN_PACKS)))
    _CHAIN_PACKS[key] = ([b for b, _ in parts], out)
    return out


def _ln_params(ln, n=256):
    if isinstance(ln, torch.nn.LayerNorm):
        if tuple(ln.normalized_shape) != (n,) or ln.weight is None or ln.bias is None:
            raise OccAmdUnsupported("linear chain: LayerNorm must be affine over the 256 outputs")
        g, b, eps = ln.weight, ln.bias, ln.eps
    else:
        g, b, eps = ln
    _need_cuda_f32("ln_gamma", g)
    _need_cuda_f32("ln_beta", b)
    if g.numel() != n or b.numel() != n:
        raise OccAmdUnsupported("linear chain: LayerNorm must span the 256 outputs")
    return g, b, float(eps)


def _chain_rows(name, t):
    t_, M, K, ld = _rows2d(name, t)
    if K != 256:
        raise OccAmdUnsupported(f"linear chain: {name} must have 256 columns (embed_dims = 256), got {K}")
    return t_, M, ld


def _chain_time(flops):
    if _TIMING is not None and (_TIMING_ONLY is None or 'linear' in _TIMING_ONLY):
        _TIMING.setdefault('linear_flops', []).append(float(flops))


def linear_ln_chain(a, residual, w1, b1, ln, w2, b2, act2=None):
    """Program A of csrc/linear_chain_x3.hip in ONE launch:  y = LayerNorm(a @ w1^T + b1 + residual),
    z = act2(y @ w2^T + b2).  a, residual (…, 256) float32 device tensors; w1 (256, 256); w2 (n2, 256) with
    n2 % 32 == 0; ln = nn.LayerNorm(256) or (gamma, beta, eps); act2 None | 'relu'.  -> (y (…, 256), z (…, n2)).
    Raises OccAmdUnsupported for other shapes (the caller keeps occ `linear`)."""
    if LINEAR_PRECISION != "bf16x3":
        raise OccAmdUnsupported("linear chain: bf16x3 precision mode only")
    a_, M, lda = _chain_rows("a", a)
    r_, Mr, ldres = _chain_rows("residual", residual)
    if Mr != M:
        raise OccAmdError("linear_ln_chain: residual differs in rows")
    if tuple(w1.shape) != (256, 256) or w2.dim() != 2 or w2.shape[1] != 256 or w2.shape[0] % 32:
        raise OccAmdUnsupported("linear_ln_chain: w1 must be (256, 256) and w2 (n2 % 32 == 0, 256)")
    if act2 not in (None, 'relu'):
        raise OccAmdError("linear_ln_chain: act2 must be None or 'relu'")
    g, b, eps = _ln_params(ln)
    n2 = w2.shape[0]
    wp = linear_chain_pack([w1, w2])
    bias = _chain_bias([(b1, 256), (b2, (n2 + 255) // 256 * 256)], a.device)
    y = torch.empty(a.shape[:-1] + (256,), dtype=torch.float32, device=a.device)
    z = torch.empty(a.shape[:-1] + (n2,), dtype=torch.float32, device=a.device)
    _chain_time(2.0 * M * 256 * (256 + n2))
    with torch.cuda.device(a.device), _timed('linear'):
        rc = _lib.lib().occ_linear_ln_chain_bf16x3_f32(
            ptr(a_), i64(lda), ptr(r_), i64(ldres), ptr(wp), ptr(bias), ptr(g), ptr(b), f32(eps), ptr(y), i64(256),
            ptr(z), i64(n2), i32(n2), i32(1 if act2 == 'relu' else 0), i32(M), stream_ptr(a.device))
    _lib.check(rc, "linear_ln_chain")
    return y, z


def _chain_out(name, t, M, ncols):
    """caller-provided output rows (a row band of a larger buffer): -> leading dimension"""
    _, Mo, _, ld = _rows2d(name, t, ncols)
    if Mo != M:
        raise OccAmdError(f"linear chain: {name} differs in rows")
    return ld


def encoder_ffn_chain(a, residual, wo, bo, ln1, w1, b1, w2, b2, ln2, tail=None, out=None):
    """Program B of csrc/linear_chain_x3.hip in ONE launch:
        x2 = LayerNorm1(a @ wo^T + bo + residual);  y = LayerNorm2(relu(x2 @ w1^T + b1) @ w2^T + b2 + x2)
    and, with tail = (wq (nq, 256), q_term (…, nq) or None, wv (256, 256), bv):  zq = y @ wq^T + q_term,
    zv = y @ wv^T + bv.  a, residual (…, 256); wo (256, 256); w1 (512, 256); w2 (256, 512).
    -> (y, zq, zv)  (zq = zv = None without a tail).  Raises OccAmdUnsupported for other shapes.
    out = (y, zq, zv) caller-provided result rows (zq / zv None without a tail), e.g. row bands of full-size buffers."""
    if LINEAR_PRECISION != "bf16x3":
        raise OccAmdUnsupported("linear chain: bf16x3 precision mode only")
    a_, M, lda = _chain_rows("a", a)
    r_, Mr, ldres = _chain_rows("residual", residual)
    if Mr != M:
        raise OccAmdError("encoder_ffn_chain: residual differs in rows")
    if tuple(wo.shape) != (256, 256) or tuple(w1.shape) != (512, 256) or tuple(w2.shape) != (256, 512):
        raise OccAmdUnsupported("encoder_ffn_chain: needs output_proj (256, 256) and a 256 -> 512 -> 256 FFN")
    g1, be1, eps1 = _ln_params(ln1)
    g2, be2, eps2 = _ln_params(ln2)
    weights = [wo, w1, w2]
    biases = [(bo, 256), (b1, 512), (b2, 256)]
    zq = zv = q_term = None
    nq, ldq = 0, 0
    if tail is not None:
        wq, q_term, wv, bv = tail
        nq = wq.shape[0]
        if wq.dim() != 2 or wq.shape[1] != 256 or nq > 256 or nq % 64 or tuple(wv.shape) != (256, 256):
            raise OccAmdUnsupported("encoder_ffn_chain: tail needs wq (nq <= 256, nq % 64 == 0, 256) and wv (256, 256)")
        if q_term is not None:
            _, Mq, _, ldq = _rows2d("q_term", q_term, nq)
            if Mq != M:
                raise OccAmdError("encoder_ffn_chain: q_term differs in rows")
        weights += [wq, wv]
        biases += [(None, 256), (bv, 256)]
        if out is None:
            zq = torch.empty(a.shape[:-1] + (nq,), dtype=torch.float32, device=a.device)
            zv = torch.empty(a.shape[:-1] + (256,), dtype=torch.float32, device=a.device)
    wp = linear_chain_pack(weights)
    bias = _chain_bias(biases, a.device)
    ldy, ldzq, ldzv = 256, nq, 256
    if out is None:
        y = torch.empty(a.shape[:-1] + (256,), dtype=torch.float32, device=a.device)
    else:
        y = out[0]
        ldy = _chain_out("out y", y, M, 256)
        if tail is not None:
            zq, zv = out[1], out[2]
            ldzq, ldzv = _chain_out("out zq", zq, M, nq), _chain_out("out zv", zv, M, 256)
    _chain_time(2.0 * M * 256 * (256 + 512 + 512 + (nq + 256 if tail is not None else 0)))
    with torch.cuda.device(a.device), _timed('linear'):
        rc = _lib.lib().occ_encoder_ffn_chain_bf16x3_f32(
            ptr(a_), i64(lda), ptr(r_), i64(ldres), ptr(wp), ptr(bias), ptr(g1), ptr(be1), f32(eps1), ptr(g2),
            ptr(be2), f32(eps2), ptr(y), i64(ldy), ptr(q_term), i64(ldq), ptr(zq), i64(ldzq), i32(nq), ptr(zv),
            i64(ldzv), i32(M), stream_ptr(a.device))
    _lib.check(rc, "encoder_ffn_chain")
    return y, zq, zv


def linear_pair_chain(a, wq, q_term, wv, bv):
    """Program C of csrc/linear_chain_x3.hip: zq = a @ wq^T (+ q_term), zv = a @ wv^T + bv in ONE launch (the rows are
    read once).  a (…, 256); wq (nq <= 256, nq % 64 == 0, 256); wv (256, 256).  -> (zq, zv)."""
    if LINEAR_PRECISION != "bf16x3":
        raise OccAmdUnsupported("linear chain: bf16x3 precision mode only")
    a_, M, lda = _chain_rows("a", a)
    nq = wq.shape[0]
    if wq.dim() != 2 or wq.shape[1] != 256 or nq > 256 or nq % 64 or tuple(wv.shape) != (256, 256):
        raise OccAmdUnsupported("linear_pair_chain: needs wq (nq <= 256, nq % 64 == 0, 256) and wv (256, 256)")
    ldq = 0
    if q_term is not None:
        _, Mq, _, ldq = _rows2d("q_term", q_term, nq)
        if Mq != M:
            raise OccAmdError("linear_pair_chain: q_term differs in rows")
    wp = linear_chain_pack([wq, wv])
    bias = _chain_bias([(None, 256), (bv, 256)], a.device)
    zq = torch.empty(a.shape[:-1] + (nq,), dtype=torch.float32, device=a.device)
    zv = torch.empty(a.shape[:-1] + (256,), dtype=torch.float32, device=a.device)
    _chain_time(2.0 * M * 256 * (nq + 256))
    with torch.cuda.device(a.device), _timed('linear'):
        rc = _lib.lib().occ_linear_pair_chain_bf16x3_f32(
            ptr(a_), i64(lda), ptr(wp), ptr(bias), ptr(q_term), i64(ldq), ptr(zq), i64(nq), i32(nq), ptr(zv), i64(256),
            i32(M), stream_ptr(a.device))
    _lib.check(rc, "linear_pair_chain")
    return zq, zv


def linear_wgrad(dy, x, with_bias=True):
    """Weight / bias gradient of out = x @ W^T + b on the bf16x3 matrix-core kernel (csrc/linear_wgrad.hip):
    dW (N, K) = dy^T @ x, db (N) = dy.sum(rows).  dy (…, N), x (…, K) float32 device tensors with the same leading
    shape and unit column stride.  Deterministic (chunked reduction, fixed order)."""
    def rows(name, t):          # dword loads: any row stride / alignment, unit column stride
        _need_cuda_f32(name, t, contiguous=False)
        if t.dim() == 2 and t.stride(1) == 1:
            return t, t.shape[0], t.shape[1], t.stride(0)
        if t.is_contiguous():
            return t, t.numel() // t.shape[-1], t.shape[-1], t.shape[-1]
        raise OccAmdError(f"linear_wgrad: {name} must be contiguous or a 2-D row-strided view")
    dy_, M, N, lddy = rows("dy", dy)
    x_, Mx, K, ldx = rows("x", x)
    if Mx != M:
        raise OccAmdError("linear_wgrad: dy and x differ in rows")
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if with_bias else None
    lib = _lib.lib()
    lib.occ_linear_wgrad_workspace_bytes.restype = ctypes.c_int64
    nbytes = int(lib.occ_linear_wgrad_workspace_bytes(i32(M), i32(N), i32(K)))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device), _timed('linear_wgrad'):
        rc = lib.occ_linear_wgrad_bf16x3_f32(ptr(dy_), i64(lddy), ptr(x_), i64(ldx), ptr(dw), ptr(db), ptr(ws),
                                             i64(nbytes), i32(M), i32(N), i32(K), stream_ptr(dy.device))
    _lib.check(rc, "linear_wgrad")
    return dw, db


class Conv3dX3Function(torch.autograd.Function):
    """3x3x3 / stride 1 / pad 1 Conv3d (no bias, Cout = 32) with forward AND backward on this library's kernels — the
    training-mode replacement for the decoder's nn.Conv3d (reference transformer_occ.py:106-126; round 3 trained it on
    MIOpen under bf16 autocast, narrower than the reference's fp32):
      forward  occ_conv3d_bn_relu_bf16x3_f32 with scale 1 / shift 0 / no ReLU (BatchNorm and ReLU stay autograd ops);
      dx       the same kernel on the flipped, transposed weight (rows beyond Cin zero: the kernel writes 32 channels);
      dW       occ_conv3d_wgrad_bf16x3_f32 on zero-padded copies of x and dy: in the padded grid a tap is a constant
               row offset, so dW = dy_pad^T @ im2col(x_pad) is ONE launch of the N = 32 wgrad kernel (deterministic).
    x: in_layout 0 (B, Y, X, Z, Cin) or 1 (B, Y*X, Cin*Z) (the lifter view of the BEV embedding); -> (B, Y, X, Z, 32)."""

    @staticmethod
    def forward(ctx, x, weight, Z, Y, X, in_layout):
        cout, cin = weight.shape[:2]
        one = torch.ones(cout, dtype=torch.float32, device=x.device)
        zero = torch.zeros(cout, dtype=torch.float32, device=x.device)
        out = conv3d_bn_relu(x.contiguous(), conv3d_pack_weight(weight.detach().contiguous()), one, zero, Z, Y, X, cin, cout,
                             in_layout, relu=False)
        ctx.save_for_backward(x, weight)
        ctx.geom = (int(Z), int(Y), int(X), int(in_layout))
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight = ctx.saved_tensors
        Z, Y, X, in_layout = ctx.geom
        cout, cin = weight.shape[:2]
        B = x.shape[0]
        gout = gout.contiguous()
        gx = gw = None
        one = torch.ones(32, dtype=torch.float32, device=x.device)
        zero = torch.zeros(32, dtype=torch.float32, device=x.device)
        if ctx.needs_input_grad[0]:
            wt = weight.detach().flip(2, 3, 4).transpose(0, 1)               # (cin, cout, 3, 3, 3): dY -> dX
            parts = []
            for c0 in range(0, cin, 32):                                     # the kernel writes 32 channels per launch
                wg = wt[c0:c0 + 32]
                if wg.shape[0] < 32:
                    wg = torch.cat([wg, wg.new_zeros((32 - wg.shape[0],) + tuple(wg.shape[1:]))], 0)
                g32 = conv3d_bn_relu(gout, conv3d_pack_weight(wg.contiguous()), one, zero, Z, Y, X, cout, 32, 0, relu=False)
                parts.append(g32[..., :min(32, cin - c0)])
            gx = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
            gx = gx.permute(0, 1, 2, 4, 3).reshape(B, Y * X, cin * Z) if in_layout == 1 else gx.contiguous()
        if ctx.needs_input_grad[1]:
            xc = x.detach()
            if in_layout == 1:
                xc = xc.view(B, Y, X, cin, Z).permute(0, 1, 2, 4, 3)                # (B, Y, X, Z, cin) view
            xp = torch.nn.functional.pad(xc, (0, 0, 1, 1, 1, 1, 1, 1)).contiguous()      # (B, Y+2, X+2, Z+2, cin)
            gp = torch.nn.functional.pad(gout, (0, 0, 1, 1, 1, 1, 1, 1)).contiguous()    # (B, Y+2, X+2, Z+2, 32)
            lib = _lib.lib()
            lib.occ_conv3d_wgrad_workspace_bytes.restype = ctypes.c_int64
            nbytes = int(lib.occ_conv3d_wgrad_workspace_bytes(i32(B), i32(Z), i32(Y), i32(X), i32(cin)))
            if nbytes <= 0:
                raise OccAmdUnsupported("conv3d wgrad: grid beyond the kernel's 32-bit row range")
            ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
            dw = torch.empty((cout, 27, cin), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _timed('conv3d_wgrad'):
                rc = lib.occ_conv3d_wgrad_bf16x3_f32(ptr(gp), ptr(xp), ptr(dw), ptr(ws), i64(nbytes), i32(B), i32(Z),
                                                     i32(Y), i32(X), i32(cin), stream_ptr(x.device))
            _lib.check(rc, "conv3d_wgrad")
            gw = dw.permute(0, 2, 1).reshape(cout, cin, 3, 3, 3)
        return gx, gw, None, None, None, None


def conv3d_autograd(x, weight, Z, Y, X, in_layout=0):
    """Differentiable Conv3d(k3, s1, p1, no bias, Cout = 32) on the bf16x3 kernels (Conv3dX3Function)."""
    cout, cin = weight.shape[:2]
    if cout != 32 or not (cin % 16 == 0 or cin == 8) or Z not in (4, 8, 16, 32) or CONV3D_PRECISION != "bf16x3":
        raise OccAmdUnsupported("conv3d_autograd: needs Cout = 32, Cin % 16 == 0 or Cin == 8, Z in {4, 8, 16, 32}")
    return Conv3dX3Function.apply(x, weight, Z, Y, X, in_layout)


class LinearX3Function(torch.autograd.Function):
    """y = act(x @ W^T + b) with forward AND backward on the bf16x3 kernels: forward = linear(), dx = linear(dy, W^T),
    dW / db = linear_wgrad().  The training-mode replacement for F.linear at the encoder's Linear call sites
    (reference: nn.Linear + ATen autograd; temporal_self_attention.py:197-209, spatial_cross_attention.py:334-341,
    mmcv FFN)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        out = linear(x, weight, bias, act=act, precision='bf16x3')
        ctx.act = act
        ctx.save_for_backward(x, weight, out if act == 'relu' else None)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, weight, out = ctx.saved_tensors
        gout = gout.contiguous()
        if ctx.act == 'relu':
            gout = gout * (out > 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = weight.t().contiguous()
            wt._occ_no_cache = True                 # a temporary: packed for this call only
            gx = linear(gout, wt, None, precision='bf16x3')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = linear_wgrad(gout, x, with_bias=ctx.has_bias)
        return gx, gw, gb, None


def rows_gather_sum(x, index):
    """out[b, r] = sum_k x[b, index[r, k]] over the entries with index >= 0 (csrc/rows_index.hip).  x (B, rows_in, F)
    float32 contiguous, index (rows_out, K) int64 on the device -> (B, rows_out, F)."""
    _need_cuda_f32("x", x)
    if x.dim() != 3 or index.dim() != 2 or index.dtype != torch.int64 or not index.is_cuda \
            or not index.is_contiguous():
        raise OccAmdError("rows_gather_sum: x must be (B, rows, F), index a contiguous (rows_out, K) int64 device "
                          "tensor")
    B, rows_in, F = x.shape
    rows_out, K = index.shape
    out = torch.empty((B, rows_out, F), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed('rows_gather_sum'):
        rc = _lib.lib().occ_rows_gather_sum_f32(ptr(x), i64(rows_in * F), ptr(index), i32(K), ptr(out), i32(B),
                                                i64(rows_out), i64(rows_in), i32(F), stream_ptr(x.device))
    _lib.check(rc, "rows_gather_sum")
    return out


class RowsGatherSumFunction(torch.autograd.Function):
    """rows_gather_sum with its gradient expressed as the gather-sum over the INVERSE index (the caller supplies
    both maps; they must be each other's transpose as relations)."""

    @staticmethod
    def forward(ctx, x, index, inverse_index):
        ctx.save_for_backward(inverse_index)
        return rows_gather_sum(x.contiguous(), index)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        (inverse_index,) = ctx.saved_tensors
        return rows_gather_sum(gout.contiguous(), inverse_index), None, None


class SCAPrepFunction(torch.autograd.Function):
    """(loc, attn) of SpatialCrossAttention's training path from the per-query Linear outputs: rebatch + softmax +
    offset normalisation + anchor add as one kernel, and its gradient as one kernel (csrc/sca_prep.hip).
    proj (B, Q, 3*M*L*P) float32, row_to_query (R, 1) / query_to_rows (Q, K) int64 (the maps of
    RowsGatherSumFunction), ref_rb (B, R, Z, 2), spatial_shapes (L, 2) int64 ->
    loc (B, R, M, L, P, 2), attn (B, R, M, L, P)."""

    @staticmethod
    def forward(ctx, proj, row_to_query, query_to_rows, ref_rb, spatial_shapes, M, L, P):
        _need_cuda_f32("proj", proj)
        _need_cuda_f32("ref_rb", ref_rb)
        B, Q, ld = proj.shape
        R = row_to_query.shape[0]
        Z = ref_rb.shape[2]
        loc = torch.empty((B, R, M, L, P, 2), dtype=torch.float32, device=proj.device)
        attn = torch.empty((B, R, M, L, P), dtype=torch.float32, device=proj.device)
        with torch.cuda.device(proj.device), _timed('sca_prep'):
            rc = _lib.lib().occ_sca_prep_forward_f32(ptr(proj), i64(Q * ld), i32(ld), ptr(row_to_query), ptr(ref_rb),
                                                     ptr(spatial_shapes), ptr(loc), ptr(attn), i32(B), i64(R),
                                                     i32(M), i32(L), i32(P), i32(Z), stream_ptr(proj.device))
        _lib.check(rc, "sca_prep_forward")
        ctx.save_for_backward(attn, query_to_rows, spatial_shapes)
        ctx.dims = (B, Q, ld, R, M, L, P)
        return loc, attn

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loc, g_attn):
        attn, query_to_rows, spatial_shapes = ctx.saved_tensors
        B, Q, ld, R, M, L, P = ctx.dims
        g_loc = (torch.zeros_like(attn).unsqueeze(-1).expand(*attn.shape, 2) if g_loc is None else g_loc).contiguous()
        g_attn = (torch.zeros_like(attn) if g_attn is None else g_attn).contiguous()
        dproj = torch.empty((B, Q, ld), dtype=torch.float32, device=attn.device)
        if ld > 3 * M * L * P:
            dproj.zero_()
        with torch.cuda.device(attn.device), _timed('sca_prep_bwd'):
            rc = _lib.lib().occ_sca_prep_backward_f32(ptr(g_loc), ptr(g_attn), ptr(attn), ptr(query_to_rows),
                                                      i32(query_to_rows.shape[1]), ptr(spatial_shapes), ptr(dproj),
                                                      i32(ld), i32(B), i64(R), i64(Q), i32(M), i32(L), i32(P),
                                                      stream_ptr(attn.device))
        _lib.check(rc, "sca_prep_backward")
        return dproj, None, None, None, None, None, None, None


class LinearWgradFunction(torch.autograd.Function):
    """y = x @ W^T + b for the shapes linear_bf16x3 does not cover (the occupancy heads' 64 -> 17 and 64 -> 2 layers:
    N not a multiple of 16) but whose WEIGHT gradient is the expensive part: 640 000 voxels reduced into a 17 x 64
    matrix is a 1.0-1.5 ms fp32 library GEMM plus a 1.6 ms column reduction for the bias in ATen; linear_wgrad does
    both in one pass.  Forward and dx stay library GEMMs (thin, fast)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, weight = ctx.saved_tensors
        gout = gout.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gout.matmul(weight)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = linear_wgrad(gout, x if x.is_contiguous() else x.contiguous(), with_bias=ctx.has_bias)
        return gx, gw, gb


TRAIN_LINEAR = os.environ.get("OCC_TRAIN_LINEAR", "x3")      # 'x3' (own kernels) or 'torch' (F.linear + ATen autograd)
TRAIN_WGRAD_ONLY = os.environ.get("OCC_TRAIN_WGRAD_ONLY", "1") != "0"   # LinearWgradFunction for the other shapes


class DropoutAddLayerNormFunction(torch.autograd.Function):
    """y = LayerNorm(dropout(x, p) + residual) as ONE autograd node (csrc/ln_dropout_train.hip): the tail of every attention /
    FFN block of a BEVFormerLayer in training (reference: encoder.py:377-404, spatial_cross_attention.py:173-175,
    temporal_self_attention.py:270-272, mmcv FFN).  One launch forward, one (+ a small fixed-order reduce) backward instead
    of ATen's dropout + add + LayerNorm and masked scale + two LayerNorm kernels + the residual fork's add.  The keep mask is
    a counter-based hash of (seed, element index): never stored, regenerated in backward; the seed comes from torch's CPU
    generator (reproducible under torch.manual_seed, no device sync)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, p):
        shape = x.shape
        C = shape[-1]
        x2, r2 = x.reshape(-1, C).contiguous(), residual.reshape(-1, C).contiguous()
        rows = x2.shape[0]
        z, y = torch.empty_like(x2), torch.empty_like(x2)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()) if p > 0 else 0
        with torch.cuda.device(x.device):
            rc = _lib.lib().occ_dropout_add_ln_fwd_f32(ptr(x2), ptr(r2), ptr(weight), ptr(bias), f32(eps), f32(p),
                                                       ctypes.c_uint64(seed), ptr(z), ptr(y), ptr(stats), i64(rows), i32(C),
                                                       stream_ptr(x.device))
        _lib.check(rc, "dropout_add_ln_fwd")
        ctx.save_for_backward(z, stats, weight)
        ctx.cfg = (float(p), seed, tuple(shape), tuple(residual.shape))
        return y.view(shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        z, stats, weight = ctx.saved_tensors
        p, seed, shape, rshape = ctx.cfg
        rows, C = z.shape
        gy2 = gy.reshape(rows, C).contiguous()
        lib = _lib.lib()
        lib.occ_dropout_add_ln_bwd_partial_floats.restype = ctypes.c_int64
        partial = torch.empty(int(lib.occ_dropout_add_ln_bwd_partial_floats(i64(rows), i32(C))), dtype=torch.float32,
                              device=z.device)
        gres = torch.empty_like(z)
        gx = torch.empty_like(z) if p > 0 else gres
        ggb = torch.empty(2 * C, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            rc = lib.occ_dropout_add_ln_bwd_f32(ptr(gy2), ptr(z), ptr(stats), ptr(weight), f32(p), ctypes.c_uint64(seed),
                                                ptr(gx) if p > 0 else ptr(None), ptr(gres), ptr(partial), ptr(ggb), i64(rows),
                                                i32(C), stream_ptr(z.device))
        _lib.check(rc, "dropout_add_ln_bwd")
        return gx.view(shape), gres.view(rshape), ggb[:C], ggb[C:], None, None


def dropout_add_layernorm_ok(x, residual, norm):
    """True when DropoutAddLayerNormFunction covers this site (fp32 device rows of 256 columns, an affine LayerNorm over them)."""
    return (isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and tuple(norm.normalized_shape) == (256,) and x.is_cuda and x.dtype == torch.float32
            and residual.dtype == torch.float32 and x.shape == residual.shape and x.shape[-1] == 256
            and x.numel() < (1 << 32) and norm.weight.dtype == torch.float32)


def dropout_add_layernorm(x, residual, norm, p, training):
    """LayerNorm(dropout(x) + residual) on the fused training node."""
    return DropoutAddLayerNormFunction.apply(x, residual, norm.weight, norm.bias, float(norm.eps),
                                             float(p) if training else 0.0)


def linear_autograd(x, weight, bias=None, act=None):
    """F.linear(+ReLU) that is differentiable: on a float32 device tensor with supported shapes the forward and the
    backward run on the bf16x3 kernels; shapes the forward kernel does not cover keep the library forward and take
    only the weight/bias gradient kernel (many rows); anything else is F.linear (still on the device: no CPU
    branch here)."""
    dev_ok = (TRAIN_LINEAR == "x3" and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
              and weight.is_contiguous() and torch.is_grad_enabled() and not torch.is_autocast_enabled())
    if dev_ok and weight.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0:
        if not (x.is_contiguous() or (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0)):
            x = x.contiguous()
        return LinearX3Function.apply(x, weight, bias, act)
    if dev_ok and TRAIN_WGRAD_ONLY and act is None and x.numel() // x.shape[-1] >= 4096:
        return LinearWgradFunction.apply(x, weight, bias)
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if act == 'relu' else y


def dvr_render_forward(sigma, origin, points, tindex, grid=None, phase_name="test"):
    """Same call shape as the reference's `dvr.render_forward(sigma, origin, points, tindex, grid,
    phase_name)` (tools/ray_iou/lib/dvr/dvr.cpp:68-72; used at ray_metrics.py:116-123):
    sigma (N, T, Z, Y, X), origin (N, T, 3), points (N, M, >=3) in voxel units, tindex (N, M), all float32
    device tensors; grid = [T, Z, Y, X] (checked against sigma when given).
    -> (pred_dist (N, M), gt_dist (N, M), coord_index (N, M, 3)) float32."""
    for n, t in (("sigma", sigma), ("origin", origin), ("points", points), ("tindex", tindex)):
        _need_cuda_f32(n, t)
    if sigma.dim() != 5 or origin.dim() != 3 or points.dim() != 3 or tindex.dim() != 2:
        raise OccAmdError("dvr_render_forward: expected sigma (N,T,Z,Y,X), origin (N,T,3), points "
                          "(N,M,>=3), tindex (N,M)")
    N, T, Z, Y, X = sigma.shape
    M = points.shape[1]
    if grid is not None and [int(v) for v in grid] != [T, Z, Y, X]:
        raise OccAmdError(f"dvr_render_forward: grid {list(grid)} does not match sigma {[T, Z, Y, X]}")
    if (points.shape[0] != N or tuple(tindex.shape) != (N, M) or origin.shape[0] != N or
            origin.shape[2] != 3 or points.shape[2] < 3):
        raise OccAmdError("dvr_render_forward: inconsistent shapes")
    if phase_name not in ("test", "train"):
        raise OccAmdError(f"UNKNOWN PHASE NAME: {phase_name}")
    dev = sigma.device
    pred = torch.empty((N, M), dtype=torch.float32, device=dev)
    gt = torch.empty((N, M), dtype=torch.float32, device=dev)
    coord = torch.empty((N, M, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed('dvr_render_forward'):
        rc = _lib.lib().occ_dvr_render_forward_f32(
            ptr(sigma), ptr(origin), ptr(points), ptr(tindex), ptr(pred), ptr(gt), ptr(coord), i32(N),
            i32(T), i32(Z), i32(Y), i32(X), i32(M), i32(points.shape[2]),
            i32(1 if phase_name == "train" else 0), stream_ptr(dev))
    _lib.check(rc, "dvr_render_forward")
    return pred, gt, coord


def bias_act_nhwc_(x, bias, residual=None, relu=True):
    """In place x = relu?(x + bias[c] (+ residual)) on a channels_last bf16 (N, C, H, W) tensor (memory
    order N, H, W, C).  bias (C) float32.  Returns x."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("bias_act_nhwc_: x must be a channels_last bfloat16 device tensor")
    _need_cuda_f32("bias", bias)
    N, C, H, W = x.shape
    if bias.numel() != C:
        raise OccAmdError("bias_act_nhwc_: bias must have C entries")
    if residual is not None and not (residual.dtype == torch.bfloat16 and residual.shape == x.shape and
                                     residual.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("bias_act_nhwc_: residual must match x (channels_last bfloat16)")
    with torch.cuda.device(x.device):
        rc = _lib.lib().occ_bias_act_nhwc_bf16(ptr(x), ptr(bias), ptr(residual), i64(N * H * W), i32(C),
                                               i32(1 if relu else 0), stream_ptr(x.device))
    _lib.check(rc, "bias_act_nhwc_")
    return x


def bias_act_bwd_nhwc(grad_y, y=None, relu=True):
    """Backward of bias_act_nhwc_ in one pass: -> (g, bias_grad) with g = grad_y * (y > 0) (grad_y itself when not relu) and
    bias_grad (C) float32 = sum of g over N, H, W.  grad_y / y: channels_last bfloat16 (N, C, H, W) device tensors."""
    ok = lambda t: (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4
                    and t.is_contiguous(memory_format=torch.channels_last))
    if not ok(grad_y) or (relu and not (y is not None and ok(y) and y.shape == grad_y.shape)):
        raise OccAmdUnsupported("bias_act_bwd_nhwc: grad_y / y must be matching channels_last bfloat16 device tensors")
    N, C, H, W = grad_y.shape
    rows = N * H * W
    lib = _lib.lib()
    lib.occ_bias_act_bwd_partial_floats.restype = ctypes.c_int64
    nfl = int(lib.occ_bias_act_bwd_partial_floats(i64(rows), i32(C)))
    if nfl <= 0:
        raise OccAmdUnsupported(f"bias_act_bwd_nhwc: no kernel for C={C}")
    partial = torch.empty(nfl, dtype=torch.float32, device=grad_y.device)
    bias_grad = torch.empty(C, dtype=torch.float32, device=grad_y.device)
    g = torch.empty_like(grad_y) if relu else grad_y
    with torch.cuda.device(grad_y.device):
        rc = _lib.lib().occ_bias_act_bwd_nhwc_bf16(ptr(grad_y), ptr(y) if relu else ptr(None), ptr(g) if relu else ptr(None),
                                                   ptr(partial), ptr(bias_grad), i64(rows), i32(C), i32(1 if relu else 0),
                                                   stream_ptr(grad_y.device))
    _lib.check(rc, "bias_act_bwd_nhwc")
    return g, bias_grad


def conv_bn_fold_fwd(weight, gamma, beta, rstd, mean_rstd):
    """Eval-mode BatchNorm folded into a convolution weight, one launch: -> (w_folded fp32 (O, I, kh, kw) contiguous,
    w16 = the same as bfloat16 channels_last, bias (O) fp32)."""
    for n, t in (("weight", weight), ("gamma", gamma), ("beta", beta), ("rstd", rstd), ("mean_rstd", mean_rstd)):
        _need_cuda_f32(n, t)
    O, I, KH, KW = weight.shape
    wf = torch.empty_like(weight, memory_format=torch.contiguous_format)
    w16 = torch.empty((O, I, KH, KW), dtype=torch.bfloat16, device=weight.device, memory_format=torch.channels_last)
    b = torch.empty(O, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.lib().occ_conv_bn_fold_fwd_f32(ptr(weight), ptr(gamma), ptr(beta), ptr(rstd), ptr(mean_rstd), ptr(wf), ptr(w16),
                                                 ptr(b), i32(O), i32(I), i32(KH), i32(KW), stream_ptr(weight.device))
    _lib.check(rc, "conv_bn_fold_fwd")
    return wf, w16, b


def conv_bn_fold_bwd(grad_w16, weight, gamma, rstd, mean_rstd, grad_bias):
    """Chain rule of conv_bn_fold_fwd, one launch: grad_w16 (O, I, kh, kw) bfloat16 (any strides) -> (grad_weight fp32,
    grad_gamma fp32); grad_beta is grad_bias."""
    if not (grad_w16.is_cuda and grad_w16.dtype == torch.bfloat16 and grad_w16.shape == weight.shape):
        raise OccAmdUnsupported("conv_bn_fold_bwd: grad_w16 must be a bfloat16 device tensor of the weight's shape")
    for n, t in (("weight", weight), ("gamma", gamma), ("rstd", rstd), ("mean_rstd", mean_rstd), ("grad_bias", grad_bias)):
        _need_cuda_f32(n, t)
    O, I, KH, KW = weight.shape
    dW = torch.empty_like(weight, memory_format=torch.contiguous_format)
    dgamma = torch.empty(O, dtype=torch.float32, device=weight.device)
    so, si, sh, sw = grad_w16.stride()
    with torch.cuda.device(weight.device):
        rc = _lib.lib().occ_conv_bn_fold_bwd_f32(ptr(grad_w16), i64(so), i64(si), i64(sh), i64(sw), ptr(weight), ptr(gamma),
                                                 ptr(rstd), ptr(mean_rstd), ptr(grad_bias), ptr(dW), ptr(dgamma), i32(O), i32(I),
                                                 i32(KH), i32(KW), stream_ptr(weight.device))
    _lib.check(rc, "conv_bn_fold_bwd")
    return dW, dgamma


def stem_pack_weight(weight):
    """(64, 3, 7, 7) stem weight (BatchNorm folded) -> the fragment-ordered (64, 224) operand of stem_conv7x7_pool:
    column ky*32 + kx*4 + c holds weight[:, c, ky, kx]; the pad columns (kx = 7, c = 3) are zero."""
    if tuple(weight.shape) != (64, 3, 7, 7):
        raise OccAmdUnsupported("stem_pack_weight: expected a (64, 3, 7, 7) weight")
    w2 = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=weight.device)
    w2[:, :, :7, :3] = weight.float().permute(0, 2, 3, 1)
    return mfma_pack_b_frag(w2.reshape(64, 224))


def stem_conv7x7_pool(x, weight_frag, bias):
    """max_pool2d(relu(conv2d(x, W, stride 2, padding 3) + bias), 3, 2, 1) in one launch.
    x (N, 3, H, W) fp32 contiguous (NCHW); weight_frag from stem_pack_weight; bias (64) fp32
    -> (N, 64, Hp, Wp) channels_last bf16."""
    _need_cuda_f32("x", x)
    _need_cuda_f32("bias", bias)
    if x.dim() != 4 or x.shape[1] != 3 or not x.is_contiguous():
        raise OccAmdUnsupported("stem_conv7x7_pool: x must be a contiguous (N, 3, H, W) fp32 tensor")
    if weight_frag.numel() != 64 * 224 or bias.numel() != 64:
        raise OccAmdError("stem_conv7x7_pool: inconsistent shapes")
    N, _, H, W = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((N, 64, Hp, Wp), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device), _timed('bb_stem7x7_pool'):
        rc = _lib.lib().occ_stem_conv7x7_pool_f32_bf16(ptr(x), ptr(weight_frag), ptr(bias), ptr(out), i32(N),
                                                       i32(H), i32(W), stream_ptr(x.device))
    _note_flops('bb_stem7x7_pool', 2.0 * N * Hc * Wc * 64 * 147)
    _lib.check(rc, "stem_conv7x7_pool")
    return out


def stem_conv7x7_pool_u8(x_u8, weight_frag, bias, mean, std, to_rgb=False, size_divisor=32):
    """The stem fed with RAW camera images: x_u8 (N, Hs, Ws, 3) uint8 HWC on the device.  NormalizeMultiviewImage
    ((x[to_rgb ? 2-c : c] - mean[c]) / std[c]) and PadMultiViewImage (zeros to the next multiple of `size_divisor`;
    reference transform_3d.py:31-45,82-94) are applied while the input tile is staged.
    -> (N, 64, Hp, Wp) channels_last bf16, and the padded (H, W)."""
    if not (x_u8.is_cuda and x_u8.dtype == torch.uint8 and x_u8.dim() == 4 and x_u8.shape[-1] == 3
            and x_u8.is_contiguous()):
        raise OccAmdUnsupported("stem_conv7x7_pool_u8: x must be a contiguous (N, H, W, 3) uint8 device tensor")
    _need_cuda_f32("bias", bias)
    if weight_frag.numel() != 64 * 224 or bias.numel() != 64:
        raise OccAmdError("stem_conv7x7_pool_u8: inconsistent shapes")
    N, Hs, Ws, _ = x_u8.shape
    d = int(size_divisor)
    H, W = (Hs + d - 1) // d * d, (Ws + d - 1) // d * d
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((N, 64, Hp, Wp), dtype=torch.bfloat16, device=x_u8.device, memory_format=torch.channels_last)
    mean_c = (f32 * 3)(*[float(v) for v in mean])
    std_c = (f32 * 3)(*[float(v) for v in std])
    with torch.cuda.device(x_u8.device):
        rc = _lib.lib().occ_stem_conv7x7_pool_u8_bf16(ptr(x_u8), ptr(weight_frag), ptr(bias), ptr(out), i32(N),
                                                      i32(Hs), i32(Ws), i32(H), i32(W), mean_c, std_c,
                                                      i32(1 if to_rgb else 0), stream_ptr(x_u8.device))
    _lib.check(rc, "stem_conv7x7_pool_u8")
    return out, (H, W)


def bias_relu_maxpool_nhwc(y, bias):
    """max_pool2d(relu(y + bias), 3, stride 2, padding 1) in one launch on a channels_last bf16 activation
    (the ResNet stem's tail).  y (N, C, H, W) channels_last bf16 raw convolution output; bias (C) f32."""
    if not (y.is_cuda and y.dtype == torch.bfloat16 and y.dim() == 4
            and y.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("bias_relu_maxpool_nhwc: y must be a channels_last bfloat16 device tensor")
    _need_cuda_f32("bias", bias)
    N, C, H, W = y.shape
    if bias.numel() != C:
        raise OccAmdError("bias_relu_maxpool_nhwc: bias must have C entries")
    out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.bfloat16, device=y.device,
                      memory_format=torch.channels_last)
    with torch.cuda.device(y.device):
        rc = _lib.lib().occ_bias_relu_maxpool_nhwc_bf16(ptr(y), ptr(bias), ptr(out), i32(N), i32(H), i32(W),
                                                        i32(C), stream_ptr(y.device))
    _lib.check(rc, "bias_relu_maxpool_nhwc")
    return out


def conv1x1_pack_weight(weight2d):
    """(Cout, Cin) weight matrix (any float dtype, on the device) -> bf16 in MFMA B-fragment order (what
    conv1x1_nhwc takes; see mfma_pack_b_frag): a wave reads its operand straight from global memory."""
    cout, cin = weight2d.shape
    if cin % 32 or cout % 32:
        raise OccAmdUnsupported("conv1x1_pack_weight: Cin and Cout must be multiples of 32")
    return mfma_pack_b_frag(weight2d.float().contiguous())


def conv1x1_nhwc(x, weight_frag, bias, residual=None, relu=False, stride=1, residual_upsample2=False):
    """1x1 convolution + bias (+ residual) (+ ReLU) on a channels_last bf16 activation, one launch.
    x (N, Cin, H, W) channels_last bf16; weight_frag = conv1x1_pack_weight((Cout, Cin) matrix); bias (Cout) f32;
    residual (N, Cout, Ho, Wo) channels_last bf16 or None -> (N, Cout, Ho, Wo) channels_last bf16.
    residual_upsample2: residual is (N, Cout, Ho/2, Wo/2) and is added nearest-upsampled x2 (FPN top-down)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("conv1x1_nhwc: x must be a channels_last bfloat16 device tensor")
    _need_cuda_f32("bias", bias)
    N, Cin, H, W = x.shape
    Cout = bias.numel()
    if not (weight_frag.dtype == torch.int16 and weight_frag.dim() == 1 and weight_frag.is_contiguous()
            and weight_frag.numel() == Cout * Cin):
        raise OccAmdError("conv1x1_nhwc: weight must come from conv1x1_pack_weight (Cout*Cin bf16, fragment order)")
    s = int(stride)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    out = torch.empty((N, Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device,
                      memory_format=torch.channels_last)
    want = (N, Cout, Ho // 2, Wo // 2) if residual_upsample2 else tuple(out.shape)
    if residual_upsample2 and (residual is None or Ho % 2 or Wo % 2):
        raise OccAmdUnsupported("conv1x1_nhwc: an upsampled residual needs even output sizes")
    if residual is not None and not (residual.dtype == torch.bfloat16 and tuple(residual.shape) == want and
                                     residual.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("conv1x1_nhwc: residual must match the output (channels_last bfloat16)")
    with torch.cuda.device(x.device), _timed('bb_conv1x1'):
        rc = _lib.lib().occ_conv1x1_nhwc_bf16(ptr(x), ptr(weight_frag), ptr(bias), ptr(residual), ptr(out),
                                              i32(N), i32(H), i32(W), i32(Cin), i32(Cout), i32(s),
                                              i32(1 if relu else 0), i32(1 if residual_upsample2 else 0),
                                              stream_ptr(x.device))
    _note_flops('bb_conv1x1', 2.0 * N * Ho * Wo * Cin * Cout)
    _lib.check(rc, "conv1x1_nhwc")
    return out


def mfma_pack_b_frag(weight2d):
    """(N, K) float32 device matrix -> bf16 in MFMA B-fragment order (int16 storage, N*K elements): what the
    fused bottleneck kernel streams straight from global memory."""
    _need_cuda_f32("weight", weight2d)
    if weight2d.dim() != 2:
        raise OccAmdError("mfma_pack_b_frag: expected a (N, K) matrix")
    weight2d = weight2d.contiguous()
    n, k = weight2d.shape
    packed = torch.empty(n * k, dtype=torch.int16, device=weight2d.device)
    with torch.cuda.device(weight2d.device):
        rc = _lib.lib().occ_mfma_pack_b_frag_bf16(ptr(weight2d), ptr(packed), i32(n), i32(k),
                                                  stream_ptr(weight2d.device))
    _lib.check(rc, "mfma_pack_b_frag")
    return packed


def bottleneck64_pack(w1, b1, w2, b2, w3, b3, wds=None, bds=None):
    """Folded (BatchNorm already merged) weights of one 64-mid-channel bottleneck -> the operand set of
    bottleneck64_nhwc.  w1 (64, Cin[,1,1]), w2 (64, 64, 3, 3), w3 (256, 64[,1,1]), optional projection
    wds (256, Cin[,1,1]) / bds; biases float32.  Cin = 256 without projection, 64 with."""
    w1 = w1.float().reshape(w1.shape[0], -1)
    w3 = w3.float().reshape(w3.shape[0], -1)
    if tuple(w2.shape) != (64, 64, 3, 3) or w1.shape[0] != 64 or tuple(w3.shape) != (256, 64):
        raise OccAmdUnsupported("bottleneck64_pack: only 64 mid / 256 output channels")
    ds = wds is not None
    cin = w1.shape[1]
    if (ds and cin != 64) or (not ds and cin != 256):
        raise OccAmdUnsupported("bottleneck64_pack: Cin must be 256 (identity) or 64 (projection)")
    w2m = w2.float().permute(0, 2, 3, 1).reshape(64, 9 * 64)          # k = (ky*3 + kx)*64 + ci
    b3 = b3.float()
    if ds:
        w3 = torch.cat([w3, wds.float().reshape(256, 64)], dim=1)
        b3 = b3 + bds.float()
    return dict(w1=mfma_pack_b_frag(w1), b1=b1.float().contiguous(), w2=mfma_pack_b_frag(w2m),
                b2=b2.float().contiguous(), w3=mfma_pack_b_frag(w3), b3=b3.contiguous(), cin=cin, ds=ds)


def bottleneck64_nhwc(x, pack):
    """One whole stride-1 ResNet bottleneck (64 mid channels) on a channels_last bf16 activation, one launch.
    x (N, Cin, H, W) channels_last bf16; pack from bottleneck64_pack -> (N, 256, H, W) channels_last bf16."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("bottleneck64_nhwc: x must be a channels_last bfloat16 device tensor")
    N, Cin, H, W = x.shape
    if Cin != pack['cin']:
        raise OccAmdError("bottleneck64_nhwc: x has %d channels, the pack was built for %d" % (Cin, pack['cin']))
    out = torch.empty((N, 256, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device), _timed('bb_bottleneck64'):
        rc = _lib.lib().occ_bottleneck64_nhwc_bf16(ptr(x), ptr(pack['w1']), ptr(pack['b1']), ptr(pack['w2']),
                                                   ptr(pack['b2']), ptr(pack['w3']), ptr(pack['b3']), ptr(out),
                                                   i32(N), i32(H), i32(W), i32(Cin), i32(1 if pack['ds'] else 0),
                                                   stream_ptr(x.device))
    _note_flops('bb_bottleneck64', 2.0 * N * H * W * (Cin * 64 + 9 * 64 * 64 + 64 * 256 + (Cin * 256 if pack['ds'] else 0)))
    _lib.check(rc, "bottleneck64_nhwc")
    return out


def conv3x3_pack_weight(weight):
    """torch Conv2d weight (Cout, Cin, 3, 3) float32 -> packed bf16 [Cin/32][tap][co][32] (int16 storage)."""
    _need_cuda_f32("weight", weight)
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise OccAmdError("conv3x3_pack_weight: expected a (Cout, Cin, 3, 3) weight")
    cout, cin = weight.shape[:2]
    packed = torch.empty(weight.numel(), dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.lib().occ_conv3x3_pack_weight_bf16(ptr(weight), ptr(packed), i32(cout), i32(cin),
                                                     stream_ptr(weight.device))
    _lib.check(rc, "conv3x3_pack_weight")
    return packed


def conv3x3_nhwc(x, w_packed, bias, cout, relu=False, stride=1, amax=None):
    """3x3 / pad 1 / stride 1 or 2 convolution + bias (+ ReLU) on a channels_last bf16 activation, one launch.
    x (N, Cin, H, W) channels_last bf16; w_packed from conv3x3_pack_weight; bias (Cout) f32
    -> (N, Cout, (H-1)//stride+1, (W-1)//stride+1) channels_last bf16.
    amax: 8 int32 device words (new_absmax_words) the launch folds max|out| into (atomic maxima of the sign-stripped bf16
    patterns; they accumulate over launches) — the input of value_range_scale_from_amax."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        raise OccAmdUnsupported("conv3x3_nhwc: x must be a channels_last bfloat16 device tensor")
    _need_cuda_f32("bias", bias)
    N, Cin, H, W = x.shape
    if w_packed.numel() != cout * Cin * 9 or bias.numel() != cout:
        raise OccAmdError("conv3x3_nhwc: inconsistent shapes")
    st = int(stride)
    if st not in (1, 2):
        raise OccAmdUnsupported("conv3x3_nhwc: stride must be 1 or 2")
    if amax is not None and not (amax.is_cuda and amax.dtype == torch.int32 and amax.numel() == 8 and amax.is_contiguous()):
        raise OccAmdError("conv3x3_nhwc: amax must be 8 contiguous int32 device words")
    out = torch.empty((N, cout, (H - 1) // st + 1, (W - 1) // st + 1), dtype=torch.bfloat16, device=x.device,
                      memory_format=torch.channels_last)
    with torch.cuda.device(x.device), _timed('bb_conv3x3'):
        if amax is None:
            rc = _lib.lib().occ_conv3x3_nhwc_bf16(ptr(x), ptr(w_packed), ptr(bias), ptr(out), i32(N), i32(H),
                                                  i32(W), i32(Cin), i32(cout), i32(st), i32(1 if relu else 0),
                                                  stream_ptr(x.device))
        else:
            rc = _lib.lib().occ_conv3x3_nhwc_bf16_amax(ptr(x), ptr(w_packed), ptr(bias), ptr(out), i32(N), i32(H),
                                                       i32(W), i32(Cin), i32(cout), i32(st), i32(1 if relu else 0),
                                                       ptr(amax), stream_ptr(x.device))
    _note_flops('bb_conv3x3', 2.0 * N * out.shape[2] * out.shape[3] * 9 * Cin * cout)
    _lib.check(rc, "conv3x3_nhwc")
    return out
