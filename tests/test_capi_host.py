"""not-gpu: the C-ABI library loads and exports every symbol include/occnet_amd.h declares; argument
validation runs without a GPU (no compute call is made); the product refuses host tensors."""
import ctypes

import pytest
import torch

from occnet_amd import _lib, ext


def test_library_exports_declared_symbols():
    lib = _lib.lib()
    declared = _lib.declared_symbols()
    assert 'occ_ms_deform_attn_forward_f32' in declared and 'occ_sca_fused_forward_f32' in declared
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.occ_abi_version() == _lib.ABI == 3


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    rc = lib.occ_ms_deform_attn_forward_f32(null, null, null, null, null, null, 1, 1, 1, 1, 1, 1, 1,
                                            64, null)
    assert rc == -1 and b'null' in lib.occ_last_error()
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.occ_ms_deform_attn_forward_f32(p, p, p, p, p, p, 3, 1, 1, 1, 1, 1, 1, 2, null)
    assert rc == -1 and b'im2col_step' in lib.occ_last_error()     # batch 3 vs step 2
    rc = lib.occ_sca_fused_forward_f32(p, p, p, p, ctypes.c_int64(512), p, ctypes.c_int64(256), p, p,
                                       null, p, null, 1, 6, 100, 4, 64, 4, 8, 8, 10, null)
    assert rc == -3                                                # M=4,D=64: no fused kernel
    with pytest.raises(_lib.OccAmdUnsupported):
        _lib.check(rc, 'sca')
    rc = lib.occ_point_sampling_f32(p, p, p, p, ctypes.c_float(0.0), ctypes.c_float(1.0), p, p, null,
                                    1, 6, 4, 8, null)
    assert rc == -1


def test_backbone_and_projection_entry_points_validate_without_gpu():
    """The newer entry points reject bad arguments / unsupported shapes before any launch."""
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    i64 = ctypes.c_int64
    assert lib.occ_conv1x1_nhwc_bf16(null, p, p, null, p, 1, 4, 4, 32, 32, 1, 0, 0, null) == -1
    assert lib.occ_conv1x1_nhwc_bf16(p, p, p, null, p, 1, 4, 4, 48, 32, 1, 0, 0, null) == -3      # Cin % 32
    assert lib.occ_conv1x1_nhwc_bf16(p, p, p, null, p, 1, 4, 4, 32, 40, 1, 0, 0, null) == -3      # Cout % 32
    assert lib.occ_conv1x1_nhwc_bf16(p, p, p, p, p, 1, 5, 4, 32, 32, 1, 0, 1, null) == -1         # odd size, up2
    assert b'upsampled residual' in lib.occ_last_error()
    assert lib.occ_conv3x3_nhwc_bf16(p, p, p, p, 1, 4, 4, 32, 128, 3, 1, null) == -3              # stride 3
    assert lib.occ_conv3x3_nhwc_bf16(p, p, p, p, 1, 4, 4, 32, 64, 1, 1, null) == -3               # Cout % 128
    assert lib.occ_bottleneck64_nhwc_bf16(p, p, p, p, p, p, p, p, 1, 4, 4, 128, 0, null) == -3    # Cin 128
    assert lib.occ_bottleneck64_nhwc_bf16(p, p, p, p, p, p, p, p, 1, 4, 4, 64, 0, null) == -3     # 64 needs ds
    assert lib.occ_mfma_pack_b_frag_bf16(p, p, 48, 16, null) == -3                                # N % 32
    assert lib.occ_stem_conv7x7_pool_f32_bf16(p, p, p, null, 1, 8, 8, null) == -1
    assert lib.occ_bias_relu_maxpool_nhwc_bf16(p, p, p, 1, 4, 4, 12, null) == -3                  # C % 8
    assert lib.occ_conv3d_pack_weight_bf16x3(p, p, 24, 32, null) == -3                            # Cin % 16
    assert lib.occ_conv3d_bn_relu_bf16x3_f32(p, p, p, p, p, 1, 6, 4, 4, 32, 32, 0, i64(1), i64(1), i64(1), 1,
                                             null) == -3                                          # Z = 6
    # value projection: segment table checks
    ptrs = (ctypes.c_void_p * 1)(ctypes.addressof(buf))
    one = (ctypes.c_int64 * 1)(64)
    zero = (ctypes.c_int64 * 1)(0)
    args = lambda n, lda, K, N: (n, ptrs, lda, one, one, zero, null, 0, p, p, i64(256), K, N, i64(64), null)
    assert lib.occ_value_proj_bf16_f32(*args(0, one, 64, 256)) == -1                              # no segments
    assert lib.occ_value_proj_bf16_f32(*args(9, one, 64, 256)) == -1                              # > 8 segments
    assert lib.occ_value_proj_bf16_f32(*args(1, one, 48, 256)) == -3                              # K % 32
    assert lib.occ_value_proj_bf16_f32(*args(1, (ctypes.c_int64 * 1)(32), 64, 256)) == -1         # lda < K
    # stacked projections (round 3): plane geometry
    pargs = lambda n_planes, plane_cols, stride: (1, ptrs, one, one, one, zero, null, 0, p, p, 1, i64(256), 64, n_planes,
                                                  plane_cols, i64(stride), i64(64), null, null)
    assert lib.occ_value_proj_bf16_planes(*pargs(0, 256, 1 << 20)) == -1                          # no planes
    assert lib.occ_value_proj_bf16_planes(*pargs(4, 256, 0)) == -1                                # plane stride
    assert lib.occ_value_proj_bf16_planes(*pargs(4, 192, 1 << 20)) == -3                          # plane_cols % 256
    assert b'plane_cols' in lib.occ_last_error()
    # range scales of the fp16 value rows (round 5): argument checks
    f4 = (ctypes.c_float * 4)(1.0, 1.0, 1.0, 1.0)
    rargs = lambda n, lda, K, planes, l1=f4: (n, ptrs, lda, one, K, planes, l1, f4, p, p, null)
    assert lib.occ_value_range_scale_bf16(*rargs(0, one, 64, 4)) == -1                            # no segments
    assert lib.occ_value_range_scale_bf16(*rargs(1, one, 64, 9)) == -1                            # > 8 planes
    assert lib.occ_value_range_scale_bf16(*rargs(1, one, 60, 4)) == -3                            # K % 8
    assert lib.occ_value_range_scale_bf16(*rargs(1, (ctypes.c_int64 * 1)(32), 64, 4)) == -1       # lda < K
    # (round 6, abi 3: row_l1 / bias_max are DEVICE arrays — their sign cannot be checked on the host; the kernel takes |.|)
    assert lib.occ_value_range_scale_from_amax(null, 4, f4, f4, p, null) == -1                      # no maxima
    assert lib.occ_value_range_scale_from_amax(p, 9, f4, f4, p, null) == -1                         # > 8 planes
    assert lib.occ_conv3x3_nhwc_bf16_amax(p, p, p, p, 1, 4, 4, 32, 128, 1, 1, null, null) == -1     # null amax8
    assert lib.occ_value_range_scale_bf16(1, ptrs, one, one, 64, 4, f4, f4, null, p, null) == -1  # no output
    # fused second convolution + heads + decode (round 3)
    lib.occ_conv3d_heads_pack_bytes.restype = ctypes.c_int64
    assert lib.occ_conv3d_heads_pack_bytes() == 16 * 1024 + 16 * 1024 + 128 * 4 + 32 * 4
    hargs = lambda Z, Cin, ncls, occ=p: (p, p, p, p, p, occ, p, null, 1, Z, 4, 4, Cin, ncls, null)
    assert lib.occ_conv3d_heads_decode_bf16x3_f32(*hargs(16, 32, 17, null)) == -1                 # null output
    assert lib.occ_conv3d_heads_decode_bf16x3_f32(*hargs(8, 32, 17)) == -3                        # Z = 8 (16 and 32 have kernels)
    assert lib.occ_conv3d_heads_decode_bf16x3_f32(*hargs(16, 16, 17)) == -3                       # Cin = 16
    assert lib.occ_conv3d_heads_decode_bf16x3_f32(*hargs(16, 32, 40)) == -3                       # > 30 classes
    # row-local Linear chains (round 4): argument checks before any launch
    lib.occ_linear_chain_packed_bytes.restype = ctypes.c_int64
    assert lib.occ_linear_chain_packed_bytes(768, 256) == 768 * 256 * 4
    assert lib.occ_linear_chain_packed_bytes(192, 256) == 256 * 256 * 4                           # padded to a 256-row group
    assert lib.occ_linear_chain_pack_bf16x3(p, p, 256, 40, null) == -3                            # K % 16
    f = ctypes.c_float(1e-5)
    assert lib.occ_linear_ln_chain_bf16x3_f32(null, i64(256), p, i64(256), p, p, p, p, f, p, i64(256), p, i64(768),
                                              768, 0, 64, null) == -1                             # null input
    assert lib.occ_linear_ln_chain_bf16x3_f32(p, i64(256), p, i64(256), p, p, p, p, f, p, i64(256), p, i64(768),
                                              770, 0, 64, null) == -1                             # ldz < n2
    assert lib.occ_linear_ln_chain_bf16x3_f32(p, i64(256), p, i64(256), p, p, p, p, f, p, i64(256), p, i64(800),
                                              776, 0, 64, null) == -3                             # n2 % 32
    assert lib.occ_linear_pair_chain_bf16x3_f32(p, i64(256), p, p, null, i64(0), null, i64(192), 192, p, i64(256), 64,
                                                null) == -1                                       # null zq
    assert lib.occ_linear_pair_chain_bf16x3_f32(p, i64(256), p, p, null, i64(0), p, i64(192), 200, p, i64(256), 64,
                                                null) == -1                                       # ldzq < nq
    assert lib.occ_linear_pair_chain_bf16x3_f32(p, i64(256), p, p, null, i64(0), p, i64(192), 160, p, i64(256), 64,
                                                null) == -3                                       # nq % 64
    assert lib.occ_linear_ln_chain_bf16x3_f32(p, i64(256), p, i64(256), p, p, p, p, f, p, i64(256), p, i64(1536),
                                              1536, 0, 64, null) == -3                            # more tail columns than staged parameters
    assert lib.occ_encoder_ffn_chain_bf16x3_f32(p, i64(256), p, i64(256), p, p, p, p, f, p, p, f, p, i64(256), null,
                                                i64(0), p, i64(192), 192, null, i64(256), 64, null) == -1   # zq without zv
    assert lib.occ_encoder_ffn_chain_bf16x3_f32(p, i64(256), p, i64(256), p, p, p, p, f, p, p, f, p, i64(256), null,
                                                i64(0), p, i64(200), 200, p, i64(256), 64, null) == -3      # nq % 64
    with pytest.raises(_lib.OccAmdUnsupported):
        ext.conv1x1_pack_weight(torch.zeros(8, 32))
    with pytest.raises(_lib.OccAmdUnsupported):
        ext.stem_pack_weight(torch.zeros(64, 3, 3, 3))


def test_product_refuses_host_tensors():
    v = torch.zeros(1, 4, 8, 32)
    shapes = torch.tensor([[2, 2]])
    start = torch.tensor([0])
    loc = torch.zeros(1, 3, 8, 1, 2, 2)
    aw = torch.zeros(1, 3, 8, 1, 2)
    with pytest.raises(_lib.OccAmdError, match='no CPU fallback'):
        ext.ms_deform_attn_forward(v, shapes, start, loc, aw, im2col_step=64)
    from occnet_amd.plugin import build_head
    from tests.util import head_cfg, small_cfg
    from occnet_amd import synthetic
    g = small_cfg(bev=(4, 4), feat_shapes=((2, 2), (1, 1), (1, 1), (1, 1)), num_layers=1)
    head = build_head(head_cfg(g)).eval()
    with pytest.raises(_lib.OccAmdError, match='no CPU fallback'), torch.no_grad():
        head(synthetic.make_features(g), synthetic.make_img_metas(g))


def test_ext_loader_surface():
    from occnet_amd import ext_loader
    m = ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])
    assert callable(m.ms_deform_attn_forward) and callable(m.ms_deform_attn_backward)
    with pytest.raises(AssertionError):
        ext_loader.load_ext('_ext', ['no_such_op'])
    from occnet_amd.plugin import (MultiScaleDeformableAttnFunction_fp16,
                                   MultiScaleDeformableAttnFunction_fp32)
    assert hasattr(MultiScaleDeformableAttnFunction_fp32, 'apply')
    assert hasattr(MultiScaleDeformableAttnFunction_fp16, 'apply')


def test_training_entry_points_validate_without_gpu():
    """Round-2 training entry points (Linear weight gradient, row gather-sum, SCA query-side preparation, backward
    with caller-provided scratch): argument checks and workspace sizing run before any launch."""
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    i64 = ctypes.c_int64
    lib.occ_linear_wgrad_workspace_bytes.restype = ctypes.c_int64
    lib.occ_ms_deform_attn_backward_workspace_bytes.restype = ctypes.c_int64
    # one partial (N*K + N floats) per row chunk; 512 blocks when the rows allow it
    ws = lib.occ_linear_wgrad_workspace_bytes(40000, 512, 256)
    assert ws > 0 and ws % ((512 * 256 + 512) * 4) == 0 and ws // ((512 * 256 + 512) * 4) == 63    # 640-row chunks
    assert lib.occ_linear_wgrad_workspace_bytes(10, 512, 256) == (512 * 256 + 512) * 4           # one chunk
    assert lib.occ_linear_wgrad_workspace_bytes(0, 512, 256) == 0
    assert lib.occ_linear_wgrad_bf16x3_f32(null, i64(8), p, i64(8), p, p, p, i64(1 << 20), 4, 8, 8, null) == -1
    assert lib.occ_linear_wgrad_bf16x3_f32(p, i64(4), p, i64(8), p, p, p, i64(1 << 20), 4, 8, 8, null) == -1
    assert b'row strides' in lib.occ_last_error()
    assert lib.occ_linear_wgrad_bf16x3_f32(p, i64(8), p, i64(8), p, p, p, i64(16), 4, 8, 8, null) == -1
    assert b'workspace too small' in lib.occ_last_error()
    idx = (ctypes.c_int64 * 4)(0, 1, -1, 2)
    ip = ctypes.cast(idx, ctypes.c_void_p)
    assert lib.occ_rows_gather_sum_f32(null, i64(16), ip, 1, p, 1, i64(4), i64(4), 4, null) == -1
    assert lib.occ_rows_gather_sum_f32(p, i64(16), ip, 1, p, 1, i64(4), i64(4), 6, null) == -1   # F % 4
    assert b'16-byte aligned' in lib.occ_last_error()
    assert lib.occ_sca_prep_forward_f32(p, i64(768), 768, ip, p, ip, p, p, 1, i64(4), 4, 4, 8, 4, null) == -3   # M=4
    assert lib.occ_sca_prep_forward_f32(p, i64(768), 768, ip, p, ip, p, p, 1, i64(4), 8, 4, 8, 3, null) == -1   # P % Z
    assert lib.occ_sca_prep_forward_f32(p, i64(512), 512, ip, p, ip, p, p, 1, i64(4), 8, 4, 8, 4, null) == -1   # short rows
    assert lib.occ_sca_prep_backward_f32(p, p, p, ip, 2, ip, p, 768, 1, i64(4), i64(2), 8, 2, 8, null) == -3
    # backward scratch: flags + counters + work list + 4 * samples row items of 12 bytes + (deterministic mode) one tile of
    # 32 x 32 + 32 64-bit words per possible split bin (more than 2 048 items); 0 for other head sizes
    B, S, M, D, L, Lq, P = 6, 30825, 8, 32, 4, 9900, 8
    need = lib.occ_ms_deform_attn_backward_workspace_bytes(B, S, M, D, L, Lq, P)
    items = 4 * (B * Lq * M * L * P)
    tiles = (items // 2048 + 1) * (32 * 32 + 32) * 8
    assert need > items * 12 + tiles and need < 1.1 * items * 12 + tiles + (64 << 20)
    assert lib.occ_ms_deform_attn_backward_workspace_bytes(B, S, M, 64, L, Lq, P) == 0
    rc = lib.occ_ms_deform_attn_backward_ws_f32(p, p, p, p, p, p, p, p, p, 1, 4, 8, 32, 1, 4, 4, 64, p, i64(16), null)
    assert rc == -1 and b'workspace too small' in lib.occ_last_error()


def test_round6_training_nodes_validate_without_gpu():
    """Round-6 training entry points (masked gradient + bias gradient pass, BatchNorm fold, dropout + residual + LayerNorm
    node): scratch sizing, argument checks before any launch, and the keep-mask hash — a HOST function of the library — against
    its numpy restatement (the restatement the GPU test derives its reference mask from)."""
    import numpy as np
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    i64, f32 = ctypes.c_int64, ctypes.c_float
    lib.occ_bias_act_bwd_partial_floats.restype = ctypes.c_int64
    lib.occ_dropout_add_ln_bwd_partial_floats.restype = ctypes.c_int64
    lib.occ_ln_dropout_hash.restype = ctypes.c_uint32
    # bias_act backward: <= 512 blocks of 256 / (C / 8) rows each, one row of C partial sums per block; C % 8 == 0, C <= 2048
    assert lib.occ_bias_act_bwd_partial_floats(i64(6 * 116 * 200), 512) == 512 * 512
    assert lib.occ_bias_act_bwd_partial_floats(i64(7), 64) == 1 * 64
    assert lib.occ_bias_act_bwd_partial_floats(i64(100), 12) == 0 and lib.occ_bias_act_bwd_partial_floats(i64(100), 4096) == 0
    assert lib.occ_bias_act_bwd_nhwc_bf16(null, p, p, p, p, i64(8), 64, 1, null) == -1
    assert lib.occ_bias_act_bwd_nhwc_bf16(p, null, null, p, p, i64(8), 64, 1, null) == -1        # relu needs y and g
    assert lib.occ_bias_act_bwd_nhwc_bf16(p, null, null, p, p, i64(8), 12, 0, null) == -3        # C % 8
    assert lib.occ_conv_bn_fold_fwd_f32(null, p, p, p, p, p, p, p, 8, 8, 1, 1, null) == -1
    assert lib.occ_conv_bn_fold_bwd_f32(p, i64(8), i64(1), i64(1), i64(1), p, p, p, p, p, p, p, 0, 8, 1, 1, null) == -1
    # dropout + residual + LayerNorm: C = 256 only; <= 1024 blocks of 4 rows, 2 C partial sums per block
    assert lib.occ_dropout_add_ln_bwd_partial_floats(i64(40000), 256) == 1024 * 512
    assert lib.occ_dropout_add_ln_bwd_partial_floats(i64(5), 256) == 2 * 512
    assert lib.occ_dropout_add_ln_bwd_partial_floats(i64(5), 128) == 0
    u64 = ctypes.c_uint64
    assert lib.occ_dropout_add_ln_fwd_f32(null, p, p, p, f32(1e-5), f32(0.1), u64(1), p, p, p, i64(4), 256, null) == -1
    assert lib.occ_dropout_add_ln_fwd_f32(p, p, p, p, f32(1e-5), f32(1.0), u64(1), p, p, p, i64(4), 256, null) == -1      # p < 1
    assert lib.occ_dropout_add_ln_fwd_f32(p, p, p, p, f32(1e-5), f32(0.1), u64(1), p, p, p, i64(4), 128, null) == -3
    assert b'C = 256' in lib.occ_last_error()
    assert lib.occ_dropout_add_ln_bwd_f32(p, p, p, p, f32(0.1), u64(1), null, p, p, p, i64(4), 256, null) == -1            # p > 0 needs grad_x
    # the keep-mask hash
    seed = 0x1234567_89abcdef
    idx = np.concatenate([np.arange(4096, dtype=np.uint32), np.array([2 ** 31, 2 ** 32 - 1], dtype=np.uint32)])
    s0, s1 = np.uint32(seed & 0xffffffff), np.uint32(seed >> 32)
    with np.errstate(over='ignore'):
        h = idx ^ s0
        h ^= h >> np.uint32(16); h *= np.uint32(0x85ebca6b); h ^= h >> np.uint32(13); h *= np.uint32(0xc2b2ae35); h ^= h >> np.uint32(16)
        h += s1
        h ^= h >> np.uint32(15); h *= np.uint32(0x2c1b3c6d); h ^= h >> np.uint32(12)
    got = np.array([lib.occ_ln_dropout_hash(ctypes.c_uint32(int(i)), ctypes.c_uint32(int(s0)), ctypes.c_uint32(int(s1))) for i in idx],
                   dtype=np.uint32)
    assert np.array_equal(got, h)
    keep = (h[:4096] >= np.uint32(int(0.1 * 4294967296.0))).mean()
    assert abs(keep - 0.9) < 0.03
