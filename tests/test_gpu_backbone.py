"""-m gpu: the inference execution plan of the stock ResNet-50 + FPN backbone (eval BatchNorm folded into the
convolutions, NHWC bf16, one HIP bias/residual/ReLU launch per convolution) vs the fp32 modules."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bias_act_nhwc_matches_torch():
    from occnet_amd import ext
    g = torch.Generator().manual_seed(0)
    for (N, C, H, W) in [(2, 64, 7, 9), (1, 256, 5, 3), (3, 8, 1, 1)]:
        x = torch.randn(N, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
            memory_format=torch.channels_last)
        r = torch.randn(N, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
            memory_format=torch.channels_last)
        b = torch.randn(C, generator=g).cuda()
        for res, relu in ((None, True), (r, True), (None, False)):
            want = x.float() + b.view(1, -1, 1, 1)
            if res is not None:
                want = want + res.float()
            if relu:
                want = want.relu()
            got = ext.bias_act_nhwc_(x.clone(memory_format=torch.channels_last), b, residual=res, relu=relu)
            assert torch.equal(got, want.to(torch.bfloat16))        # fp32 math, one RNE rounding: exact


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride,res,relu", [
    (2, 64, 256, 9, 13, 1, True, True), (1, 256, 64, 7, 5, 1, False, True), (3, 512, 1024, 6, 7, 2, False, False),
    (1, 2048, 512, 3, 4, 1, False, True), (2, 32, 32, 5, 5, 1, True, False), (1, 1024, 2048, 5, 8, 2, False, False),
    # K chunk counts that are not multiples of the unroll, column tiles that do not fill the block
    (2, 96, 160, 5, 5, 1, True, True), (1, 224, 96, 4, 4, 1, False, False), (1, 160, 416, 3, 3, 1, True, True)])
def test_conv1x1_nhwc_matches_torch(N, Cin, Cout, H, W, stride, res, relu):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(Cin + Cout)
    cl = lambda t: t.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(N, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(Cout, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = cl(torch.randn(N, Cout, Ho, Wo, generator=g)) if res else None
    got = ext.conv1x1_nhwc(x, ext.conv1x1_pack_weight(w), b, residual=r, relu=relu, stride=stride)
    want = torch.nn.functional.conv2d(x.float(), w.float().view(Cout, Cin, 1, 1), b, stride=stride)
    if res:
        want = want + r.float()
    if relu:
        want = want.relu()
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    d = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    print(f"conv1x1 {Cin}->{Cout} s{stride}: max diff {d:.3e} (scale {scale:.2f})")
    assert d <= scale * 2 ** -8 + 1e-6          # one bf16 rounding of the f32-accumulated result


def test_conv1x1_upsampled_residual_matches_torch():
    """FPN top-down step in one launch: lateral 1x1 conv + nearest x2 upsampling of the coarser lateral."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(5)
    cl = lambda t: t.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for (N, Cin, Cout, H, W) in [(2, 64, 256, 6, 10), (1, 512, 256, 58, 100)]:
        x = cl(torch.randn(N, Cin, H, W, generator=g))
        r = cl(torch.randn(N, Cout, H // 2, W // 2, generator=g))
        w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).cuda().to(torch.bfloat16)
        b = torch.randn(Cout, generator=g).cuda()
        got = ext.conv1x1_nhwc(x, ext.conv1x1_pack_weight(w), b, residual=r, relu=False, residual_upsample2=True)
        F = torch.nn.functional
        want = F.conv2d(x.float(), w.float().view(Cout, Cin, 1, 1), b) + F.interpolate(r.float(), size=(H, W))
        d, scale = float((got.float() - want).abs().max()), float(want.abs().max())
        print(f"conv1x1+up2 {Cin}->{Cout} {H}x{W}: max diff {d:.3e} (scale {scale:.2f})")
        assert d <= scale * 2 ** -8 + 1e-6
    with pytest.raises(ext.OccAmdUnsupported):
        ext.conv1x1_nhwc(cl(torch.randn(1, 32, 5, 6)), ext.conv1x1_pack_weight(torch.randn(32, 32).cuda()),
                         torch.zeros(32).cuda(), residual=cl(torch.randn(1, 32, 2, 3)), residual_upsample2=True)


@pytest.mark.parametrize("N,C,Cout,H,W,relu,stride", [
    (2, 128, 128, 9, 21, True, 1), (1, 256, 256, 16, 32, False, 1), (1, 512, 512, 5, 7, True, 1),
    (3, 64, 256, 8, 16, True, 1), (1, 32, 128, 1, 1, False, 1), (1, 256, 128, 29, 50, True, 1),
    # stride 2 (first blocks of layer2..4, FPN extra level): even / odd sizes, ragged tiles, tiny maps
    (2, 128, 128, 16, 64, True, 2), (1, 256, 256, 29, 50, True, 2), (1, 512, 512, 7, 9, False, 2),
    (2, 64, 128, 1, 1, True, 2), (1, 128, 256, 58, 100, True, 2), (1, 32, 128, 2, 3, False, 2)])
def test_conv3x3_nhwc_matches_torch(N, C, Cout, H, W, relu, stride):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(N * 1000 + C + H * W + stride)
    x = torch.randn(N, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    w_bf = w.to(torch.bfloat16).float()                       # the kernel rounds the weights to bf16
    b = torch.randn(Cout, generator=g).cuda()
    got = ext.conv3x3_nhwc(x, ext.conv3x3_pack_weight(w), b, Cout, relu=relu, stride=stride)
    want = torch.nn.functional.conv2d(x.float(), w_bf, b, padding=1, stride=stride)
    if relu:
        want = want.relu()
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    d = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    print(f"conv3x3 {C}->{Cout} {H}x{W} s{stride}: max diff {d:.3e} (scale {scale:.2f})")
    assert d <= scale * 2 ** -8 + 1e-5          # one bf16 rounding of the f32-accumulated result


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 53), (1, 928 // 8, 1600 // 8), (1, 9, 7)])
def test_stem_conv7x7_pool_matches_torch(N, H, W):
    """Whole stem in one kernel vs torch on the same bf16-rounded operands (fp32 accumulation, one rounding of the
    convolution output)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(H * W)
    x = (torch.randn(N, 3, H, W, generator=g) * 50.0).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12.0).cuda()
    b = torch.randn(64, generator=g).cuda()
    r = lambda t: t.to(torch.bfloat16).float()
    got = ext.stem_conv7x7_pool(x, ext.stem_pack_weight(w), b)
    F = torch.nn.functional
    conv = r(F.conv2d(r(x), r(w), b, stride=2, padding=3).relu())
    want = F.max_pool2d(conv, 3, 2, 1)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    d = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    print(f"stem {N}x3x{H}x{W}: max diff {d:.3e} (scale {scale:.1f})")
    assert d <= scale * 2 ** -8 + 1e-5          # a bf16 rounding flip of the f32-accumulated value
    assert float((got.float() - want).abs().mean()) <= scale * 2 ** -13


def test_bias_relu_maxpool_matches_torch():
    from occnet_amd import ext
    g = torch.Generator().manual_seed(3)
    for (N, C, H, W) in [(2, 64, 12, 20), (1, 8, 7, 9), (1, 16, 1, 1), (3, 64, 5, 2)]:
        y = torch.randn(N, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
            memory_format=torch.channels_last)
        b = torch.randn(C, generator=g).cuda()
        got = ext.bias_relu_maxpool_nhwc(y, b)
        want = torch.nn.functional.max_pool2d((y.float() + b.view(1, -1, 1, 1)).relu(), 3, 2, 1)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, want.to(torch.bfloat16))         # one RNE rounding of the same fp32 value


def _bottleneck_reference(x, w1, b1, w2, b2, w3, b3, wds=None, bds=None):
    """fp32 torch restatement of what the fused kernel computes: bf16 weights, f32 accumulation, the two
    64-channel intermediates rounded to bf16 (they are bf16 tensors in the unfused plan too)."""
    r = lambda t: t.to(torch.bfloat16).float()
    F = torch.nn.functional
    xf = x.float()
    c1 = r(F.conv2d(xf, r(w1), b1).relu())
    c2 = r(F.conv2d(c1, r(w2), b2, padding=1).relu())
    idn = xf if wds is None else F.conv2d(xf, r(wds), bds)
    return (F.conv2d(c2, r(w3), b3) + idn).relu()


@pytest.mark.parametrize("cin,ds,N,H,W", [
    (256, False, 2, 24, 48),     # whole tiles
    (256, False, 1, 13, 21),     # ragged right / bottom tiles
    (64, True, 2, 16, 32),       # first block: projection shortcut folded into the third GEMM
    (64, True, 1, 9, 17),
    (256, False, 1, 3, 5),       # smaller than one tile
])
def test_bottleneck64_fused_matches_torch(cin, ds, N, H, W):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(cin + H * W)
    x = torch.randn(N, cin, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    w1 = (torch.randn(64, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda()
    w3 = (torch.randn(256, 64, 1, 1, generator=g) / 8.0).cuda()
    b1, b2, b3 = (torch.randn(n, generator=g).cuda() * 0.3 for n in (64, 64, 256))
    wds = (torch.randn(256, cin, 1, 1, generator=g) / cin ** 0.5).cuda() if ds else None
    bds = torch.randn(256, generator=g).cuda() * 0.3 if ds else None
    pack = ext.bottleneck64_pack(w1, b1, w2, b2, w3, b3, wds, bds)
    got = ext.bottleneck64_nhwc(x, pack)
    want = _bottleneck_reference(x, w1, b1, w2, b2, w3, b3, wds, bds)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    d = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    print(f"bottleneck64 cin={cin} ds={ds} {N}x{H}x{W}: max diff {d:.3e} (scale {scale:.2f})")
    # one bf16 rounding of the output + rounding flips of the bf16 intermediates (f32 summation order differs)
    assert d <= scale * 2 ** -7 + 1e-5
    assert float((got.float() - want).abs().mean()) <= scale * 2 ** -11


def test_bottleneck64_rejects_other_shapes():
    from occnet_amd import ext
    with pytest.raises(ext.OccAmdUnsupported):
        ext.bottleneck64_pack(torch.zeros(64, 128, 1, 1).cuda(), torch.zeros(64).cuda(),
                              torch.zeros(64, 64, 3, 3).cuda(), torch.zeros(64).cuda(),
                              torch.zeros(256, 64, 1, 1).cuda(), torch.zeros(256).cuda())


def test_fused_bottleneck_plan_matches_layerwise_plan():
    """The plan with the whole-bottleneck kernel against the same plan built layer by layer."""
    from occnet_amd.plugin.backbone import FPN, FusedInferenceBackbone, ResNet
    torch.manual_seed(0)
    bb = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_eval=True).eval()
    bb.init_weights()
    nk = FPN(in_channels=[512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_output',
             num_outs=4, relu_before_extra_convs=True).eval()
    bb, nk = bb.cuda(), nk.cuda()
    x = torch.randn(2, 3, 96, 160).cuda() * 50.0
    with torch.no_grad():
        a = FusedInferenceBackbone(bb, nk, fused_bottleneck=True)
        b = FusedInferenceBackbone(bb, nk, fused_bottleneck=False)
        assert len(a._bneck) == 3 and not b._bneck
        for u, v in zip(a(x), b(x)):
            rel = float((u.float() - v.float()).abs().max() / v.float().abs().max())
            print(f"level {tuple(u.shape)}: fused vs layerwise max rel diff {rel:.3e}")
            assert rel < 0.03


def test_folded_plan_matches_fp32_modules():
    from occnet_amd.plugin.backbone import FPN, FusedInferenceBackbone, ResNet
    torch.manual_seed(0)
    bb = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_eval=True).eval()
    bb.init_weights()
    g = torch.Generator().manual_seed(1)
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    nk = FPN(in_channels=[512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_output',
             num_outs=4, relu_before_extra_convs=True).eval()
    bb, nk = bb.cuda(), nk.cuda()
    x = torch.randn(2, 3, 96, 160, generator=g).cuda() * 50.0
    with torch.no_grad():
        ref = nk(bb(x))
        for hip_tail in (True, False):
            plan = FusedInferenceBackbone(bb, nk, dtype=torch.bfloat16, hip_tail=hip_tail)
            out = plan(x)
            for a, b in zip(ref, out):
                assert b.dtype == torch.bfloat16 and b.shape == a.shape
                assert b.is_contiguous(memory_format=torch.channels_last)
                rel = float((a - b.float()).abs().max() / a.abs().max())
                print(f"hip_tail={hip_tail} level {tuple(a.shape)}: max rel diff {rel:.3e}")
                assert rel < 0.06       # bf16 through 53 convolutions


@pytest.mark.parametrize("N,Hs,Ws,to_rgb,std", [(2, 37, 50, False, (1.0, 1.0, 1.0)),
                                                (1, 64, 96, True, (58.395, 57.12, 57.375)),
                                                (3, 900 // 10, 1600 // 10, False, (1.0, 1.0, 1.0))])
def test_stem_u8_input_equals_normalise_pad_then_stem(N, Hs, Ws, to_rgb, std):
    """SURVEY.md §8f N4, device side: the stem fed with raw uint8 HWC camera images (NormalizeMultiviewImage +
    PadMultiViewImage fused into the tile staging; reference transform_3d.py:31-45,82-94) is BIT-IDENTICAL to the
    stem run on the float tensor the host pipeline (occnet_amd/io.py = the pipeline's numpy restatement) builds."""
    import numpy as np
    from occnet_amd import ext, io
    rng = np.random.default_rng(N * 1000 + Hs)
    raw = [rng.integers(0, 256, (Hs, Ws, 3), dtype=np.uint8) for _ in range(N)]
    mean = (103.530, 116.280, 123.675)
    normed, _ = io.normalize_multiview(raw, mean, std, to_rgb=to_rgb)
    padded, meta = io.pad_multiview(normed, size_divisor=32)
    x = io.to_batch(padded)[0].cuda().contiguous()                      # (N, 3, H, W) fp32
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).cuda()
    b = torch.randn(64, generator=g).cuda()
    frag = ext.stem_pack_weight(w)
    want = ext.stem_conv7x7_pool(x, frag, b)
    got, hw = ext.stem_conv7x7_pool_u8(torch.from_numpy(np.stack(raw)).cuda(), frag, b, mean, std, to_rgb=to_rgb)
    assert hw == tuple(x.shape[2:]) and got.shape == want.shape
    assert torch.equal(got, want)


def test_detector_u8_path_equals_float_path():
    """BEVFormerOcc.extract_feat_u8 (raw images -> FPN maps on the inference plan) == extract_feat on the
    host-normalised / padded float tensor."""
    import numpy as np
    import os
    from occnet_amd import io
    from occnet_amd.plugin import Config, build_model, import_plugin
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'occ_base_200x200x16.py'))
    import_plugin(cfg)
    torch.manual_seed(0)
    model = build_model(cfg.model)
    model.init_weights()
    model = model.cuda().eval()
    model.enable_fused_backbone(dtype=torch.bfloat16)
    rng = np.random.default_rng(5)
    raw = [rng.integers(0, 256, (90, 160, 3), dtype=np.uint8) for _ in range(6)]
    ncfg = dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False)
    normed, _ = io.normalize_multiview(raw, **ncfg)
    padded, _ = io.pad_multiview(normed, size_divisor=32)
    with torch.no_grad():
        want = model.extract_feat(img=io.to_batch(padded).cuda())
        got, hw = model.extract_feat_u8(torch.from_numpy(np.stack(raw))[None].cuda(), ncfg)
    assert hw == (96, 160)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)
    # without the plan: torch device ops for normalise / pad, stock modules -> same maps up to bf16-vs-fp32 backbone
    model.enable_fused_backbone(dtype=None)
    with torch.no_grad():
        ref = model.extract_feat(img=io.to_batch(padded).cuda())
        alt, _ = model.extract_feat_u8(torch.from_numpy(np.stack(raw))[None].cuda(), ncfg)
    for a, b in zip(alt, ref):       # same fp32 modules; MIOpen may pick another algorithm for the other memory layout
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


def _train_backbone():
    from occnet_amd.plugin.backbone import ResNet
    torch.manual_seed(0)
    bb = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_eval=True)
    bb.init_weights()
    g = torch.Generator().manual_seed(1)
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return bb.cuda().train(), g


def test_training_backbone_folded_bn_matches_batchnorm_modules_fp32():
    """norm_eval training: conv+BN as one folded convolution (conv_bn_folded) vs the conv -> BatchNorm2d modules,
    fp32, outputs and parameter gradients (same function, different association: 1e-4 relative)."""
    from occnet_amd.plugin.backbone import Bottleneck
    bb, g = _train_backbone()
    x = (torch.randn(2, 3, 64, 96, generator=g) * 50.0).cuda()
    res = {}
    for fold in (True, False):
        Bottleneck.fold_eval_bn = fold
        try:
            bb.zero_grad(set_to_none=True)
            outs = bb(x)
            sum((o.float() ** 2).mean() for o in outs).backward()
            res[fold] = ([o.detach().clone() for o in outs],
                         {n: p.grad.detach().clone() for n, p in bb.named_parameters() if p.grad is not None})
        finally:
            Bottleneck.fold_eval_bn = True
    assert res[True][1].keys() == res[False][1].keys() and len(res[True][1]) > 100
    assert not any(n.startswith(('conv1.', 'bn1.', 'layer1.')) for n in res[True][1])       # frozen_stages=1
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-4
    rel = sorted(float((res[True][1][n] - gb).abs().max() / (gb.abs().max() + 1e-12)) for n, gb in res[False][1].items())
    worst, median = rel[-1], rel[len(rel) // 2]
    print(f"folded-BN training backbone: relative gradient difference worst {worst:.2e}, median {median:.2e}")
    # both sides run MIOpen's fp32 convolutions, whose solver (and summation order) is picked per shape and per run; the
    # BatchNorm-parameter gradients are differences of large terms (inputs scaled x50), so the worst tensor moves between
    # 1e-3 and 1e-2 from box to box.  A wrong gradient is an O(1) difference: the bound on the worst tensor stays far
    # below that, the median pins the typical agreement.
    assert worst < 5e-2 and median < 1e-3


def test_training_backbone_frozen_prefix_runs_on_the_plan_kernels():
    """frozen_stages=1 under bf16 autocast: stem + layer1 on the own kernels (no autograd graph needed) vs the
    torch modules; the trainable stages see the same activations up to bf16 rounding and still get gradients."""
    from occnet_amd.plugin.backbone import ResNet
    bb, g = _train_backbone()
    x = (torch.randn(2, 3, 96, 160, generator=g) * 50.0).cuda()
    res = {}
    for use in (True, False):
        ResNet.use_frozen_prefix_plan = use
        try:
            bb.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                outs = bb(x)
            sum((o.float() ** 2).mean() for o in outs).backward()
            res[use] = ([o.detach().float() for o in outs], bb.layer2[0].conv1.weight.grad.detach().clone())
        finally:
            ResNet.use_frozen_prefix_plan = True
    assert getattr(bb, '_prefix_plan', None) is not None and len(bb._prefix_plan.stages) == 1
    for a, b in zip(res[True][0], res[False][0]):
        rel = float((a - b).abs().max() / b.abs().max())
        print(f"frozen prefix on plan kernels, level {tuple(a.shape)}: max rel diff {rel:.3e}")
        assert rel < 0.06
    gr = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
    assert gr < 0.1


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 7, 9), (1, 256, 5, 3), (3, 8, 1, 1), (2, 192, 6, 5), (1, 2048, 3, 4), (6, 512, 29, 50)])
def test_bias_act_bwd_nhwc_matches_torch(N, C, H, W):
    """g = grad_y * (y > 0) bit for bit; the bias gradient = fp32 column sums of g (a fixed summation order: two runs agree
    bit for bit; vs torch's fp64 sum to fp32 accumulation noise)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(C + H)
    cl = lambda t: t.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = cl(torch.randn(N, C, H, W, generator=g))
    yf = torch.randn(N, C, H, W, generator=g).relu()
    yf.view(-1)[::7] = -0.0           # a negative zero is not > 0
    y = cl(yf)
    for relu in (True, False):
        got_g, got_b = ext.bias_act_bwd_nhwc(gy, y if relu else None, relu=relu)
        again_g, again_b = ext.bias_act_bwd_nhwc(gy, y if relu else None, relu=relu)
        want_g = torch.where(y.float() > 0, gy, torch.zeros_like(gy)) if relu else gy
        assert torch.equal(got_g, want_g) and torch.equal(got_b, again_b)
        want_b = want_g.double().sum((0, 2, 3))
        assert float((got_b.double() - want_b).abs().max()) <= 1e-5 * max(1.0, float(want_b.abs().max())) * (N * H * W) ** 0.5


@pytest.mark.parametrize("cin,cout,k,stride,bn,res,relu", [
    (256, 128, 1, 1, True, False, True),      # bottleneck conv1: own 1x1 kernel
    (128, 128, 3, 2, True, False, True),      # conv2, stride 2: own 3x3 kernel
    (128, 512, 1, 1, True, True, True),       # conv3 + identity + ReLU: own 1x1 kernel with residual
    (256, 512, 1, 2, True, False, False),     # downsample projection
    (64, 64, 3, 1, True, False, True),        # a shape without an own kernel: MIOpen + the fused tail
    (512, 256, 1, 1, False, False, False),    # FPN lateral: bias, no norm
    (256, 256, 3, 1, False, False, False)])   # FPN output convolution
def test_conv_bn_act_function_matches_the_autocast_chain(cin, cout, k, stride, bn, res, relu):
    """ConvBNActFunction (one autograd node: fold + own / MIOpen convolution with fused tail; backward = one masked-gradient +
    bias-gradient pass, MIOpen's data / weight gradients, the fold's chain rule) against the chain it replaces under
    torch.autocast(bf16): conv_bn_folded (or the biased convolution) -> + residual -> ReLU.  Outputs and every gradient agree to
    bf16 rounding (both sides round the activations to bf16; the fused node rounds once where the chain rounds three times)."""
    from occnet_amd.plugin.backbone import conv_bn_act, conv_bn_folded
    g = torch.Generator().manual_seed(cin + cout + k)
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=not bn).cuda()
    norm = None
    if bn:
        norm = torch.nn.BatchNorm2d(cout).cuda().eval()
        with torch.no_grad():
            norm.running_mean.copy_(torch.randn(cout, generator=g).cuda() * 0.1)
            norm.running_var.copy_(torch.rand(cout, generator=g).cuda() * 0.5 + 0.75)
            norm.weight.copy_(torch.rand(cout, generator=g).cuda() * 0.5 + 0.75)
            norm.bias.copy_(torch.randn(cout, generator=g).cuda() * 0.1)
    cl = lambda t: t.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    N, H, W = 2, 12, 20
    x0 = cl(torch.randn(N, cin, H, W, generator=g))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r0 = cl(torch.randn(N, cout, Ho, Wo, generator=g)) if res else None
    gy = cl(torch.randn(N, cout, Ho, Wo, generator=g))
    params = [conv.weight] + ([conv.bias] if conv.bias is not None else []) + ([norm.weight, norm.bias] if bn else [])
    out = {}
    for mode in ("fused", "chain", "fp32"):
        x = (x0.float() if mode == "fp32" else x0.clone(memory_format=torch.channels_last)).requires_grad_(True)
        r = None if r0 is None else (r0.float() if mode == "fp32" else r0.clone(memory_format=torch.channels_last)).requires_grad_(True)
        for p in params:
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=mode != "fp32"):
            if mode == "fused":
                y = conv_bn_act(x, conv, norm, relu=relu, residual=r)
            else:
                y = conv_bn_folded(x, conv, norm) if bn else conv(x)
                if r is not None:
                    y = y + r
                if relu:
                    y = torch.relu(y)
        assert y.dtype == (torch.float32 if mode == "fp32" else torch.bfloat16)
        y.backward(gy.float() if mode == "fp32" else gy)
        out[mode] = [y.detach().float(), x.grad.float()] + ([] if r is None else [r.grad.float()]) + [p.grad.float().clone() for p in params]
    # the yardstick is the fp32 chain on the same (bf16-valued) inputs: where y lands within a bf16 rounding of zero the ReLU
    # mask of a bf16 path may flip and move whole terms of the gradients, so the two bf16 paths differ from EACH OTHER by
    # several per cent in L2 — but the fused node (fp32 tail, one rounding) must sit as close to fp32 as the chain it replaces
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    for i, (f, c, ref) in enumerate(zip(out["fused"], out["chain"], out["fp32"])):
        ef, ec = rel(f, ref), rel(c, ref)
        assert f.shape == ref.shape and ef <= max(1.5 * ec, 1e-2), (i, tuple(ref.shape), ef, ec)
        assert ef < 8e-2, (i, ef)


def test_training_backbone_fused_nodes_match_the_autocast_modules():
    """ResNet-50 + FPN under bf16 autocast, training mode (norm_eval, frozen_stages=1): with the fused autograd nodes
    (ConvBNActFunction, default) against the module graph they replace (OCC_TRAIN_FUSED_CONV=0's path) — the four FPN maps and
    the parameter gradients agree to bf16 noise, every trainable parameter receives a gradient on both sides."""
    from occnet_amd.plugin.backbone import FPN, Bottleneck
    bb, g = _train_backbone()
    neck = FPN(in_channels=[512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_output', num_outs=4,
               relu_before_extra_convs=True).cuda().train()
    x = (torch.randn(2, 3, 96, 160, generator=g) * 50.0).cuda()
    res = {}
    for fused in (True, False):
        Bottleneck.fused_train_nodes = fused
        try:
            bb.zero_grad(set_to_none=True)
            neck.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                outs = neck(bb(x))
            assert len(outs) == 4
            sum((o.float() ** 2).mean() for o in outs).backward()
            grads = {n: p.grad.detach().float().clone() for m, pre in ((bb, 'bb.'), (neck, 'neck.'))
                     for n, p in ((pre + k, v) for k, v in m.named_parameters()) if p.grad is not None}
            res[fused] = ([o.detach().float() for o in outs], grads)
        finally:
            Bottleneck.fused_train_nodes = True
    assert res[True][1].keys() == res[False][1].keys() and len(res[True][1]) > 120
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).abs().max() / b.abs().max()) < 0.06
    rel = sorted(float((res[True][1][n] - gb).abs().max() / (gb.abs().max() + 1e-12)) for n, gb in res[False][1].items())
    print(f"fused training nodes: relative gradient difference worst {rel[-1]:.2e}, median {rel[len(rel) // 2]:.2e}")
    assert rel[len(rel) // 2] < 0.05 and rel[-1] < 0.5


@pytest.mark.parametrize("O,I,k", [(128, 256, 1), (128, 128, 3), (2048, 512, 1), (64, 3, 7)])
def test_conv_bn_fold_kernels_match_torch(O, I, k):
    """The one-launch fold and its chain rule against the ATen expressions of conv_bn_folded: w_folded bit for bit (one fp32
    product), its bf16 channels_last copy = RNE of it, the bias one fma; grad_weight bit for bit, grad_gamma a fixed-order sum."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(O + I + k)
    W = torch.randn(O, I, k, k, generator=g).cuda()
    gamma, beta = (torch.rand(O, generator=g) + 0.5).cuda(), torch.randn(O, generator=g).cuda()
    rstd, mean_rstd = (torch.rand(O, generator=g) + 0.5).cuda(), torch.randn(O, generator=g).cuda()
    wf, w16, b = ext.conv_bn_fold_fwd(W, gamma, beta, rstd, mean_rstd)
    s = gamma * rstd
    assert torch.equal(wf, W * s.view(-1, 1, 1, 1))
    assert w16.is_contiguous(memory_format=torch.channels_last) and torch.equal(w16, wf.to(torch.bfloat16))
    assert float((b - (beta - gamma * mean_rstd)).abs().max()) <= 1e-6 * float(b.abs().max() + 1)
    gb = torch.randn(O, generator=g).cuda()
    for fmt in (torch.contiguous_format, torch.channels_last):
        gw = torch.randn(O, I, k, k, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=fmt)
        dW, dgamma = ext.conv_bn_fold_bwd(gw, W, gamma, rstd, mean_rstd, gb)
        assert torch.equal(dW, gw.float() * s.view(-1, 1, 1, 1))
        want = rstd.double() * (gw.double() * W.double()).sum((1, 2, 3)) - mean_rstd.double() * gb.double()
        assert float((dgamma.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
