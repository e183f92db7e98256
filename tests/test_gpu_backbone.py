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
    (1, 2048, 512, 3, 4, 1, False, True), (2, 32, 8, 5, 5, 1, True, False), (1, 1024, 2048, 5, 8, 2, False, False)])
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


@pytest.mark.parametrize("N,C,Cout,H,W,relu", [
    (2, 128, 128, 9, 21, True), (1, 256, 256, 16, 32, False), (1, 512, 512, 5, 7, True),
    (3, 64, 256, 8, 16, True), (1, 32, 128, 1, 1, False), (1, 256, 128, 29, 50, True)])
def test_conv3x3_nhwc_matches_torch(N, C, Cout, H, W, relu):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(C + Cout + H)
    x = torch.randn(N, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    w_bf = w.to(torch.bfloat16).float()                       # the kernel rounds the weights to bf16
    b = torch.randn(Cout, generator=g).cuda()
    got = ext.conv3x3_nhwc(x, ext.conv3x3_pack_weight(w), b, Cout, relu=relu)
    want = torch.nn.functional.conv2d(x.float(), w_bf, b, padding=1)
    if relu:
        want = want.relu()
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    d = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    print(f"conv3x3 {C}->{Cout} {H}x{W}: max diff {d:.3e} (scale {scale:.2f})")
    assert d <= scale * 2 ** -8 + 1e-5          # one bf16 rounding of the f32-accumulated result


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
