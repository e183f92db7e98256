"""-m gpu: lifter + Conv3d/BN/ReLU implicit-GEMM kernel and the fused occupancy heads (through the
C ABI) vs the oracle (oracle/decoder.py, float64 CPU)."""
import pytest
import torch

from oracle import decoder as odec

pytestmark = pytest.mark.gpu


def _bn(cout, g):
    return (torch.rand(cout, generator=g) * 0.5 + 0.75, torch.randn(cout, generator=g) * 0.1,
            torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) * 0.5 + 0.75)


def _fold(bn, eps=1e-5):
    w, b, mean, var = bn
    scale = w / torch.sqrt(var + eps)
    return scale, b - mean * scale


CONV_CASES = [
    # name, B, Z, Y, X, Cin   (Y/X deliberately not multiples of the block tile)
    ("base_lifter", 1, 16, 9, 21, 16),
    ("base_conv2", 2, 16, 7, 10, 32),
    ("tiny_z4", 1, 4, 13, 37, 64),
    ("hires_z32_c8", 1, 32, 5, 6, 8),          # BASELINE configs[4]'s lifter: 8 channels, two taps per MFMA step
    ("hires_z32_c8_b2", 2, 32, 3, 9, 8),
    ("z16_c8", 1, 16, 7, 11, 8),
    ("z4_c8", 1, 4, 5, 33, 8),
    ("hires_z32_c32", 1, 32, 4, 7, 32),
    ("z8", 1, 8, 6, 19, 32),
    ("one_pillar", 1, 16, 1, 1, 16),
]


@pytest.mark.parametrize("name,B,Z,Y,X,Cin", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_conv3d_bn_relu_matches_oracle(name, B, Z, Y, X, Cin, layout, precision):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(20 + layout)
    cout = 32
    x = torch.randn(B, Cin, Z, Y, X, generator=g)                  # torch NCDHW
    w = torch.randn(cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    bn = _bn(cout, g)
    ref = odec.conv3d_bn_relu(x.double(), w.double(), *[t.double() for t in bn])
    if layout == 0:
        xin = x.permute(0, 3, 4, 2, 1).contiguous()                # (B, Y, X, Z, Cin)
    else:
        xin = x.permute(0, 3, 4, 1, 2).reshape(B, Y * X, Cin * Z).contiguous()   # BEV embedding
        assert torch.equal(odec.lifter(xin, Z, Y, X), x)           # the lifter view is exactly this
    scale, shift = _fold(bn)
    wp = ext.conv3d_pack_weight(w.cuda(), precision=precision)
    x3 = wp.dtype == torch.int16            # bf16x3 kernels: Cin % 16 == 0, or Cin == 8 (two taps per MFMA step)
    assert x3 == (precision == "bf16x3" and (Cin % 16 == 0 or Cin == 8))
    for xy_major in (False, True):
        out = ext.conv3d_bn_relu(xin.cuda(), wp, scale.cuda(), shift.cuda(), Z, Y, X, Cin, cout,
                                 in_layout=layout, out_xy_major=xy_major)
        torch.cuda.synchronize()
        got = out.cpu().double()
        want = ref.permute(0, 4, 3, 2, 1) if xy_major else ref.permute(0, 3, 4, 2, 1)
        d = float((got - want).abs().max())
        print(f"{name} layout={layout} {precision} xy_major={xy_major}: max|hip - oracle(f64)| = {d:.3e}")
        assert got.shape == want.shape
        # exact f32: summation-order noise; bf16x3: 2^-16 per product over K = 27*Cin terms (bound 1e-3)
        assert d < (1e-4 if x3 else 2e-5)


def test_conv3d_linearity_full_size():
    """Base-config size (200x200x16, 32->32): conv is linear before the ReLU — size-independent
    property checked on the whole grid, plus a spot check of one block against the oracle."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(30)
    Z, Y, X, C = 16, 200, 200, 32
    x = torch.randn(1, Y, X, Z, C, generator=g).cuda()
    w = torch.randn(C, C, 3, 3, 3, generator=g) * 0.05
    wp = ext.conv3d_pack_weight(w.cuda())          # default precision (bf16x3)
    one, zero = torch.ones(C).cuda(), torch.zeros(C).cuda()
    f = lambda t: ext.conv3d_bn_relu(t, wp, one, zero, Z, Y, X, C, C, in_layout=0, relu=False)
    o1, o2 = f(x), f(x * 2.0)
    assert torch.allclose(o2, o1 * 2.0, atol=1e-4, rtol=1e-5)
    ys, xs = slice(90, 100), slice(0, 12)
    crop = x[:, 88:102, 0:14].cpu().double().permute(0, 4, 3, 1, 2)          # halo included
    ref = torch.nn.functional.conv3d(crop, w.double(), padding=1)[:, :, :, 2:12, 0:12]
    got = o1[:, ys, xs].cpu().double().permute(0, 4, 3, 1, 2)
    d = float((got - ref).abs().max())
    print(f"full-size crop: max|hip - oracle(f64)| = {d:.3e}")
    assert d < 1e-4        # bf16x3 products (default precision)


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("n_rows,ncls", [(1, 17), (31, 17), (32, 17), (1000, 17), (4099, 18), (77, 3)])
def test_occ_heads_match_oracle(n_rows, ncls, precision):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(40 + n_rows)
    feat = torch.randn(n_rows, 32, generator=g) * 2.0
    feat[0, :] = 30.0          # drives Softplus past its threshold (x > 20 -> identity)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.3
    ws = (mk(64, 32), mk(64), mk(ncls, 64), mk(ncls), mk(64, 32), mk(64), mk(2, 64), mk(2))
    occ_ref, flow_ref = odec.heads(feat.double(), *[t.double() for t in ws])
    occ, flow = ext.occ_heads(feat.cuda(), *[t.cuda() for t in ws], precision=precision)
    torch.cuda.synchronize()
    # row 0 (features = 30) produces outputs of magnitude ~1e2: compare relative to the output scale
    d1 = float((occ.cpu().double() - occ_ref).abs().max()) / max(1.0, float(occ_ref.abs().max()))
    d2 = float((flow.cpu().double() - flow_ref).abs().max()) / max(1.0, float(flow_ref.abs().max()))
    print(f"heads n={n_rows} ncls={ncls}: occ {d1:.3e} flow {d2:.3e} (relative to output scale)")
    assert occ.shape == (n_rows, ncls) and flow.shape == (n_rows, 2)
    # f32: exact-f32 MFMA; bf16x3: 2^-16 per product on O(1) operands over K = 32 and 64 (bound 1e-3 end to end)
    tol = 2e-5 if precision == "f32" else 1e-4
    assert d1 < tol and d2 < tol
    d_typ = float((occ[1:].cpu().double() - occ_ref[1:]).abs().max()) if n_rows > 1 else 0.0
    assert d_typ < (5e-5 if precision == "f32" else 3e-4)
    # fused decode: the class written by the same pass is exactly the reference's softmax(-1).argmax(-1)
    # (bevformer_occ_head.py:210-212) of the logits the kernel wrote, and the logits are bit-identical with or
    # without it
    occ2, flow2, cls = ext.occ_heads(feat.cuda(), *[t.cuda() for t in ws], decode=True, precision=precision)
    assert torch.equal(occ2, occ) and torch.equal(flow2, flow)
    assert cls.dtype == torch.int64 and cls.shape == (n_rows,)
    assert torch.equal(cls, occ.softmax(-1).argmax(-1))


def test_fused_decode_ties_take_the_first_index():
    """Equal logits: torch.argmax (and therefore the reference's decode) returns the first maximal index."""
    from occnet_amd import ext
    z = lambda *s: torch.zeros(*s)
    feat = torch.randn(64, 32, generator=torch.Generator().manual_seed(2))
    b2 = torch.tensor([0.5, 2.0, 2.0, -1.0, 2.0])                 # classes 1, 2, 4 tie for every voxel
    occ, flow, cls = ext.occ_heads(feat.cuda(), z(64, 32).cuda(), z(64).cuda(), z(5, 64).cuda(), b2.cuda(),
                                   z(64, 32).cuda(), z(64).cuda(), z(2, 64).cuda(), z(2).cuda(), decode=True)
    assert torch.equal(cls.cpu(), torch.ones(64, dtype=torch.int64))
    assert torch.equal(cls, occ.softmax(-1).argmax(-1))


def test_unsupported_shapes_raise_unsupported():
    from occnet_amd import ext
    from occnet_amd._lib import OccAmdUnsupported
    with pytest.raises(OccAmdUnsupported):
        ext.conv3d_pack_weight(torch.randn(16, 16, 3, 3, 3).cuda())      # Cout != 32
    with pytest.raises(OccAmdUnsupported):
        ext.occ_heads(torch.randn(8, 16).cuda(), torch.randn(64, 16).cuda(), torch.randn(64).cuda(),
                      torch.randn(17, 64).cuda(), torch.randn(17).cuda(), torch.randn(64, 16).cuda(),
                      torch.randn(64).cuda(), torch.randn(2, 64).cuda(), torch.randn(2).cuda())


@pytest.mark.parametrize("B,Y,X,Z", [(1, 9, 21, 16), (2, 7, 10, 16), (1, 1, 1, 16), (1, 16, 40, 16),
                                     (1, 5, 9, 32), (2, 3, 4, 32), (1, 1, 1, 32)])
def test_fused_conv_heads_decode_equals_two_launches(B, Y, X, Z):
    """occ_conv3d_heads_decode_bf16x3_f32 (second convolution + BN + ReLU + both heads + decode in one kernel; the
    convolution's output never reaches HBM) vs the two launches it replaces — conv3d_bn_relu(out_xy_major) then
    occ_heads(decode=True) — and vs the float64 oracle chain.  The fused kernel contracts the heads' first layer in a
    permuted k order, so logits agree to fp32 summation noise; the decoded classes must be the argmax of ITS logits
    bit for bit (first index on ties).  Ragged tiles (Y, X not multiples of the 2 x 8 block tile), batch 2."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(61)
    C, ncls = 32, 17                      # Z = 32: BASELINE configs[4] (2 x 4 pillars per block, one pillar per row tile)
    x = torch.randn(B, Y, X, Z, C, generator=g)
    w = torch.randn(C, C, 3, 3, 3, generator=g) * (2.0 / (C * 27)) ** 0.5
    bn = _bn(C, g)
    scale, shift = _fold(bn)
    hw = dict(w1o=torch.randn(64, C, generator=g) / C ** 0.5, b1o=torch.randn(64, generator=g) * 0.1,
              w2o=torch.randn(ncls, 64, generator=g) / 8, b2o=torch.randn(ncls, generator=g) * 0.1,
              w1f=torch.randn(64, C, generator=g) / C ** 0.5, b1f=torch.randn(64, generator=g) * 0.1,
              w2f=torch.randn(2, 64, generator=g) / 8, b2f=torch.randn(2, generator=g) * 0.1)
    hc = [hw[k].cuda() for k in ("w1o", "b1o", "w2o", "b2o", "w1f", "b1f", "w2f", "b2f")]
    wp = ext.conv3d_pack_weight(w.cuda(), precision="bf16x3")
    pack = ext.conv3d_heads_pack(*hc)
    occ, flow, cls = ext.conv3d_heads_decode(x.cuda(), wp, scale.cuda(), shift.cuda(), pack, Z, Y, X, ncls)
    feat = ext.conv3d_bn_relu(x.cuda(), wp, scale.cuda(), shift.cuda(), Z, Y, X, C, C, in_layout=0, out_xy_major=True)
    occ2, flow2, cls2 = ext.occ_heads(feat, *hc, decode=True, precision="bf16x3")
    torch.cuda.synchronize()
    assert occ.shape == (B, X, Y, Z, ncls) and flow.shape == (B, X, Y, Z, 2) and cls.shape == (B, X, Y, Z)
    d_occ, d_flow = float((occ - occ2).abs().max()), float((flow - flow2).abs().max())
    print(f"fused vs two launches ({B},{Y},{X}): occ {d_occ:.3e} flow {d_flow:.3e}")
    assert d_occ < 2e-5 and d_flow < 2e-5
    assert torch.equal(cls, occ.softmax(-1).argmax(-1))                      # decode == argmax of its own logits
    # float64 oracle: conv + BN + ReLU, permute to (B, X, Y, Z, C), the two MLPs
    ref = odec.conv3d_bn_relu(x.permute(0, 4, 3, 1, 2).double(), w.double(), *[t.double() for t in bn])
    f64 = ref.permute(0, 4, 3, 2, 1)
    o_ref = torch.nn.functional.linear(torch.nn.functional.softplus(torch.nn.functional.linear(f64, hw["w1o"].double(), hw["b1o"].double())), hw["w2o"].double(), hw["b2o"].double())
    f_ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(f64, hw["w1f"].double(), hw["b1f"].double())), hw["w2f"].double(), hw["b2f"].double())
    d1, d2 = float((occ.cpu().double() - o_ref).abs().max()), float((flow.cpu().double() - f_ref).abs().max())
    print(f"fused vs oracle(f64): occ {d1:.3e} flow {d2:.3e}")
    assert d1 < 2e-4 and d2 < 2e-4


@pytest.mark.parametrize("B,Z,Y,X,Cin,layout", [(1, 16, 5, 7, 16, 1), (2, 16, 4, 6, 32, 0), (1, 32, 3, 5, 8, 1),
                                                (1, 4, 6, 9, 64, 0)])
def test_conv3d_autograd_function_matches_float64_autograd(B, Z, Y, X, Cin, layout):
    """ext.Conv3dX3Function (the decoder's training convolution): forward, dx (the same kernel on the flipped transposed
    weight) and dW (27 shifted-row linear_wgrad calls on zero-padded copies) vs torch.autograd through F.conv3d in
    float64."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(71)
    x = torch.randn(B, Cin, Z, Y, X, generator=g)
    w = torch.randn(32, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    go = torch.randn(B, 32, Z, Y, X, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = torch.nn.functional.conv3d(xr, wr, padding=1)
    ref.backward(go.double())
    if layout == 0:
        xin = x.permute(0, 3, 4, 2, 1).contiguous()
    else:
        xin = x.permute(0, 3, 4, 1, 2).reshape(B, Y * X, Cin * Z).contiguous()
    xin = xin.cuda().requires_grad_(True)
    wc = w.cuda().requires_grad_(True)
    out = ext.conv3d_autograd(xin, wc, Z, Y, X, in_layout=layout)               # (B, Y, X, Z, 32)
    out.backward(go.permute(0, 3, 4, 2, 1).contiguous().cuda())
    torch.cuda.synchronize()
    d_out = float((out.detach().cpu().double() - ref.detach().permute(0, 3, 4, 2, 1)).abs().max())
    gx_ref = xr.grad.permute(0, 3, 4, 2, 1) if layout == 0 else xr.grad.permute(0, 3, 4, 1, 2).reshape(B, Y * X, Cin * Z)
    d_x = float((xin.grad.cpu().double() - gx_ref).abs().max())
    d_w = float((wc.grad.cpu().double() - wr.grad).abs().max())
    s_w = float(wr.grad.abs().max())
    print(f"conv3d autograd Z={Z} Cin={Cin} layout={layout}: out {d_out:.2e} dx {d_x:.2e} dW {d_w:.2e} (scale {s_w:.1f})")
    assert d_out < 1e-4 and d_x < 2e-4 and d_w < 2e-4 * max(1.0, s_w)
