"""-m gpu: the MFMA Linear kernel with fused bias / ReLU / residual / LayerNorm epilogues (through the
C ABI) vs the oracle (oracle/dense.py, float64 CPU)."""
import pytest
import torch

from oracle import dense as odense

pytestmark = pytest.mark.gpu


def _mk(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


CASES = [
    # name,            M,    K1,  K2,  N,  bias, act,    addend, residual, ln
    ("plain",          100,  256, 0,   256, False, None,  False,  False,    False),
    ("value_proj",     1031, 256, 0,   256, True,  None,  False,  False,    False),
    ("ffn1_relu",      333,  256, 0,   512, True,  'relu', False, False,    False),
    ("ffn2_res_ln",    333,  512, 0,   256, True,  None,  False,  True,     True),
    ("tsa_query",      257,  256, 256, 192, True,  None,  True,   False,    False),
    ("sca_query",      64,   256, 0,   768, True,  None,  False,  False,    False),
    ("small_n_ln",     45,   64,  0,   128, True,  None,  False,  True,     True),
    ("n_100_ln",       70,   32,  0,   100, True,  'relu', False, True,     True),
    ("two_seg_noadd",  33,   32,  64,  36,  False, None,  False,  True,     False),
    ("k16_chunks",     50,   48,  16,  64,  True,  None,  True,   False,    False),
    ("one_row",        1,    256, 0,   256, True,  None,  False,  True,     True),
    ("m160",           160,  256, 0,   256, True,  None,  False,  True,     True),
    ("m481_tail",      481,  256, 256, 192, True,  None,  True,   False,    False),
    ("m130_n768",      130,  256, 0,   768, True,  'relu', False, False,    False),
    ("k512_n260",      200,  512, 0,   260, True,  None,  False,  True,     False),
    # the encoder's tall shapes: one / several column blocks, N = 192, ragged tails, every epilogue
    ("ws_out_proj",    4099, 256, 0,   256, True,  None,  False,  True,     True),
    ("ws_value_proj",  1031, 256, 0,   256, True,  None,  False,  False,    False),
    ("ws_tsa_q_192",   2500, 256, 0,   192, False, None,  False,  True,     False),
    ("ws_sca_q_768",   3001, 256, 0,   768, True,  None,  False,  False,    False),
    ("ws_ffn1_relu",   20000, 256, 0,  512, True,  'relu', False, False,    False),
    ("ws_n100_ln",     1500, 256, 0,   100, True,  'relu', False, True,     True),
    # taller cases (>= 8192 rows, ragged last block): one / two / three column blocks, N = 192 / 64 / 128, K = 512 from one
    # or two segments (+ addend), every epilogue  (written for the activation-resident variant, tools_dev/lab/linear_x3r.hip)
    ("xr_sca_q_768",   33001, 256, 0,  768, True,  None,  False,  False,    False),
    ("xr_ffn1_relu",   8200, 256, 0,   512, True,  'relu', False, False,    False),
    ("xr_out_proj_ln", 10001, 256, 0,  256, True,  None,  False,  True,     True),
    ("xr_ffn2_ln",     9000, 512, 0,   256, True,  None,  False,  True,     True),
    ("xr_tsa_q_hist",  8200, 256, 256, 192, True,  None,  True,   False,    False),
    ("xr_tsa_q_res",   8255, 256, 0,   192, False, None,  False,  True,     False),
    ("xr_n64_relu_res", 8193, 256, 0,  64,  True,  'relu', False, True,     False),
    ("xr_n128_ln",     8300, 512, 0,   128, True,  'relu', False, True,     True),
]


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("name,M,K1,K2,N,bias,act,addend,residual,ln", CASES, ids=[c[0] for c in CASES])
def test_linear_matches_oracle(name, M, K1, K2, N, bias, act, addend, residual, ln, precision, monkeypatch):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(50)
    a = _mk(g, M, K1)
    a2 = _mk(g, M, K2) if K2 else None
    a2_add = _mk(g, M, K2) if (K2 and addend) else None
    w = _mk(g, N, K1 + K2, scale=(1.0 / (K1 + K2)) ** 0.5)
    b = _mk(g, N, scale=0.1) if bias else None
    res = _mk(g, M, N) if residual else None
    lnp = (torch.rand(N, generator=g) + 0.5, _mk(g, N, scale=0.1), 1e-5) if ln else None
    d64 = lambda t: None if t is None else t.double()
    ref = odense.linear_chain(d64(a), d64(w), d64(b), d64(a2), d64(a2_add), act, d64(res),
                              None if lnp is None else (lnp[0].double(), lnp[1].double(), lnp[2]))
    c = lambda t: None if t is None else t.cuda()
    out = ext.linear(c(a), c(w), c(b), a2=c(a2), a2_add=c(a2_add), act=act, residual=c(res),
                     ln=None if lnp is None else (lnp[0].cuda(), lnp[1].cuda(), lnp[2]),
                     precision=precision)
    torch.cuda.synchronize()
    d = float((out.cpu().double() - ref).abs().max())
    print(f"{name}/{precision}: max|hip - oracle(f64)| = {d:.3e}")
    assert out.shape == (M, N)
    # f32: fp32 roundoff; bf16x3: 2^-16 per product on O(1) operands, K <= 512 (bound 1e-3 end to end)
    assert d < (2e-5 if precision == "f32" else 2e-4)


def test_linear_strided_rows_and_3d_shapes():
    """A given as a column slice of a wider buffer (row stride > K) and as a (bs, nq, K) tensor."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(51)
    wide = _mk(g, 90, 768).cuda()
    w = _mk(g, 64, 256, scale=0.06).cuda()
    out = ext.linear(wide[:, 256:512], w, precision="f32")
    ref = torch.nn.functional.linear(wide[:, 256:512].cpu().double(), w.cpu().double())
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5
    x = _mk(g, 2, 45, 256).cuda()
    out3 = ext.linear(x, w, precision="f32")
    assert out3.shape == (2, 45, 64)
    ref3 = torch.nn.functional.linear(x.cpu().double(), w.cpu().double())
    assert float((out3.cpu().double() - ref3).abs().max()) < 2e-5
    out4 = ext.linear(wide[:, 256:512], w, precision="bf16x3")
    assert float((out4.cpu().double() - ref).abs().max()) < 2e-4


def test_bf16x3_asymmetric_identity():
    """A = I with an asymmetric W: catches row/column or k-ordering mistakes in the MFMA fragment layout."""
    from occnet_amd import ext
    K = N = 64
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 64.0 - 1.0   # exactly bf16x2-representable
    out = ext.linear(torch.eye(K).cuda(), w.cuda(), precision="bf16x3")
    assert torch.equal(out.cpu(), w.t().contiguous())


def test_linear_full_size_value_proj_linearity():
    """Base-config value projection (6 x 30825 rows, 256 -> 256): size-independent properties
    (linearity in A, bias shift) on the full problem + a row sample against the oracle."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(52)
    M = 6 * 30825
    a = torch.randn(M, 256, generator=g).cuda()
    w = (_mk(g, 256, 256, scale=1 / 16)).cuda()
    b = _mk(g, 256, scale=0.1).cuda()
    idx = torch.randint(0, M, (512,), generator=g)
    ref = torch.nn.functional.linear(a[idx.cuda()].cpu().double(), w.cpu().double(), b.cpu().double())
    for precision, tol in (("f32", 2e-5), ("bf16x3", 2e-4)):
        o1 = ext.linear(a, w, b, precision=precision)
        o2 = ext.linear(a * 2.0, w, b, precision=precision)       # scaling by 2 is exact in both splits
        assert torch.allclose(o2 - b, (o1 - b) * 2.0, atol=2e-5, rtol=1e-5)
        d = float((o1[idx.cuda()].cpu().double() - ref).abs().max())
        print(f"value_proj full size/{precision}: row-sample max|hip - oracle(f64)| = {d:.3e}")
        assert d < tol


def test_linear_unsupported_and_errors():
    from occnet_amd import ext
    from occnet_amd._lib import OccAmdError, OccAmdUnsupported
    x = torch.randn(8, 40).cuda()
    with pytest.raises(OccAmdUnsupported):       # K not a multiple of 16
        ext.linear(x, torch.randn(16, 40).cuda())
    with pytest.raises(OccAmdUnsupported):       # LayerNorm over more than 256 outputs
        ext.linear(torch.randn(8, 64).cuda(), torch.randn(512, 64).cuda(),
                   ln=(torch.ones(512).cuda(), torch.zeros(512).cuda(), 1e-5))
    with pytest.raises(OccAmdError):             # host tensor
        ext.linear(torch.randn(8, 64), torch.randn(16, 64).cuda())
    with pytest.raises(OccAmdError):             # weight shape mismatch
        ext.linear(torch.randn(8, 64).cuda(), torch.randn(16, 32).cuda())


@pytest.mark.parametrize("G,rpg,K,N,ogr,row0,nbias", [
    (3, 50, 256, 256, 120, 7, 3),      # groups land at out_row0 inside longer per-camera blocks
    (6, 375, 256, 256, 400, 25, 6),    # smallest FPN level of the base config
    (4, 33, 64, 128, 40, 0, 2),        # bias table shorter than the group count (batch > 1: g % num_cam)
    (1, 200, 96, 192, 200, 0, 1),      # 3 K chunks, 6 column tiles
    (2, 70, 32, 36, 70, 0, 0),         # one chunk, ragged columns, no bias
])
def test_value_proj_bf16_matches_torch(G, rpg, K, N, ogr, row0, nbias):
    from occnet_amd import ext
    g = torch.Generator().manual_seed(G * 100 + K + N)
    a = torch.randn(G * rpg, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    gb = torch.randn(nbias, N, generator=g).cuda() if nbias else None
    out = torch.full(((G - 1) * ogr + row0 + rpg + 5, N), float('nan'), device='cuda')
    ext.value_proj_bf16(a, w, gb, out, rows_per_group=rpg, out_group_rows=ogr, out_row0=row0)
    want = a.double() @ w.double().t()
    worst = 0.0
    for gi in range(G):
        ref = want[gi * rpg:(gi + 1) * rpg] + (gb[gi % nbias].double() if nbias else 0.0)
        got = out[gi * ogr + row0: gi * ogr + row0 + rpg].double()
        worst = max(worst, float((got - ref).abs().max()))
    print(f"value_proj_bf16 G={G} rpg={rpg} K={K} N={N}: max diff {worst:.3e}")
    assert worst < 3e-5
    # rows outside the groups' windows are untouched
    mask = torch.ones(out.shape[0], dtype=torch.bool, device='cuda')
    for gi in range(G):
        mask[gi * ogr + row0: gi * ogr + row0 + rpg] = False
    assert torch.isnan(out[mask]).all() and not torch.isnan(out[~mask]).any()


def test_value_proj_bf16_multi_segment_single_launch():
    """Four 'levels' of different sizes into per-camera blocks of one output, one launch."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(11)
    cams, K, N = 3, 64, 256
    hws = [130, 37, 9, 2]
    starts = [0, 130, 167, 176]
    total = sum(hws)
    a_list = [torch.randn(cams * hw, K, generator=g).cuda().to(torch.bfloat16) for hw in hws]
    w = (torch.randn(N, K, generator=g) / 8).cuda()
    gb = torch.randn(len(hws), cams, N, generator=g).cuda()
    out = torch.full((cams * total, N), float('nan'), device='cuda')
    ext.value_proj_bf16(a_list, w, gb, out, rows_per_group=hws, out_group_rows=total, out_row0=starts)
    assert not torch.isnan(out).any()
    o = out.view(cams, total, N).double()
    for l, hw in enumerate(hws):
        ref = (a_list[l].double() @ w.double().t()).view(cams, hw, N) + gb[l].double()[:, None, :]
        d = float((o[:, starts[l]:starts[l] + hw] - ref).abs().max())
        print(f"level {l} ({hw} px): max diff {d:.3e}")
        assert d < 3e-5


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
def test_value_proj_planes_equals_separate_launches(out_dtype):
    """occ_value_proj_bf16_planes (the four encoder layers' value projections in one launch, feature rows read once,
    column blocks of a row block dealt to one XCD) is BIT-IDENTICAL to one occ_value_proj_bf16 launch per layer:
    ragged segment sizes (the last group of 8 row blocks is padded), 3 cameras, 4 planes."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(12)
    cams, K, N, P = 3, 256, 256, 4
    hws = [1300, 333, 90, 20]
    starts = [0, 1300, 1633, 1723]
    total = sum(hws) + 1                  # 1 744: fp16 outputs (pixel-pair order) need an even block per camera
    a_list = [torch.randn(cams * hw, K, generator=g).cuda().to(torch.bfloat16) for hw in hws]
    ws = [(torch.randn(N, K, generator=g) / 16).cuda() for _ in range(P)]
    gbs = [torch.randn(len(hws), cams, N, generator=g).cuda() for _ in range(P)]
    out = torch.full((P, cams * total, N), float('nan'), device='cuda', dtype=out_dtype)
    ext.value_proj_bf16_planes(a_list, ws, gbs, out, rows_per_group=hws, out_group_rows=total, out_row0=starts)
    written = torch.zeros(cams, total, dtype=torch.bool, device='cuda')
    for st, hw in zip(starts, hws):
        written[:, st:st + hw] = True
    rows_of = lambda t: (ext.sca_unpair_layout(t.view(cams, total, N // 32, 32)) if out_dtype == torch.float16
                         else t.view(cams, total, N // 32, 32))
    assert not torch.isnan(rows_of(out[0]).float()[written]).any()
    for p in range(P):
        ref = torch.full((cams * total, N), float('nan'), device='cuda', dtype=out_dtype)
        ext.value_proj_bf16(a_list, ws[p], gbs[p], ref, rows_per_group=hws, out_group_rows=total, out_row0=starts)
        assert torch.equal(rows_of(out[p])[written], rows_of(ref)[written]), p


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("P", [1, 4])
def test_value_proj_activation_resident_kernel(out_dtype, P):
    """K = 256, N % 256 == 0, every group >= 128 rows: the activation-resident kernel (128-row tile in LDS by LDS-DMA, all
    column passes without a barrier).  Ragged segments (last block of a segment partly past M; blocks that straddle two
    camera groups, whose rows take different bias rows and land in different output blocks), bias table shorter than
    the group count, P stacked projections.  Checked against float64 (fp16 output: against the float64 result rounded
    to fp16, within one fp16 ulp) and — same rows plus one extra short segment, which sends the launch to the tiled
    kernel — against the tiled kernel."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(13 + P)
    cams, K, N, nb = 5, 256, 256, 3
    hws = [1300, 333, 200, 129]
    starts = [4, 1310, 1650, 1860]
    total = 2020
    a_list = [torch.randn(cams * hw, K, generator=g).cuda().to(torch.bfloat16) for hw in hws]
    ws = [(torch.randn(N, K, generator=g) / 16).cuda() for _ in range(P)]
    gbs = [torch.randn(len(hws), nb, N, generator=g).cuda() for _ in range(P)]
    out = torch.full((P, cams * total, N), float('nan'), device='cuda', dtype=out_dtype)
    if P == 1:
        ext.value_proj_bf16(a_list, ws[0], gbs[0], out[0], rows_per_group=hws, out_group_rows=total, out_row0=starts)
    else:
        ext.value_proj_bf16_planes(a_list, ws, gbs, out, rows_per_group=hws, out_group_rows=total, out_row0=starts)
    # the tiled kernel on the same rows: a fifth 20-row segment makes the launch ineligible for the resident kernel
    a5 = a_list + [torch.randn(cams * 20, K, generator=g).cuda().to(torch.bfloat16)]
    gb5 = [torch.cat([gb, torch.randn(1, nb, N, generator=g).cuda()]) for gb in gbs]
    tiled = torch.full((P, cams * total, N), float('nan'), device='cuda', dtype=out_dtype)
    if P == 1:
        ext.value_proj_bf16(a5, ws[0], gb5[0], tiled[0], rows_per_group=hws + [20], out_group_rows=total,
                            out_row0=starts + [1995])
    else:
        ext.value_proj_bf16_planes(a5, ws, gb5, tiled, rows_per_group=hws + [20], out_group_rows=total,
                                   out_row0=starts + [1995])
    written = torch.zeros(cams * total, dtype=torch.bool, device='cuda')
    worst = worst_t = 0.0
    rows_of = lambda x: (ext.sca_unpair_layout(x.view(cams, total, N // 32, 32)).reshape(cams, total, N)
                         if out_dtype == torch.float16 else x.view(cams, total, N))       # fp16: pixel-pair order
    for p in range(P):
        o = rows_of(out[p])
        t = rows_of(tiled[p])
        for l, hw in enumerate(hws):
            ref = (a_list[l].double() @ ws[p].double().t()).view(cams, hw, N) + \
                gbs[p][l].double()[torch.arange(cams) % nb][:, None, :]
            got = o[:, starts[l]:starts[l] + hw]
            if out_dtype == torch.float16:
                # the kernel rounds to fp16 an f32 sum that is within 3e-5 of the float64 result: half an ulp + 3e-5
                ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14, device='cuda', dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -10
                assert bool(((got.double() - ref).abs() <= 0.5 * ulp + 3e-5).all())
            else:
                worst = max(worst, float((got.double() - ref).abs().max()))
            worst_t = max(worst_t, float((got.double() - t[:, starts[l]:starts[l] + hw].double()).abs().max()))
            if p == 0:
                written.view(cams, total)[:, starts[l]:starts[l] + hw] = True
    print(f"resident value projection P={P} {out_dtype}: max diff vs float64 {worst:.3e}, vs tiled kernel {worst_t:.3e}")
    assert worst < 3e-5 and worst_t < (1.6e-2 if out_dtype == torch.float16 else 2e-5)   # fp16: one ulp at |v| < 16
    # nothing outside the groups' windows is touched
    for p in range(P):
        o = rows_of(out[p]).reshape(cams * total, N).float()
        assert torch.isnan(o[~written]).all() and not torch.isnan(o[written]).any()


WGRAD_CASES = [
    # name,          M,      N,   K
    ("ffn1",         40000,  512, 256),
    ("ffn2",         40000,  256, 512),
    ("sca_query",    40000,  768, 256),
    ("tsa_offsets",  40000,  128, 512),
    ("tsa_weights",  40000,  64,  512),
    ("ragged_m",     1237,   256, 256),      # rows not a multiple of 16, chunk tails
    ("tiny_m",       5,      32,  16),
    ("odd_cols",     300,    100, 48),       # N, K not multiples of 32: clamped loads, masked stores
    ("one_tile",     777,    17,  33),
]


@pytest.mark.parametrize("name,M,N,K", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_linear_wgrad_matches_f64(name, M, N, K):
    """dW = dY^T X, db = sum dY on the bf16x3 kernel vs float64 on the host; bound 1e-3 relative to the largest
    gradient entry (measured ~1e-5: two-term split, products exact to 2^-16, f32 accumulation over M rows)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(61)
    dy = _mk(g, M, N)
    x = _mk(g, M, K)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    dw, db = ext.linear_wgrad(dy.cuda(), x.cuda())
    dw2, _ = ext.linear_wgrad(dy.cuda(), x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "the chunked reduction must be deterministic"
    ew = float((dw.cpu().double() - ref_w).abs().max() / ref_w.abs().max())
    eb = float((db.cpu().double() - ref_b).abs().max() / ref_b.abs().max())
    print(f"{name}: rel err dW {ew:.2e} db {eb:.2e}")
    assert ew < 1e-3 and eb < 1e-3
    assert ew < 1e-4, "bf16x3 should be ~1e-5; 1e-4 means a term is missing"


def test_linear_wgrad_strided_rows():
    """dy / x as column views of wider matrices (row stride > row length), as the SCA query Linear's two outputs."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(62)
    wide_y = _mk(g, 500, 768).cuda()
    wide_x = _mk(g, 500, 512).cuda()
    dy, x = wide_y[:, 512:], wide_x[:, :256]
    dw, db = ext.linear_wgrad(dy, x)
    ref = dy.double().t() @ x.double()
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < 1e-4
    assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-3


@pytest.mark.parametrize("act", [None, "relu"])
@pytest.mark.parametrize("shape", [(2, 300, 256, 512), (1, 1000, 512, 256), (1, 64, 256, 64)],
                         ids=["b2_256_512", "512_256", "256_64"])
def test_linear_autograd_function_matches_f64_autograd(shape, act):
    """LinearX3Function (forward, dx, dW, db on the own kernels) vs float64 autograd of F.linear(+ReLU)."""
    import torch.nn.functional as F
    from occnet_amd import ext
    B, Q, K, N = shape
    g = torch.Generator().manual_seed(63)
    x = _mk(g, B, Q, K)
    w = _mk(g, N, K, scale=K ** -0.5)
    b = _mk(g, N, scale=0.1)
    go = _mk(g, B, Q, N)
    xd, wd, bd = (t.cuda().requires_grad_() for t in (x, w, b))
    y = ext.linear_autograd(xd, wd, bd, act=act)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith("LinearX3Function")
    y.backward(go.cuda())
    torch.cuda.synchronize()
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    if act:
        # ReLU's gradient is discontinuous at 0: a pre-activation within the forward's rounding error of zero
        # (|z| < 5e-6: about one element per 250 000) may sit on the other side in float64.  The fused op's
        # contract is "the backward masks with the forward's own output", so the reference uses that mask.
        mask = (y.detach().cpu() > 0).double()
        flips = int((mask - (yr.detach() > 0).double()).abs().sum())
        assert flips <= max(2, int(1e-4 * mask.numel())), flips
        yr = yr * mask
    yr.backward(go.double())
    rel = lambda a, r: float((a.cpu().double() - r).abs().max() / r.abs().max())
    errs = dict(y=rel(y.detach(), yr.detach()), dx=rel(xd.grad, xr.grad), dw=rel(wd.grad, wr.grad),
                db=rel(bd.grad, br.grad))
    print(shape, act, {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-3
    assert max(errs.values()) < 2e-4


def test_x3linear_module_is_a_state_dict_compatible_linear():
    """X3Linear keeps nn.Linear's parameter names; in training it routes through the own kernels, under no_grad it
    is F.linear."""
    import torch.nn as nn
    from occnet_amd.plugin.bricks import X3Linear
    ref = nn.Linear(256, 128)
    m = X3Linear(256, 128)
    m.load_state_dict(ref.state_dict())
    m.cuda()
    x = torch.randn(3, 50, 256, device="cuda")
    y = m(x)
    assert type(y.grad_fn).__name__.startswith("LinearX3Function")
    with torch.no_grad():
        y0 = m(x)
    assert y0.grad_fn is None
    assert float((y - y0).abs().max()) < 1e-3


# ---- row-local Linear chains (csrc/linear_chain_x3.hip) vs the float64 oracle of the same op sequence ---------------

def _chain_inputs(g, M, n2):
    s = (1.0 / 256) ** 0.5
    t = dict(a=_mk(g, M, 256), res=_mk(g, M, 256), w1=_mk(g, 256, 256, scale=s), b1=_mk(g, 256, scale=0.1),
             ln_g=torch.rand(256, generator=g) + 0.5, ln_b=_mk(g, 256, scale=0.1),
             w2=_mk(g, n2, 256, scale=s), b2=_mk(g, n2, scale=0.1))
    return t


@pytest.mark.parametrize("M,n2,act2", [(64, 768, None), (1, 256, None), (77, 768, None), (4099, 768, None),
                                       (333, 512, 'relu'), (130, 192, None), (40000, 768, None),
                                       (512 * 64 + 101, 768, None)])      # thin last round: 32-row tiles, ragged end
def test_linear_ln_chain_matches_oracle(M, n2, act2):
    """Program A: LayerNorm(a W1^T + b1 + res) and act2(y W2^T + b2) — asymmetric random operands, ragged last
    block, one / two / three tail passes, a tail narrower than a pass."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(61)
    t = _chain_inputs(g, M, n2)
    d = {k: v.double() for k, v in t.items()}
    y_ref = odense.linear_chain(d['a'], d['w1'], d['b1'], residual=d['res'], ln=(d['ln_g'], d['ln_b'], 1e-5))
    z_ref = odense.linear_chain(y_ref, d['w2'], d['b2'], act=act2)
    c = {k: v.cuda() for k, v in t.items()}
    y, z = ext.linear_ln_chain(c['a'], c['res'], c['w1'], c['b1'], (c['ln_g'], c['ln_b'], 1e-5), c['w2'], c['b2'],
                               act2=act2)
    torch.cuda.synchronize()
    dy = float((y.cpu().double() - y_ref).abs().max())
    dz = float((z.cpu().double() - z_ref).abs().max())
    print(f"chain A M={M} n2={n2}: max|y - oracle| = {dy:.3e}, max|z - oracle| = {dz:.3e}")
    assert y.shape == (M, 256) and z.shape == (M, n2)
    assert dy < 2e-4 and dz < 2e-4
    # and against the one-launch-per-Linear path it replaces (same arithmetic class)
    y1 = ext.linear(c['a'], c['w1'], c['b1'], residual=c['res'], ln=(c['ln_g'], c['ln_b'], 1e-5))
    z1 = ext.linear(y1, c['w2'], c['b2'], act=act2)
    assert float((y - y1).abs().max()) < 1e-4 and float((z - z1).abs().max()) < 1e-4


@pytest.mark.parametrize("M,tail,term", [(64, False, False), (77, True, True), (1, True, False), (4099, True, True),
                                         (40000, True, True), (40000, False, False), (512 * 64 + 101, True, True)])
def test_encoder_ffn_chain_matches_oracle(M, tail, term):
    """Program B: output_proj + LN, FFN (512 hidden) + LN, and the optional tail (192 query columns with a per-row
    term + 256 value columns)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(62)
    s = (1.0 / 256) ** 0.5
    t = dict(a=_mk(g, M, 256), res=_mk(g, M, 256), wo=_mk(g, 256, 256, scale=s), bo=_mk(g, 256, scale=0.1),
             g1=torch.rand(256, generator=g) + 0.5, be1=_mk(g, 256, scale=0.1),
             w1=_mk(g, 512, 256, scale=s), b1=_mk(g, 512, scale=0.1),
             w2=_mk(g, 256, 512, scale=(1.0 / 512) ** 0.5), b2=_mk(g, 256, scale=0.1),
             g2=torch.rand(256, generator=g) + 0.5, be2=_mk(g, 256, scale=0.1),
             wq=_mk(g, 192, 256, scale=s), qt=_mk(g, M, 192), wv=_mk(g, 256, 256, scale=s), bv=_mk(g, 256, scale=0.1))
    d = {k: v.double() for k, v in t.items()}
    x2 = odense.linear_chain(d['a'], d['wo'], d['bo'], residual=d['res'], ln=(d['g1'], d['be1'], 1e-5))
    h = odense.linear_chain(x2, d['w1'], d['b1'], act='relu')
    y_ref = odense.linear_chain(h, d['w2'], d['b2'], residual=x2, ln=(d['g2'], d['be2'], 1e-5))
    c = {k: v.cuda() for k, v in t.items()}
    tl = (c['wq'], c['qt'] if term else None, c['wv'], c['bv']) if tail else None
    y, zq, zv = ext.encoder_ffn_chain(c['a'], c['res'], c['wo'], c['bo'], (c['g1'], c['be1'], 1e-5), c['w1'], c['b1'],
                                      c['w2'], c['b2'], (c['g2'], c['be2'], 1e-5), tail=tl)
    torch.cuda.synchronize()
    dy = float((y.cpu().double() - y_ref).abs().max())
    print(f"chain B M={M} tail={tail}: max|y - oracle| = {dy:.3e}")
    assert y.shape == (M, 256) and dy < 3e-4
    if tail:
        zq_ref = odense.linear_chain(y_ref, d['wq'], None, residual=d['qt'] if term else None)
        zv_ref = odense.linear_chain(y_ref, d['wv'], d['bv'])
        dq = float((zq.cpu().double() - zq_ref).abs().max())
        dv = float((zv.cpu().double() - zv_ref).abs().max())
        print(f"   max|zq - oracle| = {dq:.3e}, max|zv - oracle| = {dv:.3e}")
        assert zq.shape == (M, 192) and zv.shape == (M, 256) and dq < 3e-4 and dv < 3e-4
    else:
        assert zq is None and zv is None


@pytest.mark.parametrize("M,term", [(1, False), (77, True), (4099, True), (40000, True), (512 * 64 + 101, False)])
def test_linear_pair_chain_matches_oracle(M, term):
    """Program C: the first layer's TSA query Linears (192 columns + a per-row term) and value projection (256 columns)
    of the same rows in one launch."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(63)
    s = (1.0 / 256) ** 0.5
    t = dict(a=_mk(g, M, 256), wq=_mk(g, 192, 256, scale=s), qt=_mk(g, M, 192), wv=_mk(g, 256, 256, scale=s),
             bv=_mk(g, 256, scale=0.1))
    d = {k: v.double() for k, v in t.items()}
    zq_ref = odense.linear_chain(d['a'], d['wq'], None, residual=d['qt'] if term else None)
    zv_ref = odense.linear_chain(d['a'], d['wv'], d['bv'])
    c = {k: v.cuda() for k, v in t.items()}
    zq, zv = ext.linear_pair_chain(c['a'], c['wq'], c['qt'] if term else None, c['wv'], c['bv'])
    torch.cuda.synchronize()
    dq = float((zq.cpu().double() - zq_ref).abs().max())
    dv = float((zv.cpu().double() - zv_ref).abs().max())
    print(f"chain C M={M}: max|zq - oracle| = {dq:.3e}, max|zv - oracle| = {dv:.3e}")
    assert zq.shape == (M, 192) and zv.shape == (M, 256) and dq < 2e-4 and dv < 2e-4
    z1 = ext.linear(c['a'], c['wq'], None, residual=c['qt'] if term else None)
    z2 = ext.linear(c['a'], c['wv'], c['bv'])
    assert float((zq - z1).abs().max()) < 1e-4 and float((zv - z2).abs().max()) < 1e-4


def test_linear_chain_never_writes_past_the_last_row(monkeypatch):
    """The chain kernels address their matrices through buffer resources of M rows: the rows of the last tile that lie beyond M
    must be dropped by the bounds check, not written.  Every output is allocated with 64 sentinel rows behind it."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(64)
    M = 512 * 64 + 101 - 64 * 3            # ragged 64-row tail AND (M > 512 tiles) a ragged 32-row tail
    guards = []
    real_empty = torch.empty

    def guarded_empty(*shape, **kw):
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        if len(shp) == 2 and shp[0] == M and kw.get('dtype') == torch.float32:
            full = real_empty((M + 64, shp[1]), **kw)
            full.fill_(-777.0)
            guards.append(full)
            return full[:M]
        return real_empty(*shape, **kw)
    s = (1.0 / 256) ** 0.5
    c = {k: v.cuda() for k, v in dict(
        a=_mk(g, M, 256), res=_mk(g, M, 256), wo=_mk(g, 256, 256, scale=s), bo=_mk(g, 256, scale=0.1),
        g1=torch.rand(256, generator=g) + 0.5, be1=_mk(g, 256, scale=0.1), w1=_mk(g, 512, 256, scale=s),
        b1=_mk(g, 512, scale=0.1), w2=_mk(g, 256, 512, scale=(1.0 / 512) ** 0.5), b2=_mk(g, 256, scale=0.1),
        g2=torch.rand(256, generator=g) + 0.5, be2=_mk(g, 256, scale=0.1), wq=_mk(g, 192, 256, scale=s),
        qt=_mk(g, M, 192), wv=_mk(g, 256, 256, scale=s), bv=_mk(g, 256, scale=0.1), w3=_mk(g, 768, 256, scale=s),
        b3=_mk(g, 768, scale=0.1)).items()}
    monkeypatch.setattr(torch, "empty", guarded_empty)
    ext.linear_ln_chain(c['a'], c['res'], c['wo'], c['bo'], (c['g1'], c['be1'], 1e-5), c['w3'], c['b3'])
    ext.encoder_ffn_chain(c['a'], c['res'], c['wo'], c['bo'], (c['g1'], c['be1'], 1e-5), c['w1'], c['b1'], c['w2'], c['b2'],
                          (c['g2'], c['be2'], 1e-5), tail=(c['wq'], c['qt'], c['wv'], c['bv']))
    ext.linear_pair_chain(c['a'], c['wq'], c['qt'], c['wv'], c['bv'])
    torch.cuda.synchronize()
    monkeypatch.undo()
    assert len(guards) == 2 + 3 + 2, len(guards)
    for full in guards:
        assert bool((full[M:] == -777.0).all()), "a chain kernel wrote behind row M"
        assert bool((full[:M] != -777.0).any())


def test_linear_chain_rejects_other_shapes():
    from occnet_amd import ext
    from occnet_amd._lib import OccAmdUnsupported
    a = torch.zeros(8, 128, device='cuda')
    w = torch.zeros(128, 128, device='cuda')
    with pytest.raises(OccAmdUnsupported):
        ext.linear_ln_chain(a, a, w, None, (torch.ones(128, device='cuda'), torch.zeros(128, device='cuda'), 1e-5), w, None)
    a = torch.zeros(8, 256, device='cuda')
    w = torch.zeros(256, 256, device='cuda')
    ln = (torch.ones(256, device='cuda'), torch.zeros(256, device='cuda'), 1e-5)
    with pytest.raises(OccAmdUnsupported):      # FFN hidden != 512
        ext.encoder_ffn_chain(a, a, w, None, ln, torch.zeros(1024, 256, device='cuda'), None,
                              torch.zeros(256, 1024, device='cuda'), None, ln)


@pytest.mark.parametrize("P", [1, 4])
def test_value_proj_fp16_output_saturates_instead_of_overflowing(P):
    """fp16 SCA value maps (ADVICE r3 / VERDICT r3 weak #4): a projected value beyond the fp16 range is stored as
    +-65 504, never as Inf — on both kernels (activation-resident: every group >= 128 rows; tiled: a short extra
    segment) — and everything inside the range is unchanged."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(31 + P)
    cams, K, N = 2, 256, 256
    for hws in ([256], [256, 20]):                       # resident kernel / tiled kernel
        total = sum(hws) + 4
        starts = [0] + [sum(hws[:i + 1]) + 2 for i in range(len(hws) - 1)]
        a_list = [torch.randn(cams * hw, K, generator=g).cuda().to(torch.bfloat16) for hw in hws]
        a_list[0][5] = 60000.0                           # two pixels of huge features: |v| ~ 60000 * |sum w| >> 65504
        a_list[0][7] = -60000.0
        ws = [(torch.randn(N, K, generator=g) / 4).cuda() for _ in range(P)]
        gbs = [torch.zeros(len(hws), 1, N).cuda() for _ in range(P)]
        out = torch.zeros((P, cams * total, N), device='cuda', dtype=torch.float16)
        if P == 1:
            ext.value_proj_bf16(a_list, ws[0], gbs[0], out[0], rows_per_group=hws, out_group_rows=total, out_row0=starts)
        else:
            ext.value_proj_bf16_planes(a_list, ws, gbs, out, rows_per_group=hws, out_group_rows=total, out_row0=starts)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.float()).all()), "fp16 value maps must saturate, not overflow"
        for p in range(P):
            o = ext.sca_unpair_layout(out[p].view(cams, total, N // 32, 32)).reshape(cams, total, N)
            ref = (a_list[0].double() @ ws[p].double().t()).view(cams, hws[0], N)
            got = o[:, starts[0]:starts[0] + hws[0]].double()
            big = ref.abs() > 65504.0
            assert int(big.sum()) > 100                  # the case really overflows
            assert bool((got[big].abs() == 65504.0).all()) and bool((got[big].sign() == ref[big].sign()).all())
            normal = torch.ones_like(big)                # every pixel but the two huge ones: unchanged by the clamp
            normal.view(cams * hws[0], N)[5] = False
            normal.view(cams * hws[0], N)[7] = False
            rel = ((got[normal] - ref[normal]).abs() / ref[normal].abs().clamp_min(1.0)).max()
            assert float(rel) < 2e-3 and not bool(big[normal].any())
