"""-m gpu: the HIP multi-scale deformable attention operator (through the C ABI) vs the oracle."""
import ctypes
import os

import pytest
import torch

from oracle import msda as omsda

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, shapes, M, D, Lq, P, seed, adversarial=True):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.tensor(shapes, dtype=torch.long)
    start = torch.cat([shapes_t.new_zeros(1), shapes_t.prod(1).cumsum(0)[:-1]])
    S = int(shapes_t.prod(1).sum())
    L = len(shapes)
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.4 - 0.2       # some outside [0,1]
    attn = torch.rand(B, Lq, M, L, P, generator=g) + 1e-3
    attn = attn / attn.flatten(-2).sum(-1)[..., None, None]
    if adversarial and Lq >= 8:
        H, W = shapes[0]
        loc[0, 0] = 0.0                                   # exact top-left corner
        loc[0, 1] = 1.0                                   # exact bottom-right corner
        loc[0, 2, ..., 0] = 0.5 / W; loc[0, 2, ..., 1] = 0.5 / H      # a pixel centre
        loc[0, 3] = -1.0 / max(H, W)                      # just outside: admitted (> -1 px), corner clipped
        loc[0, 4] = 1e7                                   # absurdly far (z ~ 1e-5 projections)
        loc[0, 5] = -1e7
        loc[0, 6, ..., 0] = 1.0 - 0.5 / W; loc[0, 6, ..., 1] = 0.5 / H
        loc[0, 7] = 1.0 + 0.49 / max(H, W)                # last admitted half pixel
    return value, shapes_t, start, loc, attn


CASES = [
    # name, B, shapes, M, D, Lq, P
    ("sca_like", 3, [[12, 20], [6, 10], [3, 5], [2, 3]], 8, 32, 301, 8),
    ("tsa_like", 2, [[20, 20]], 8, 32, 400, 4),
    ("ragged_items", 1, [[5, 7], [3, 4]], 3, 32, 13, 5),     # item count not a multiple of 8
    ("scalar_d16", 2, [[9, 11], [4, 6]], 4, 16, 57, 3),      # non-32 head dim -> scalar kernel
    ("single", 1, [[1, 1]], 1, 32, 1, 1),
]


@pytest.mark.parametrize("name,B,shapes,M,D,Lq,P", CASES, ids=[c[0] for c in CASES])
def test_forward_matches_oracle(name, B, shapes, M, D, Lq, P):
    from occnet_amd import ext
    value, shapes_t, start, loc, attn = _inputs(B, shapes, M, D, Lq, P, seed=1)
    ref = omsda.multi_scale_deformable_attn_pytorch(value.double(), shapes_t, loc.double(),
                                                    attn.double()).float()
    out = ext.ms_deform_attn_forward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(),
                                     attn.cuda(), im2col_step=64)
    torch.cuda.synchronize()
    d = float((out.cpu() - ref).abs().max())
    print(f"{name}: max|hip - oracle(f64)| = {d:.3e}")
    assert out.shape == (B, Lq, M * D)
    assert d < 2e-5          # fp32 kernel vs fp64 oracle (north_star bound: 1e-3)


def test_forward_full_size_vs_c_oracle():
    """Base-config SCA shape (6 cameras x 30825 keys, 4 levels x 8 points) on 1500 rows per camera,
    against the plain-C restatement (oracle/msda_ref.c)."""
    from occnet_amd import ext
    so = os.path.join(ROOT, "oracle", "_build", "libmsda_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_build/libmsda_ref.so not built (make -C oracle)")
    lib = ctypes.CDLL(so)
    shapes = [[116, 200], [58, 100], [29, 50], [15, 25]]
    B, M, D, Lq, P = 6, 8, 32, 1500, 8
    value, shapes_t, start, loc, attn = _inputs(B, shapes, M, D, Lq, P, seed=2)
    ref = torch.empty(B, Lq, M * D)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.msda_forward_ref_f32(p(value), p(shapes_t), p(start), p(loc), p(attn), p(ref), B,
                             value.shape[1], M, D, len(shapes), Lq, P)
    out = ext.ms_deform_attn_forward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(),
                                     attn.cuda(), im2col_step=64).cpu()
    d = float((out - ref).abs().max())
    print(f"full-size: max|hip - C oracle| = {d:.3e}")
    assert d < 2e-5


def test_linearity_and_zero_weight():
    """Size-independent properties: linear in value and in attention weights; zero weights -> 0."""
    from occnet_amd import ext
    value, shapes_t, start, loc, attn = _inputs(2, [[30, 40], [15, 20]], 8, 32, 2000, 4, seed=3)
    args = (shapes_t.cuda(), start.cuda(), loc.cuda())
    f = lambda v, a: ext.ms_deform_attn_forward(v, args[0], args[1], args[2], a, im2col_step=64)
    v, a = value.cuda(), attn.cuda()
    o1 = f(v, a)
    o2 = f(v * 2.0, a)
    o3 = f(v, a * 0.5)
    assert torch.allclose(o2, o1 * 2.0, atol=1e-5, rtol=1e-5)
    assert torch.allclose(o3, o1 * 0.5, atol=1e-5, rtol=1e-5)
    assert float(f(v, torch.zeros_like(a)).abs().max()) == 0.0


def test_errors_are_raised():
    from occnet_amd import ext
    from occnet_amd._lib import OccAmdError
    value, shapes_t, start, loc, attn = _inputs(3, [[4, 4]], 8, 32, 9, 2, seed=4, adversarial=False)
    with pytest.raises(OccAmdError):     # host tensor
        ext.ms_deform_attn_forward(value, shapes_t, start, loc, attn, im2col_step=64)
    with pytest.raises(OccAmdError):     # batch 3 not divisible by im2col_step 2
        ext.ms_deform_attn_forward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(),
                                   attn.cuda(), im2col_step=2)
    with pytest.raises(OccAmdError):     # inconsistent shapes
        ext.ms_deform_attn_forward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda()[:, :5],
                                   attn.cuda(), im2col_step=64)


def test_generic_forward_never_reads_corners_outside_the_map():
    """ADVICE r1 (low) for the generic operator: a corner outside its map must not be loaded at all (round 1 loaded
    row 0 of the batch entry with weight 0: 0 * Inf = NaN).  Non-finite values at pixel 0 of level 0 may only show
    up in rows that really sample that pixel; every other row is bit-identical to the clean run."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(21)
    B, M, D, Lq, P = 2, 8, 32, 333, 4
    shapes = torch.tensor([[12, 20], [6, 10], [3, 5]])
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15            # a good share of corners off the maps
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    args = (shapes.cuda(), start.cuda(), loc.cuda(), attn.cuda())
    clean = ext.ms_deform_attn_forward(value.cuda(), *args, im2col_step=64)
    value[:, 0] = float('inf')
    value[:, 0, :, ::2] = float('nan')
    dirty = ext.ms_deform_attn_forward(value.cuda(), *args, im2col_step=64)
    touched = ~torch.isfinite(dirty.view(B, Lq, M, D)).all(-1)
    # rows whose level-0 samples have pixel (0, 0) among their in-map corners: h_im, w_im in (-1, 1)
    H0, W0 = int(shapes[0, 0]), int(shapes[0, 1])
    hx = loc[..., 0, :, 0] * W0 - 0.5
    hy = loc[..., 0, :, 1] * H0 - 0.5
    may = ((hx > -1) & (hx < 1) & (hy > -1) & (hy < 1)).any(-1)            # (B, Lq, M)
    assert not bool((touched.cpu() & ~may).any()), "a row that cannot reach pixel (0,0) became non-finite"
    assert 0.0 < float(touched.float().mean()) < 0.5
    assert torch.equal(dirty.view(B, Lq, M, D)[~touched], clean.view(B, Lq, M, D)[~touched])


def test_tsa_fused_never_reads_corners_outside_the_map():
    """Same property for the fused temporal self-attention gather (its BEV value maps)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(22)
    B, M, D, P, bh, bw = 1, 8, 32, 4, 12, 14
    Nq = bh * bw
    value = torch.randn(B * 2, Nq, M, D, generator=g)
    offs = torch.randn(B, Nq, M * 2 * P * 2, generator=g) * 4.0
    logits = torch.randn(B, Nq, M * 2 * P, generator=g)
    ref = torch.rand(B * 2, Nq, 1, 2, generator=g)
    args = (offs.cuda(), logits.cuda(), ref.cuda(), bh, bw, M, P)
    clean = ext.tsa_fused_forward(value.cuda(), *args).view(B, Nq, M, D)
    value[:, 0] = float('inf')
    value[:, 0, :, ::2] = float('nan')
    dirty = ext.tsa_fused_forward(value.cuda(), *args).view(B, Nq, M, D)
    touched = ~torch.isfinite(dirty).all(-1)
    assert 0.0 < float(touched.float().mean()) < 0.6
    assert torch.equal(dirty[~touched], clean[~touched])
    # rows that cannot reach pixel (0, 0): all of a head's 8 samples at h_im >= 1 or w_im >= 1
    o = offs.view(B, Nq, M, 2, P, 2)
    lx = (ref.view(B, 2, Nq, 1, 1, 2)[..., 0].permute(0, 2, 3, 1, 4) + o[..., 0] / bw) * bw - 0.5   # (B,Nq,M,2,P)
    ly = (ref.view(B, 2, Nq, 1, 1, 2)[..., 1].permute(0, 2, 3, 1, 4) + o[..., 1] / bh) * bh - 0.5
    may = ((lx > -1) & (lx < 1) & (ly > -1) & (ly < 1)).flatten(3).any(-1)
    assert not bool((touched.cpu() & ~may).any())
