"""-m gpu: HIP `ms_deform_attn_backward` (through the C ABI and the autograd Function) vs the oracle
(torch.autograd through the restated mmcv CPU function, float64)."""
import pytest
import torch

from oracle import msda as omsda
from tests.test_gpu_msda import _inputs

pytestmark = pytest.mark.gpu

CASES = [
    ("sca_like", 2, [[12, 20], [6, 10], [3, 5], [2, 3]], 8, 32, 150, 8),
    ("tsa_like", 2, [[20, 20]], 8, 32, 400, 4),
    ("ragged_items", 1, [[5, 7], [3, 4]], 3, 32, 13, 5),
    ("generic_d16", 2, [[9, 11], [4, 6]], 4, 16, 57, 3),
    ("generic_d24", 1, [[6, 5]], 2, 24, 11, 2),
    ("single", 1, [[1, 1]], 1, 32, 1, 1),
]


def _interior(loc, shapes):
    """Move sampling locations off the non-differentiable set (integer pixel coordinates and the
    admission boundary), where one-sided derivatives of kernel and autograd may legitimately differ."""
    loc = loc.clone()
    for l, (H, W) in enumerate(shapes):
        for d, n in ((0, W), (1, H)):
            x = loc[:, :, :, l, :, d] * n - 0.5
            frac = x - torch.floor(x)
            x = torch.where((frac < 0.05) | (frac > 0.95), torch.floor(x) + 0.5, x)
            loc[:, :, :, l, :, d] = (x + 0.5) / n
    return loc


@pytest.mark.parametrize("name,B,shapes,M,D,Lq,P", CASES, ids=[c[0] for c in CASES])
def test_backward_matches_autograd_oracle(name, B, shapes, M, D, Lq, P):
    from occnet_amd import ext
    value, shapes_t, start, loc, attn = _inputs(B, shapes, M, D, Lq, P, seed=5, adversarial=False)
    loc = _interior(loc, shapes)
    g = torch.Generator().manual_seed(6)
    grad_out = torch.randn(B, Lq, M * D, generator=g)
    gv_ref, gl_ref, ga_ref = omsda.msda_backward_autograd(
        value.double(), shapes_t, loc.double(), attn.double(), grad_out.double())
    gv = torch.zeros_like(value).cuda()
    gl = torch.zeros_like(loc).cuda()
    ga = torch.zeros_like(attn).cuda()
    ext.ms_deform_attn_backward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(), attn.cuda(),
                                grad_out.cuda(), gv, gl, ga, im2col_step=64)
    torch.cuda.synchronize()
    for nm, got, ref in (("grad_value", gv, gv_ref), ("grad_loc", gl, gl_ref), ("grad_attn", ga, ga_ref)):
        scale = max(1.0, float(ref.abs().max()))
        d = float((got.cpu().double() - ref).abs().max()) / scale
        print(f"{name} {nm}: max rel diff = {d:.3e} (scale {scale:.2f})")
        assert d < 1e-4, (name, nm, d)


def test_out_of_range_samples_have_zero_gradient():
    from occnet_amd import ext
    value, shapes_t, start, loc, attn = _inputs(1, [[6, 8]], 8, 32, 40, 4, seed=7, adversarial=False)
    loc[:, :20] = 5.0       # far outside: fail the admission test
    grad_out = torch.ones(1, 40, 8 * 32)
    gv, gl, ga = (torch.zeros_like(t).cuda() for t in (value, loc, attn))
    ext.ms_deform_attn_backward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(), attn.cuda(),
                                grad_out.cuda(), gv, gl, ga, im2col_step=64)
    assert float(gl[:, :20].abs().max()) == 0.0 and float(ga[:, :20].abs().max()) == 0.0
    assert float(ga[:, 20:].abs().max()) > 0.0


def test_autograd_function_round_trip():
    """MultiScaleDeformableAttnFunction_fp32.apply(...).backward() — the reference's call shape
    (multi_scale_deformable_attn_function.py:90-163)."""
    from occnet_amd.plugin.functions import MultiScaleDeformableAttnFunction_fp32 as Fn
    shapes = [[10, 14], [5, 7]]
    value, shapes_t, start, loc, attn = _inputs(2, shapes, 8, 32, 64, 4, seed=8, adversarial=False)
    loc = _interior(loc, shapes)
    v = value.cuda().requires_grad_(True)
    l = loc.cuda().requires_grad_(True)
    a = attn.cuda().requires_grad_(True)
    out = Fn.apply(v, shapes_t.cuda(), start.cuda(), l, a, 64)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).cuda()
    (out * w).sum().backward()
    gv_ref, gl_ref, ga_ref = omsda.msda_backward_autograd(
        value.double(), shapes_t, loc.double(), attn.double(), w.cpu().double())
    for got, ref in ((v.grad, gv_ref), (l.grad, gl_ref), (a.grad, ga_ref)):
        scale = max(1.0, float(ref.abs().max()))
        assert float((got.cpu().double() - ref).abs().max()) / scale < 1e-4


def test_integration_3a_seam_reference_call_shape():
    """INTEGRATION.md §3a, executed: the reference obtains its operator with
    `from mmcv.utils import ext_loader; ext_module = ext_loader.load_ext('_ext', ['ms_deform_attn_backward',
    'ms_deform_attn_forward'])` (multi_scale_deformable_attn_function.py:8-12) and calls it as written at
    :118-124 (forward: five positional tensors + `im2col_step=` keyword) and :146-160 (backward: zeros_like grad
    buffers, `grad_output.contiguous()`, writes in place, returns None).  With `mmcv.utils.ext_loader` resolved
    to `occnet_amd.ext_loader` — the 2-line swap — that exact call sequence must run on the HIP library and
    match the oracle.  (The call sequence below restates those reference lines; the reference file itself
    cannot travel to the GPU box.)"""
    import sys
    import types
    from occnet_amd import ext_loader as occ_loader
    saved = {k: sys.modules.get(k) for k in ('mmcv', 'mmcv.utils')}
    mmcv = types.ModuleType('mmcv')
    mmcv.utils = types.ModuleType('mmcv.utils')
    mmcv.utils.ext_loader = occ_loader
    sys.modules['mmcv'], sys.modules['mmcv.utils'] = mmcv, mmcv.utils
    try:
        from mmcv.utils import ext_loader                                            # reference :8
        ext_module = ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])  # :11-12
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    shapes = [[12, 20], [6, 10], [3, 5], [2, 3]]
    value, shapes_t, start, loc, attn = _inputs(2, shapes, 8, 32, 96, 8, seed=21, adversarial=False)
    loc = _interior(loc, shapes)
    value, shapes_t, start, loc, attn = (t.cuda() for t in (value, shapes_t, start, loc, attn))
    im2col_step = 64
    output = ext_module.ms_deform_attn_forward(value, shapes_t, start, loc, attn, im2col_step=im2col_step)
    ref = omsda.multi_scale_deformable_attn_pytorch(value.cpu().double(), shapes_t.cpu(), loc.cpu().double(),
                                                    attn.cpu().double())
    assert float((output.cpu().double() - ref).abs().max()) < 1e-5
    grad_output = torch.randn(output.shape, generator=torch.Generator().manual_seed(22)).cuda()
    grad_value = torch.zeros_like(value)                                             # reference :146-148
    grad_sampling_loc = torch.zeros_like(loc)
    grad_attn_weight = torch.zeros_like(attn)
    ret = ext_module.ms_deform_attn_backward(value, shapes_t, start, loc, attn, grad_output.contiguous(),
                                             grad_value, grad_sampling_loc, grad_attn_weight,
                                             im2col_step=im2col_step)               # :150-160
    assert ret is None
    gv_r, gl_r, ga_r = omsda.msda_backward_autograd(value.cpu().double(), shapes_t.cpu(), loc.cpu().double(),
                                                    attn.cpu().double(), grad_output.cpu().double())
    for got, want in ((grad_value, gv_r), (grad_sampling_loc, gl_r), (grad_attn_weight, ga_r)):
        scale = max(1.0, float(want.abs().max()))
        assert float((got.cpu().double() - want).abs().max()) / scale < 1e-4


@pytest.mark.parametrize("name,B,shapes,M,Lq,P", [
    ("sca_like", 2, [[12, 20], [6, 10], [3, 5], [2, 3]], 8, 3000, 8),       # coarse levels: a few very hot bins
    ("tsa_like", 1, [[40, 40]], 8, 1600, 4),
])
def test_deterministic_mode_is_bit_reproducible(name, B, shapes, M, Lq, P, monkeypatch):
    """OCC_MSDA_BWD_DETERMINISTIC=1 (reference contract: plain accumulation into caller-zeroed
    tensors, multi_scale_deformable_attn_function.py:146-163): order-independent fixed-point accumulation of grad_value
    -> repeated launches are BIT-identical (the default path is only reproducible to fp32 summation-order noise), the
    result agrees with the float64 autograd oracle, and grad_loc / grad_attn are the same bits as in the default mode."""
    from occnet_amd import ext
    value, shapes_t, start, loc, attn = _inputs(B, shapes, M, 32, Lq, P, seed=15, adversarial=False)
    loc = _interior(loc, shapes)
    grad_out = torch.randn(B, Lq, M * 32, generator=torch.Generator().manual_seed(16)) * 3.7e-3
    args = [t.cuda() for t in (value, shapes_t, start, loc, attn, grad_out)]

    def run():
        gv, gl, ga = (torch.zeros_like(t).cuda() for t in (value, loc, attn))
        ext.ms_deform_attn_backward(*args, gv, gl, ga, im2col_step=64)
        torch.cuda.synchronize()
        return gv, gl, ga
    monkeypatch.delenv("OCC_MSDA_BWD_DETERMINISTIC", raising=False)
    d_gv, d_gl, d_ga = run()
    monkeypatch.setenv("OCC_MSDA_BWD_DETERMINISTIC", "1")
    runs = [run() for _ in range(4)]
    for gv, gl, ga in runs[1:]:
        assert torch.equal(gv, runs[0][0]) and torch.equal(gl, runs[0][1]) and torch.equal(ga, runs[0][2])
    assert torch.equal(runs[0][1], d_gl) and torch.equal(runs[0][2], d_ga)
    gv_ref, _, _ = omsda.msda_backward_autograd(value.double(), shapes_t, loc.double(), attn.double(), grad_out.double())
    scale = float(gv_ref.abs().max())
    dd = float((runs[0][0].cpu().double() - gv_ref).abs().max()) / scale
    dn = float((d_gv.cpu().double() - gv_ref).abs().max()) / scale
    print(f"{name}: deterministic grad_value rel diff vs float64 {dd:.3e} (default path {dn:.3e}), scale {scale:.3e}")
    assert dd < 1e-5 and dd <= dn * 1.5 + 1e-7


def test_deterministic_mode_propagates_non_finite_gradients_and_refuses_other_head_dims(monkeypatch):
    """(i) an Inf / NaN in grad_output has no fixed-point image — the pixels its samples touch come back NaN
    (as the float path leaves them), everything else is the clean run's bits, never finite garbage; (ii) the mode exists for
    D = 32 only: any other head dim is refused instead of silently running the float-atomic kernel."""
    from occnet_amd import ext
    B, shapes, M, Lq, P = 1, [[12, 20], [6, 10]], 8, 600, 4
    value, shapes_t, start, loc, attn = _inputs(B, shapes, M, 32, Lq, P, seed=25, adversarial=False)
    loc = _interior(loc, shapes)
    grad_out = torch.randn(B, Lq, M * 32, generator=torch.Generator().manual_seed(26)) * 1e-2
    monkeypatch.setenv("OCC_MSDA_BWD_DETERMINISTIC", "1")

    def run(go):
        gv, gl, ga = (torch.zeros_like(t).cuda() for t in (value, loc, attn))
        ext.ms_deform_attn_backward(*[t.cuda() for t in (value, shapes_t, start, loc, attn, go)], gv, gl, ga, im2col_step=64)
        torch.cuda.synchronize()
        return gv.cpu()
    clean = run(grad_out)
    assert torch.isfinite(clean).all()
    for poison in (float('inf'), float('nan')):
        go = grad_out.clone()
        go[0, 17, 3 * 32 + 5] = poison                     # query 17, head 3, channel 5
        got = run(go)
        bad = ~torch.isfinite(got)
        assert bad.any() and torch.isnan(got[bad]).all()                         # NaN, never a finite stand-in or +-Inf
        assert bad.view(B, -1, M, 32)[:, :, [0, 1, 2, 4, 5, 6, 7]].sum() == 0    # only head 3's pixels are touched
        assert int(bad.view(B, -1, M, 32)[0, :, 3].any(-1).sum()) <= len(shapes) * P * 4
        assert torch.equal(got[~bad], clean[~bad])
    v16, s16, st16, l16, a16 = _inputs(B, shapes, M, 16, 64, P, seed=27, adversarial=False)
    go16 = torch.randn(B, 64, M * 16)
    with pytest.raises(ext.OccAmdUnsupported):
        gv, gl, ga = (torch.zeros_like(t).cuda() for t in (v16, l16, a16))
        ext.ms_deform_attn_backward(*[t.cuda() for t in (v16, s16, st16, l16, a16, go16)], gv, gl, ga, im2col_step=64)
