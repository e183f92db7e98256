"""-m gpu: parity at the FULL sizes of BASELINE.json configs[1] and configs[4] (missing #1 / weak #1).

Everything else in tests/ compares the HIP path with the oracle on reduced geometries (tests/util.small_cfg);
these cases run the geometry the bench times — 200x200 BEV queries, 6 cameras, FPN maps 116x200 / 58x100 /
29x50 / 15x25 (30 825 keys per camera), 16 voxels per pillar — against the CPU oracle (oracle/model.py, torch fp32;
oracle/msda.py in float64 for the backward), reference shapes from
projects/configs/bevformer/bevformer_base_occ.py:36-42,91,103-121:

* one encoder layer + lifter + Conv3d decoder + heads, fp32 FPN maps (the reference's input format);
* the bench's hot-path configuration: 4 encoder layers, bf16 NHWC maps (what the backbone plan emits);
* the deformable-attention backward on one full SCA call (6 x max_len padded rows, the sampling pattern the
  model really produces), against float64 autograd, camera by camera;
* configs[4] (400x400x32) at full size against the oracle — no longer against the repo's own library-op path;
* images -> voxels with an fp32 backbone (stock torch modules on the GPU vs the same modules on the CPU + oracle).
The oracle needs 5-40 s per case on the GPU box's host cores.
"""
import copy
import os

import pytest
import torch

from occnet_amd import synthetic
from tests.util import TOL, build_pair, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _sane_threads():
    """The oracle's torch CPU ops scale badly past ~32 threads (the 256-thread default of the GPU box is 8x
    slower than 32, BENCH_r01 vs bench.py's thread sweep)."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    yield
    torch.set_num_threads(n)


def _base(num_layers, **kw):
    return dict(synthetic.BASE, num_points=8, num_layers=num_layers, **kw)


def _check(out_p, out_o, what, keys=('bev_embed', 'occ', 'flow'), tol=TOL):
    for k in keys:
        assert out_p[k].shape == out_o[k].shape, (k, out_p[k].shape, out_o[k].shape)
        d = maxdiff(out_p[k], out_o[k])
        print(f"{what} {k} {tuple(out_p[k].shape)}: max|hip - oracle| = {d:.3e}")
        assert d < tol, (what, k, d)


def test_base_geometry_one_layer_fp32_features():
    g = _base(1)
    prod, ora = build_pair(g, seed=11)
    feats = synthetic.make_features(g, seed=11)
    metas = synthetic.make_img_metas(g)
    with torch.no_grad():
        out_o = ora(feats, metas)
        out_p = prod([f.cuda() for f in feats], metas)
        occ_p, flow_p = prod.get_occ(out_p, metas)
        occ_o, flow_o = ora.get_occ(out_o, metas)
    assert out_p['occ'].shape == (1, 200, 200, 16, 17) and out_p['flow'].shape == (1, 200, 200, 16, 2)
    _check(out_p, out_o, "base 1 layer fp32")
    # decode: argmax may only differ where the two top logits are closer than the parity bound
    diff = occ_p.cpu() != occ_o
    if bool(diff.any()):
        top2 = out_o['occ'].topk(2, -1).values
        gap = (top2[..., 0] - top2[..., 1])[diff]
        assert float(gap.max()) < 2 * TOL, float(gap.max())
    print(f"argmax decode: {int(diff.sum())} of {diff.numel()} voxels differ (all inside the tie margin)")


def test_base_geometry_four_layers_bf16_nhwc_features():
    """The bench's `--scope hotpath` configuration: LazyFeatures + value_proj straight off the bf16 NHWC maps,
    sca gather at Nq = 40 000 in 8x8 tile order, strided (B, 40 000, 768) Linear views, 4 layers."""
    g = _base(4)
    prod, ora = build_pair(g, seed=12)
    feats = [f.to(torch.bfloat16) for f in synthetic.make_features(g, seed=12)]

    def nhwc(f):
        B, N, C, h, w = f.shape
        return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)
    metas = synthetic.make_img_metas(g)
    with torch.no_grad():
        out_o = ora([f.float() for f in feats], metas)          # the same bf16-representable numbers in fp32
        out_p = prod([nhwc(f) for f in feats], metas)
    _check(out_p, out_o, "base 4 layers bf16-NHWC")


def test_full_size_sca_backward_matches_f64_autograd():
    """ms_deform_attn_backward on one full SpatialCrossAttention call of the base config: value (6, 30 825, 8, 32),
    6 x max_len (~9 900) padded query rows, locations / weights as the model produces them (captured from the
    oracle's own SCA call).  Checked camera by camera against float64 autograd through the restated mmcv op
    (reference seam: multi_scale_deformable_attn_function.py:130-163)."""
    import oracle.model as om
    from oracle import msda as omsda
    from occnet_amd import ext
    from tests.test_gpu_backward import _interior
    g = _base(1)
    _, ora = build_pair(g, seed=13, device='cpu')
    feats = synthetic.make_features(g, seed=13)
    metas = synthetic.make_img_metas(g)
    calls = []
    orig = om.multi_scale_deformable_attn_pytorch
    om.multi_scale_deformable_attn_pytorch = lambda *a: (calls.append(a), orig(*a))[1]
    try:
        with torch.no_grad():
            ora(feats, metas, only_bev=True)
    finally:
        om.multi_scale_deformable_attn_pytorch = orig
    value, shapes_t, loc, attn = calls[1]                       # call 0 = TSA, call 1 = SCA
    assert value.shape == (6, 30825, 8, 32) and loc.shape[2:] == (8, 4, 8, 2) and loc.shape[1] > 9000
    shapes = [tuple(int(v) for v in r) for r in shapes_t.tolist()]
    loc = _interior(loc, shapes)
    start = torch.cat([shapes_t.new_zeros(1), (shapes_t[:, 0] * shapes_t[:, 1]).cumsum(0)[:-1]])
    grad_out = torch.randn(loc.shape[0], loc.shape[1], 256, generator=torch.Generator().manual_seed(14))
    gv, gl, ga = (torch.zeros_like(t).cuda() for t in (value, loc, attn))
    ext.ms_deform_attn_backward(value.cuda(), shapes_t.cuda(), start.cuda(), loc.cuda(), attn.cuda(),
                                grad_out.cuda(), gv, gl, ga, im2col_step=64)
    torch.cuda.synchronize()
    gv, gl, ga = gv.cpu(), gl.cpu(), ga.cpu()
    worst = {}
    for c in range(value.shape[0]):
        s = slice(c, c + 1)
        gv_r, gl_r, ga_r = omsda.msda_backward_autograd(value[s].double(), shapes_t, loc[s].double(),
                                                        attn[s].double(), grad_out[s].double())
        for nm, got, ref in (("grad_value", gv[s], gv_r), ("grad_loc", gl[s], gl_r), ("grad_attn", ga[s], ga_r)):
            scale = max(1.0, float(ref.abs().max()))
            d = float((got.double() - ref).abs().max()) / scale
            worst[nm] = max(worst.get(nm, 0.0), d)
    print("full-size SCA backward, max rel diff over 6 cameras:", {k: f"{v:.3e}" for k, v in worst.items()})
    assert all(v < 1e-4 for v in worst.values()), worst


def test_hires_full_size_matches_oracle():
    """configs[4]: 400x400x32 (160 000 queries, 5.12 M voxels, 8 decoder input channels) at FULL size, one
    encoder layer, against the oracle."""
    g = dict(synthetic.HIRES, num_points=8, num_layers=1)
    prod, ora = build_pair(g, seed=15)
    feats = synthetic.make_features(g, seed=15)
    metas = synthetic.make_img_metas(g)
    with torch.no_grad():
        out_o = ora(feats, metas)
        out_p = prod([f.cuda() for f in feats], metas)
    assert out_p['occ'].shape == (1, 400, 400, 32, 17) and out_p['flow'].shape == (1, 400, 400, 32, 2)
    _check(out_p, out_o, "hires 400x400x32 1 layer")


def test_images_to_voxels_fp32_backbone():
    """images -> voxels with an fp32 backbone: the detector on the GPU (stock ResNet-50 + FPN modules, fp32,
    + the HIP hot path) vs the same backbone modules on the CPU feeding the oracle head.  Random-init ResNet
    activations are large, so the bound is relative: 1e-3 of the output's scale."""
    from occnet_amd.plugin import Config, build_model, import_plugin
    import oracle.model as om
    from tests.util import head_cfg, randomize
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'occ_base_200x200x16.py'))
    cfg.merge_from_dict({'model.pts_bbox_head.transformer.encoder.num_layers': 1})
    import_plugin(cfg)
    torch.manual_seed(0)
    model = build_model(cfg.model)
    model.init_weights()
    randomize(model.pts_bbox_head, 16)
    model.eval()
    g = _base(1)
    ocfg = head_cfg(g)
    ocfg.pop('type')
    ora = om.BEVFormerOccHead(**ocfg).eval()
    ora.load_state_dict(model.pts_bbox_head.state_dict(), strict=True)
    img = synthetic.make_images(g, batch=1, seed=16)
    metas = synthetic.make_img_metas(g)
    with torch.no_grad():
        feats_cpu = model.extract_feat(img=img, img_metas=metas)              # stock torch ops, CPU, fp32
        out_o = ora(feats_cpu, metas)
        gpu = copy.deepcopy(model).cuda()
        feats_gpu = gpu.extract_feat(img=img.cuda(), img_metas=metas)
        out_p = gpu.pts_bbox_head(feats_gpu, metas)
    fs = max(float(f.abs().max()) for f in feats_cpu)
    fd = max(maxdiff(a, b) for a, b in zip(feats_gpu, feats_cpu))
    print(f"FPN maps: scale {fs:.3e}, max|gpu - cpu| = {fd:.3e} (rel {fd / fs:.2e})")
    for k in ('bev_embed', 'occ', 'flow'):
        scale = max(1.0, float(out_o[k].abs().max()))
        d = maxdiff(out_p[k], out_o[k])
        print(f"images->voxels fp32 backbone {k}: max|hip - oracle| = {d:.3e} (scale {scale:.2f}, rel {d / scale:.2e})")
        assert d / scale < TOL, (k, d, scale)


# ---- pinned to the reference at the benchmarked geometry (VERDICT r2 missing #2, #3) -------------------------------
@pytest.mark.parametrize('name', ['base_full_nohist', 'base_full_hist', 'base_full_4layer', 'base_full_hist_4layer',
                                  'hires_full_4layer'])
def test_product_matches_reference_golden_at_base_geometry(name):
    """HIP head vs tests/golden/{base,hires}_full_*.npz: digests of what the reference's OWN module files produced at
    40 000 queries / 6 x 30 825 keys / max_len ~ 9 900 / 106 camera-less queries, one layer, without and with a
    history BEV rotated by 7.5 degrees (oracle/gen_golden.py::fullsize_golden; reference
    spatial_cross_attention.py:136-173, transformer_occ.py:189-205, temporal_self_attention.py:177-204) and, at the
    benchmarked depth of FOUR layers (bevformer_base_occ.py:103): BASELINE configs[1] (base_full_4layer), configs[2]'s
    history branch (base_full_hist_4layer: every layer's TSA attends to the rotated history BEV) and configs[4]
    (hires_full_4layer: 400 x 400 x 32 grid, 160 000 queries — VERDICT r4 item 7).  The digest covers every element
    (sums of 512-element chunks) besides the strided subsample.  No oracle in the loop."""
    import numpy as np
    from occnet_amd.plugin import build_head
    from tests.golden_cases import FULL_CASES, FULL_KEYS, checksum, compare_digest, full_case_inputs
    from tests.util import head_cfg, randomize
    case = FULL_CASES[name]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'{name}.npz'))
    head = build_head(head_cfg(case['geometry']))
    randomize(head, case['seed'])
    assert abs(checksum(head.state_dict().values()) / float(gold['weights_checksum']) - 1) < 1e-9
    head = head.cuda().eval()
    feats, metas, prev_bev = full_case_inputs(case)
    assert abs(checksum(feats) / float(gold['inputs_checksum']) - 1) < 1e-9
    with torch.no_grad():
        out = head([f.cuda() for f in feats], metas, prev_bev=None if prev_bev is None else prev_bev.cuda())
    for k in ('bev_embed', 'occ', 'flow'):
        assert tuple(out[k].shape) == tuple(int(v) for v in gold[f'{k}_shape'])
        sub, slab = compare_digest(k, out[k], gold, TOL)
        print(f"reference golden {name}/{k}: subsample max|hip - reference| = {sub:.3e}, slab mean diff {slab:.2e}")


def test_base_geometry_with_history():
    """BASELINE.json configs[2] at FULL size: BEVFormerOcc.obtain_history_bev over a 3-frame queue (reference
    bevformer_occ.py:159-178) followed by the current frame's full head pass with that history — TSA with a real,
    rotated prev_bev (value_bt_stride != 0, two K segments) at 200 x 200 — against the oracle driven through the same
    chain, 2 encoder layers."""
    from occnet_amd.plugin import BEVFormerOcc
    g = _base(2)
    prod, ora = build_pair(g, seed=17)
    det = BEVFormerOcc.__new__(BEVFormerOcc)
    torch.nn.Module.__init__(det)
    det.pts_bbox_head = prod
    det.video_test_mode = True
    det.eval()             # obtain_history_bev restores the mode it found: a bare nn.Module starts in training mode
    L = 3
    frames = [synthetic.make_features(g, seed=170 + i) for i in range(L + 1)]
    metas_list = [[]]
    for i in range(L + 1):
        m = synthetic.make_img_metas(g, seed=i)[0]
        m['prev_bev_exists'] = i != 1          # frame 1 starts a new scene: frame 0's BEV is dropped
        m['can_bus'][-1] = 2.5 * i - 3.0       # ego yaw change since the previous frame (degrees)
        metas_list[0].append(m)
    stacked = [torch.stack([frames[i][l][0] for i in range(L)], 0)[None].cuda() for l in range(len(frames[0]))]
    det.extract_feat = lambda img, img_metas=None, len_queue=None: stacked
    imgs = torch.zeros(1, L, g['num_cams'], 3, 8, 8, device='cuda')
    prev_p = det.obtain_history_bev(imgs, [metas_list[0][:L]])
    cur = metas_list[0][L]
    with torch.no_grad():
        out_p = prod([f.cuda() for f in frames[L]], [cur], prev_bev=prev_p)
        prev = None
        for i in range(L):
            if not metas_list[0][i]['prev_bev_exists']:
                prev = None
            prev = ora(frames[i], [metas_list[0][i]], prev, only_bev=True)
        d = maxdiff(prev_p, prev)
        print(f"3-frame history BEV at 200x200: max|hip - oracle| = {d:.3e}")
        assert d < TOL
        out_o = ora(frames[L], [cur], prev_bev=prev)
    _check(out_p, out_o, "base 2 layers + 3-frame history")
