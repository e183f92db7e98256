"""not-gpu: the CPU oracle (oracle/model.py) against the golden vectors produced by running the
reference's own module files (oracle/gen_golden.py).  This is what pins the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from tests.golden_cases import CASES, case_inputs, checksum
from tests.util import head_cfg, randomize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _oracle_head(case):
    import oracle.model as om
    cfg = head_cfg(case['geometry'])
    cfg.pop('type')
    head = om.BEVFormerOccHead(**copy.deepcopy(cfg))
    randomize(head, case['seed'])
    return head.eval()


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_matches_reference_golden(name):
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, f'{name}.npz'))
    head = _oracle_head(case)
    feats, metas, prev_bev = case_inputs(case)
    # same seeded problem as the generator saw (guards against RNG-stream drift)
    assert abs(checksum(head.state_dict().values()) - float(gold['weights_checksum'])) < 1e-3
    assert abs(checksum(feats) - float(gold['inputs_checksum'])) < 1e-3
    taps = {}
    layer0 = head.transformer.encoder.layers[0]
    layer0.attentions[0].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_tsa_out', o))
    layer0.attentions[1].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_sca_out', o))
    with torch.no_grad():
        out = head(feats, metas, prev_bev=prev_bev)
    for k in ('bev_embed', 'occ', 'flow'):
        d = float(np.abs(out[k].numpy() - gold[k]).max())
        assert out[k].shape == gold[k].shape
        assert d < 1e-5, f'{name}/{k}: oracle differs from the reference golden by {d}'
    for k, v in taps.items():
        d = float(np.abs(v.numpy() - gold[k]).max())
        assert d < 1e-5, f'{name}/{k}: {d}'


# ---- the benchmarked geometry (BASELINE.json configs[1] / configs[2]) --------------------------------------------
from tests.golden_cases import FULL_CASES, FULL_KEYS, compare_digest, full_case_inputs  # noqa: E402


@pytest.mark.parametrize('name', sorted(FULL_CASES))
def test_oracle_matches_reference_golden_at_base_geometry(name):
    """oracle/model.py == the reference's own files at 40 000 queries / 6 x 30 825 keys / max_len ~ 9 900, one
    layer, without and with a rotated history BEV, and at the benchmarked depth of four layers for BASELINE configs[1],
    configs[2] (history) and configs[4] (400 x 400 x 32: 160 000 queries) — fixtures: oracle/gen_golden.py::fullsize_golden."""
    case = FULL_CASES[name]
    gold = np.load(os.path.join(GOLD, f'{name}.npz'))
    head = _oracle_head(case)
    feats, metas, prev_bev = full_case_inputs(case)
    assert abs(checksum(head.state_dict().values()) / float(gold['weights_checksum']) - 1) < 1e-9
    assert abs(checksum(feats) / float(gold['inputs_checksum']) - 1) < 1e-9
    taps = {}
    layer0 = head.transformer.encoder.layers[0]
    layer0.attentions[0].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_tsa_out', o))
    layer0.attentions[1].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_sca_out', o))
    with torch.no_grad():
        out = dict(head(feats, metas, prev_bev=prev_bev))
    out.update(taps)
    for k in FULL_KEYS:
        assert tuple(out[k].shape) == tuple(int(v) for v in gold[f'{k}_shape'])
        sub, slab = compare_digest(k, out[k], gold, 2e-5)
        print(f'{name}/{k}: subsample max diff {sub:.2e}, slab mean diff {slab:.2e}')


def test_thirdparty_restatements_agree():
    """oracle/thirdparty.py (the leaves refshim runs the reference's files on since round 4) and the product's own
    restatements (occnet_amd.plugin.bricks, occnet_amd.io) are two independent implementations of the same mmcv / mmdet
    / pyquaternion / nuscenes-devkit behaviour: same state_dict keys, bit-identical outputs on seeded inputs."""
    import numpy as np
    import torch
    from occnet_amd import io as pio
    from occnet_amd.plugin import bricks as pb
    from oracle import thirdparty as tp
    g = torch.Generator().manual_seed(5)
    # FFN (mmcv): keys, identity handling, dropout inert in eval
    kw = dict(embed_dims=32, feedforward_channels=64, num_fcs=2, ffn_drop=0.1, act_cfg=dict(type='ReLU', inplace=True))
    a, b = pb.FFN(**kw).eval(), tp.FFN(**kw).eval()
    assert sorted(a.state_dict()) == sorted(b.state_dict()) == ['layers.0.0.bias', 'layers.0.0.weight', 'layers.1.bias',
                                                                'layers.1.weight']
    b.load_state_dict(a.state_dict())
    x, idt = torch.randn(3, 7, 32, generator=g), torch.randn(3, 7, 32, generator=g)
    with torch.no_grad():
        assert torch.equal(a(x), b(x)) and torch.equal(a(x, idt), b(x, idt))
        assert torch.equal(b(x), x + b.layers(x))
    # ConvModule (mmcv) as transformer_occ.py:106-126 builds it: Conv3d + BN3d + ReLU, no conv bias
    kw = dict(kernel_size=3, stride=1, padding=1, bias=False, conv_cfg=dict(type='Conv3d'), norm_cfg=dict(type='BN3d'),
              act_cfg=dict(type='ReLU', inplace=True))
    a, b = pb.ConvModule(4, 6, **kw).eval(), tp.ConvModule(4, 6, **kw).eval()
    assert sorted(a.state_dict()) == sorted(b.state_dict()) and 'conv.bias' not in a.state_dict() \
        and 'bn.running_var' in a.state_dict()
    sd = a.state_dict()
    sd['bn.running_mean'] = torch.randn(6, generator=g) * 0.1
    sd['bn.running_var'] = torch.rand(6, generator=g) + 0.5
    a.load_state_dict(sd), b.load_state_dict(sd)
    v = torch.randn(2, 4, 5, 6, 7, generator=g)
    with torch.no_grad():
        assert torch.equal(a(v), b(v)) and float(b(v).min()) >= 0.0
    # LearnedPositionalEncoding (mmdet)
    a, b = pb.LearnedPositionalEncoding(8, 5, 6), tp.LearnedPositionalEncoding(8, 5, 6)
    b.load_state_dict(a.state_dict())
    m = torch.zeros(2, 5, 6)
    assert torch.equal(a(m), b(m)) and tuple(b(m).shape) == (2, 16, 5, 6)
    assert torch.equal(b(m)[0, :8, 3, 2], b.col_embed.weight[2]) and torch.equal(b(m)[1, 8:, 3, 2], b.row_embed.weight[3])
    # losses (mmdet)
    logits, lab = torch.randn(50, 17, generator=g), torch.randint(0, 17, (50,), generator=g)
    assert torch.equal(pb.CrossEntropyLoss(loss_weight=1.0)(logits, lab), tp.CrossEntropyLoss(loss_weight=1.0)(logits, lab))
    w = torch.rand(50, generator=g)
    assert torch.equal(pb.CrossEntropyLoss()(logits, lab, weight=w, avg_factor=13.0),
                       tp.CrossEntropyLoss()(logits, lab, weight=w, avg_factor=13.0))
    p, t = torch.randn(40, 2, generator=g), torch.randn(40, 2, generator=g)
    assert torch.equal(pb.L1Loss(loss_weight=0.25)(p, t), tp.L1Loss(loss_weight=0.25)(p, t))
    # quaternion / homogeneous transform (pyquaternion, nuscenes-devkit): two formulations, agreement to 1e-15
    rng = np.random.default_rng(3)
    for _ in range(5):
        q, tr = rng.normal(size=4), rng.normal(size=3)
        assert np.abs(pio.quaternion_rotation_matrix(q) - tp.quaternion_rotation_matrix(q)).max() < 1e-14
        for inv in (False, True):
            assert np.abs(pio.transform_matrix(tr, q, inverse=inv) - tp.transform_matrix(tr, q, inverse=inv)).max() < 1e-14
    r = tp.quaternion_rotation_matrix([0.5, 0.5, 0.5, 0.5])      # 120 degrees about (1,1,1): x -> y -> z -> x
    assert np.abs(r @ np.array([1.0, 0, 0]) - np.array([0, 1.0, 0])).max() < 1e-15


def test_refshim_does_not_import_product_leaves():
    """The goldens' third-party leaves come from oracle/thirdparty.py, not from the product package."""
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ('refshim.py', 'thirdparty.py'):
        with open(os.path.join(here, 'oracle', name)) as f:
            lines = [l for l in f if __import__('re').match(r'\s*(from|import)\s+occnet_amd', l)]
        assert lines == [], (name, lines)


@pytest.mark.parametrize("shapes,B,M,D,Q,P", [([(6, 9), (3, 5)], 2, 4, 8, 7, 3), ([(29, 50), (15, 25), (8, 13), (4, 7)], 1, 8, 32, 40, 8),
                                             ([(10, 10)], 2, 8, 32, 25, 4)])
def test_msda_restatement_equals_the_transformers_port(shapes, B, M, D, Q, P):
    """The one function of the path whose source is NOT under /root/reference (mmcv-full's
    multi_scale_deformable_attn_pytorch, DESIGN.md 'Oracle pinning') against an INDEPENDENT third-party port of the same
    published function that ships in this image: Hugging Face transformers' MultiScaleDeformableAttention (a port of
    Deformable-DETR's ms_deform_attn_core_pytorch — the function mmcv vendored).  Sampling locations reach outside [0, 1]
    (zero padding) and sit on pixel centres / borders.  Bit-for-bit: both are the same sequence of torch ops."""
    hf = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
    if not hasattr(hf, "MultiScaleDeformableAttention"):
        pytest.skip("this transformers build has no MultiScaleDeformableAttention module")
    import oracle.msda as om
    g = torch.Generator().manual_seed(len(shapes) * 100 + Q)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    v = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Q, M, L, P, 2, generator=g) * 1.4 - 0.2
    loc[0, 0, 0, :, 0] = 0.5                                    # map centre
    loc[0, 0, 1, :, 0] = torch.tensor([0.0, 1.0])               # corners of the normalised square
    aw = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P)
    st = torch.tensor(shapes)
    try:
        want = hf.MultiScaleDeformableAttention()(v, st, shapes, None, loc, aw, 64)
    except TypeError:
        pytest.skip("MultiScaleDeformableAttention.forward signature differs in this transformers build")
    got = om.multi_scale_deformable_attn_pytorch(v, st, loc, aw)
    assert torch.equal(got, want)
    # and the scalar float64 re-derivation of the CUDA kernel's arithmetic agrees with both
    start = [0]
    for h, w in shapes[:-1]:
        start.append(start[-1] + h * w)
    ref64, _ = om.msda_scalar_f64(v[:1, :, :2].numpy(), shapes, start, loc[:1, :4, :2].numpy(), aw[:1, :4, :2].numpy())
    assert np.abs(ref64.reshape(1, 4, -1) - want[:1, :4].view(1, 4, M, D)[:, :, :2].reshape(1, 4, -1).numpy()).max() < 2e-6


def test_thirdparty_leaves_against_outside_implementations():
    """Two of the oracle-owned third-party leaves have counterparts from OUTSIDE this repository in the image: mmdet's
    LearnedPositionalEncoding is DETR's learned position embedding (transformers' DeformableDetrLearnedPositionEmbedding:
    column embedding | row embedding, channels first), and pyquaternion's rotation matrix is scipy's Rotation (scalar-last
    quaternions there); mmcv's FFN is the feed-forward block of torch.nn.TransformerEncoderLayer.  ConvModule is torch's own
    Conv3d + BatchNorm3d + ReLU by construction; the loss wrappers stay restatement against restatement
    (test_thirdparty_restatements_agree)."""
    from oracle import thirdparty as tp
    hf = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
    if hasattr(hf, "DeformableDetrLearnedPositionEmbedding"):
        h, w, nf = 7, 9, 8
        ours = tp.LearnedPositionalEncoding(nf, row_num_embed=50, col_num_embed=50)
        theirs = hf.DeformableDetrLearnedPositionEmbedding(embedding_dim=nf)
        with torch.no_grad():
            theirs.row_embeddings.weight.copy_(ours.row_embed.weight)
            theirs.column_embeddings.weight.copy_(ours.col_embed.weight)
        try:
            want = theirs(torch.Size((2, 3, h, w)), "cpu", torch.float32)
        except TypeError:
            want = None
        if want is not None:
            with torch.no_grad():
                assert torch.equal(ours(torch.zeros(2, h, w)), want)
    # mmcv FFN (two Linears, ReLU, identity added) == the feed-forward block of torch's own TransformerEncoderLayer
    ffn = tp.FFN(embed_dims=32, feedforward_channels=64, num_fcs=2, ffn_drop=0.1, act_cfg=dict(type='ReLU', inplace=True)).eval()
    enc = torch.nn.TransformerEncoderLayer(32, 4, dim_feedforward=64, dropout=0.1, activation="relu", batch_first=True).eval()
    if hasattr(enc, "_ff_block"):
        with torch.no_grad():
            enc.linear1.weight.copy_(ffn.layers[0][0].weight); enc.linear1.bias.copy_(ffn.layers[0][0].bias)
            enc.linear2.weight.copy_(ffn.layers[1].weight); enc.linear2.bias.copy_(ffn.layers[1].bias)
            x = torch.randn(3, 7, 32, generator=torch.Generator().manual_seed(2))
            assert torch.equal(ffn(x), x + enc._ff_block(x))
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11)
    for _ in range(8):
        q = rng.normal(size=4)                                  # (w, x, y, z), not normalised: both sides normalise
        want = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        assert np.abs(tp.quaternion_rotation_matrix(q) - want).max() < 1e-14
