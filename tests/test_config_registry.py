"""not-gpu: the drop-in boundary on the host side — configs load unchanged, registry names resolve,
constructor kwargs are tolerated, state_dict keys follow the reference layout."""
import glob
import os

import pytest
import torch

from occnet_amd.plugin import (ATTENTION, DETECTORS, HEADS, POSITIONAL_ENCODING, TRANSFORMER,
                               TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, Config, build_model,
                               import_plugin)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VENDORED = os.path.join(ROOT, 'projects', 'configs')          # byte-identical copies of the reference's configs
REF_CFG = os.path.join(VENDORED, 'bevformer')
REF_TREE = '/root/reference/projects/configs'                 # only in the build container


def test_registry_surface():
    assert 'BEVFormerOcc' in DETECTORS and 'BEVFormerOccHead' in HEADS
    assert 'TransformerOcc' in TRANSFORMER
    assert 'BEVFormerEncoder' in TRANSFORMER_LAYER_SEQUENCE
    for n in ('BEVFormerLayer', 'MyCustomBaseTransformerLayer'):
        assert n in TRANSFORMER_LAYER
    for n in ('TemporalSelfAttention', 'SpatialCrossAttention', 'MSDeformableAttention3D'):
        assert n in ATTENTION
    assert 'LearnedPositionalEncoding' in POSITIONAL_ENCODING
    # detection-branch names §8(b) lists (no occ config builds them)
    assert 'PerceptionTransformer' in TRANSFORMER
    assert 'DetectionTransformerDecoder' in TRANSFORMER_LAYER_SEQUENCE
    assert 'CustomMSDeformableAttention' in ATTENTION
    assert 'LearnedPositionalEncoding3D' in POSITIONAL_ENCODING


def test_vendored_configs_are_byte_identical():
    """projects/configs holds the reference's five config files unmodified: checksums match SHA256SUMS (taken
    from the reference tree) and, where the reference tree exists, the files themselves."""
    import hashlib
    with open(os.path.join(VENDORED, 'SHA256SUMS')) as f:
        sums = dict(reversed(l.split()) for l in f if l.strip())
    assert len(sums) == 5
    for rel, want in sums.items():
        with open(os.path.join(VENDORED, rel), 'rb') as f:
            data = f.read()
        assert hashlib.sha256(data).hexdigest() == want, rel
        ref = os.path.join(REF_TREE, rel)
        if os.path.exists(ref):
            with open(ref, 'rb') as f:
                assert f.read() == data, rel


def test_learned_positional_encoding_3d_shape():
    from occnet_amd.plugin import build_positional_encoding
    pe = build_positional_encoding(dict(type='LearnedPositionalEncoding3D', num_feats=4, row_num_embed=5,
                                        col_num_embed=6, height_num_embed=3))
    pos = pe(torch.zeros(2, 3, 5, 6))
    assert pos.shape == (2, 12, 3, 5, 6)
    assert torch.equal(pos[0, :4, 1, 2, :].T, pe.col_embed.weight)          # x block varies along w only
    assert torch.equal(pos[1, 8:, :, 4, 5].T, pe.height_embed.weight)


def test_plugin_import_convention():
    cfg = Config(dict(plugin=True, plugin_dir='projects/mmdet3d_plugin/'))
    mod = import_plugin(cfg)
    assert mod.__name__ == 'projects.mmdet3d_plugin'
    assert hasattr(mod, 'BEVFormerOcc')


@pytest.mark.parametrize('name', ['bevformer_base_occ.py', 'bevformer_base_occ_test.py',
                                  'bevformer_base_occ_w_lightwheel.py'])
def test_reference_configs_load_unchanged(name):
    cfg = Config.fromfile(os.path.join(REF_CFG, name))
    assert cfg.model.type == 'BEVFormerOcc'
    assert cfg.dist_params == dict(backend='nccl')          # from _base_/default_runtime.py
    assert cfg.data.samples_per_gpu == 1
    assert cfg.model.pts_bbox_head.transformer.encoder.num_layers == 4
    if 'lightwheel' in name:                                # recursive merge keeps base-only keys
        assert cfg.data.train.type == 'ConcatDataset' and 'datasets' in cfg.data.train
    import_plugin(cfg)
    model = build_model(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    n = sum(p.numel() for p in model.parameters())
    assert 40e6 < n < 41e6                                  # ~40 M parameters


def test_state_dict_layout_matches_reference():
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'occ_base_200x200x16.py'))
    model = build_model(cfg.model)
    keys = set(model.state_dict().keys())
    pre = 'pts_bbox_head.transformer.encoder.layers.3.'
    expected = [
        'pts_bbox_head.bev_embedding.weight',
        'pts_bbox_head.positional_encoding.row_embed.weight',
        'pts_bbox_head.positional_encoding.col_embed.weight',
        'pts_bbox_head.transformer.level_embeds', 'pts_bbox_head.transformer.cams_embeds',
        pre + 'attentions.0.sampling_offsets.weight', pre + 'attentions.0.attention_weights.bias',
        pre + 'attentions.0.value_proj.weight', pre + 'attentions.0.output_proj.bias',
        pre + 'attentions.1.deformable_attention.sampling_offsets.weight',
        pre + 'attentions.1.deformable_attention.attention_weights.weight',
        pre + 'attentions.1.deformable_attention.value_proj.bias',
        pre + 'attentions.1.output_proj.weight',
        pre + 'ffns.0.layers.0.0.weight', pre + 'ffns.0.layers.1.bias',
        pre + 'norms.0.weight', pre + 'norms.2.bias',
        'pts_bbox_head.transformer.decoder.0.conv.weight',
        'pts_bbox_head.transformer.decoder.1.bn.running_var',
        'pts_bbox_head.transformer.predicter.0.weight', 'pts_bbox_head.transformer.predicter.2.bias',
        'pts_bbox_head.transformer.flow_predicter.2.weight',
        'img_backbone.conv1.weight', 'img_backbone.layer4.2.conv3.weight',
        'img_backbone.layer2.0.downsample.1.running_mean',
        'img_neck.lateral_convs.0.conv.weight', 'img_neck.fpn_convs.3.conv.bias',
    ]
    missing = [k for k in expected if k not in keys]
    assert not missing, missing
    sd = model.state_dict()
    assert sd['pts_bbox_head.bev_embedding.weight'].shape == (40000, 256)
    assert sd[pre + 'attentions.0.sampling_offsets.weight'].shape == (128, 512)
    assert sd[pre + 'attentions.1.deformable_attention.sampling_offsets.weight'].shape == (512, 256)
    assert sd['pts_bbox_head.transformer.decoder.0.conv.weight'].shape == (32, 16, 3, 3, 3)
    assert not any(k.endswith('decoder.0.conv.bias') for k in keys)      # bias=False with a norm


def test_own_configs_build():
    for f in sorted(glob.glob(os.path.join(ROOT, 'configs', '*.py'))):
        cfg = Config.fromfile(f)
        model = build_model(cfg.model)
        assert model.pts_bbox_head.bev_h == cfg.bev_h


def test_config_merge_semantics(tmp_path):
    (tmp_path / 'base.py').write_text("a = dict(x=1, y=dict(p=1, q=2), z=[1, 2, 3])\nv = 5\n")
    (tmp_path / 'child.py').write_text(
        "_base_ = ['./base.py']\na = dict(y=dict(q=3, r=4), z=[9])\nw = dict(_delete_=True, k=1)\n")
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.a.x == 1 and cfg.a.y == dict(p=1, q=3, r=4) and cfg.a.z == [9] and cfg.v == 5
    assert cfg.w == dict(k=1)
    cfg.merge_from_dict({'a.y.p': 7, 'new.k': 1})
    assert cfg.a.y.p == 7 and cfg.new.k == 1 and cfg.a.x == 1
    (tmp_path / 'bad.py').write_text("_base_ = ['./base.py', './base.py']\n")
    with pytest.raises(KeyError):
        Config.fromfile(str(tmp_path / 'bad.py'))


def test_constructor_kwarg_tolerance_and_init():
    from tests.util import head_cfg, small_cfg
    from occnet_amd.plugin import build_head
    cfg = head_cfg(small_cfg(bev=(8, 8)))
    cfg.update(train_cfg=dict(grid_size=[512, 512, 1]), test_cfg=None, sync_cls_avg_factor=True)
    head = build_head(cfg)
    head.init_weights()
    sca = head.transformer.encoder.layers[0].attentions[1].deformable_attention
    assert float(sca.sampling_offsets.weight.abs().max()) == 0.0        # reference init zeroes it
    b = sca.sampling_offsets.bias.view(8, 4, 8, 2)
    assert torch.allclose(b[0, 0, :, 0], torch.arange(1, 9, dtype=torch.float32))  # (cos0, sin0)*(i+1)
    assert float(sca.attention_weights.bias.abs().max()) == 0.0
