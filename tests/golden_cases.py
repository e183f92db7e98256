"""The golden cases shared by oracle/gen_golden.py (writer, runs the reference's own files) and the
tests (readers).  Geometries are small enough that fixtures stay well under 1 MB each."""
import torch

from occnet_amd import synthetic
from tests.util import small_cfg

CASES = {
    # 6 nuScenes-rig cameras, 4 levels x 8 points, 8 z-anchors (base-config structure), no history
    'base_struct_nohist': dict(seed=11, batch=1, prev=False, geometry=small_cfg(
        bev=(12, 12), feat_shapes=((8, 13), (4, 7), (2, 4), (1, 2)), num_layers=2)),
    # same with a history BEV (TSA attends to prev_bev; rotation angle 0)
    'base_struct_hist': dict(seed=12, batch=1, prev=True, geometry=small_cfg(
        bev=(12, 12), feat_shapes=((8, 13), (4, 7), (2, 4), (1, 2)), num_layers=1)),
    # batch of 2 (quirks 1 and 5 of SURVEY.md Appendix C: batch-0 mask, interleaved TSA stack)
    'base_struct_bs2': dict(seed=13, batch=2, prev=False, geometry=small_cfg(
        bev=(12, 12), feat_shapes=((8, 13), (4, 7), (2, 4), (1, 2)), num_layers=1)),
    # tiny config structure: 1 camera, 4 z-anchors with 8 points (2 points per anchor), pillar_h 4
    'tiny_struct': dict(seed=14, batch=1, prev=False, geometry=dict(
        synthetic.TINY, bev_h=12, bev_w=12, feat_shapes=((8, 8), (4, 4), (2, 2), (1, 1)),
        num_points=8, num_layers=2)),
}


def case_inputs(case):
    g = case['geometry']
    feats = synthetic.make_features(g, batch=case['batch'], seed=case['seed'])
    metas = synthetic.make_img_metas(g, batch=case['batch'], seed=case['seed'],
                                     jitter=0.5 if case['batch'] > 1 else 0.0)
    prev_bev = None
    if case['prev']:
        gen = torch.Generator().manual_seed(case['seed'] + 100)
        prev_bev = torch.randn(case['batch'], g['bev_h'] * g['bev_w'], g['embed_dims'],
                               generator=gen) * 0.5
    return feats, metas, prev_bev


def checksum(tensors):
    return float(sum(t.detach().double().abs().sum() for t in tensors))
