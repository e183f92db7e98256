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


# ---- full BASE geometry (VERDICT r2 missing #3): the reference's own files at the benchmarked size ------------
# 200x200 BEV queries, 6 cameras, FPN maps 116x200 / 58x100 / 29x50 / 15x25, pillar_h 16, ONE encoder layer.
# The fixtures hold a strided subsample of every output tensor plus float64 sums (tests/golden/base_full_*.npz).
FULL_CASES = {
    'base_full_nohist': dict(seed=21, batch=1, prev=False, angle=0.0,
                             geometry=dict(synthetic.BASE, num_points=8, num_layers=1)),
    # history BEV rotated by can_bus[-1] = 7.5 degrees about (100, 100): TSA's two-value, rotated-history branch
    'base_full_hist': dict(seed=22, batch=1, prev=True, angle=7.5,
                           geometry=dict(synthetic.BASE, num_points=8, num_layers=1)),
    # the BENCHMARKED depth (round 4): all four encoder layers of the base config
    'base_full_4layer': dict(seed=23, batch=1, prev=False, angle=0.0,
                             geometry=dict(synthetic.BASE, num_points=8, num_layers=4)),
    # round 5: the benchmarked DEPTH for BASELINE configs[2] and configs[4] as well —
    # four layers with a rotated history BEV (every layer's TSA attends to it), and the 400 x 400 x 32 grid
    # (160 000 BEV queries, max_len ~ 39 600 padded rows per camera in the reference's rebatch)
    'base_full_hist_4layer': dict(seed=24, batch=1, prev=True, angle=7.5,
                                  geometry=dict(synthetic.BASE, num_points=8, num_layers=4)),
    'hires_full_4layer': dict(seed=25, batch=1, prev=False, angle=0.0,
                              geometry=dict(synthetic.HIRES, num_points=8, num_layers=4)),
}
FULL_STRIDE = 997            # prime, coprime to every tensor dimension: the subsample walks all axes
FULL_FINE = 512              # round 5: sums of every 512 contiguous elements (a localised error in a
                             # region the strided subsample skips moves one of these; 64 slab sums average it away)
FULL_KEYS = ('bev_embed', 'occ', 'flow', 'layer0_tsa_out', 'layer0_sca_out')


def full_case_inputs(case):
    feats, metas, prev_bev = case_inputs(case)
    for m in metas:
        m['can_bus'][-1] = case['angle']
        m['prev_bev_exists'] = bool(case['prev'])
    return feats, metas, prev_bev


def digest(t):
    """-> dict of the fixture entries for one output tensor: strided subsample, float64 sum and |sum|, and the
    sums of 64 contiguous slabs (localises a disagreement)."""
    import numpy as np
    a = np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, 'detach') else t).reshape(-1)
    slabs = np.array([s.astype(np.float64).sum() for s in np.array_split(a, 64)])
    pad = (-a.size) % FULL_FINE
    fine = np.concatenate([a, np.zeros(pad, a.dtype)]).reshape(-1, FULL_FINE).astype(np.float64).sum(1).astype(np.float32)
    return dict(sub=a[::FULL_STRIDE].astype(np.float32).copy(), sum=np.float64(a.astype(np.float64).sum()),
                abs_sum=np.float64(np.abs(a.astype(np.float64)).sum()), slabs=slabs, n=np.int64(a.size), fine=fine)


def compare_digest(name, t, gold, tol):
    """assert tensor t agrees with the stored digest: every subsampled element within tol, and the slab sums within
    tol * sqrt-free bound (tol * elements per slab would be the worst case; a mean-error bound of tol/10 is asked)."""
    import numpy as np
    d = digest(t)
    assert int(d['n']) == int(gold[f'{name}_n']), (name, int(d['n']), int(gold[f'{name}_n']))
    sub_err = float(np.abs(d['sub'].astype(np.float64) - gold[f'{name}_sub'].astype(np.float64)).max())
    assert sub_err < tol, f'{name}: subsample differs from the reference by {sub_err}'
    per = int(d['n']) / 64.0
    slab_err = float(np.abs(d['slabs'] - gold[f'{name}_slabs']).max()) / per
    assert slab_err < tol / 10, f'{name}: mean error over a slab {slab_err}'
    if f'{name}_fine' in gold:
        # every element is covered: the MEAN error over any 512 contiguous elements stays below tol / 4 — one element off by
        # 0.13 (at tol 1e-3) in an otherwise matching chunk fails, so does one wrong BEV query (every channel's chunk moves).
        # Not tighter: the errors the fp16 value rows leave (<= 2.4e-4 per element) are CORRELATED along a chunk (neighbouring
        # queries of one channel share their sampled rows): measured 0.035 per chunk at four layers, sqrt(512) x the
        # per-element error would have been 0.005
        fine = np.abs(d['fine'].astype(np.float64) - gold[f'{name}_fine'].astype(np.float64))
        worst = int(fine.argmax())
        assert float(fine.max()) < tol / 4 * FULL_FINE, \
            f'{name}: elements [{worst * FULL_FINE}, {(worst + 1) * FULL_FINE}) sum to {float(fine.max())} off the reference'
    return sub_err, slab_err


def checksum(tensors):
    return float(sum(t.detach().double().abs().sum() for t in tensors))


# ---- RayIoU / mAVE / OccScore -----------------------------------------------------------
METRIC_SEEDS = (41, 42)


def metric_scene(seed):
    """A seeded synthetic occupancy scene at the metric's hard-coded size (ray_metrics.py:209-212 reshapes to
    [200, 200, 16]): -> (sem_pred, sem_gt uint8 (200,200,16), flow_pred, flow_gt float32 (200,200,16,2),
    lidar_origins float tensor (1, T, 3)).  Ground plane + boxes of several classes; the prediction shifts /
    relabels / drops some of them so every term of the score (per-class IoU at 1/2/4 m, flow error of true
    positives, free rays) is exercised."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    FREE = 16
    gt = np.full((200, 200, 16), FREE, np.uint8)
    gt[:, :, 0:2] = 10                                       # driveable surface
    gt[:, :60, 0:2] = 12                                     # sidewalk strip
    gt[:, 185:, 0:12] = 14                                   # a wall (manmade)
    pred = gt.copy()
    fgt = np.zeros((200, 200, 16, 2), np.float32)
    fpred = np.zeros_like(fgt)
    for i in range(28):
        cls = int(rng.choice([0, 1, 3, 5, 6, 7, 8, 9, 15]))
        sx, sy, sz = (int(v) for v in rng.integers(2, 9, 3))
        x0, y0 = int(rng.integers(40, 160 - sx)), int(rng.integers(40, 160 - sy))
        gt[x0:x0 + sx, y0:y0 + sy, 2:2 + sz] = cls
        vel = rng.normal(scale=2.0, size=2).astype(np.float32) if cls < 8 else np.zeros(2, np.float32)
        fgt[x0:x0 + sx, y0:y0 + sy, 2:2 + sz] = vel
        mode = i % 4
        if mode == 3:
            continue                                         # missed object
        dx, dy = (int(v) for v in rng.integers(-4, 5, 2)) if mode == 1 else (0, 0)
        pcls = int(rng.choice([0, 1, 7])) if mode == 2 else cls
        pred[x0 + dx:x0 + dx + sx, y0 + dy:y0 + dy + sy, 2:2 + sz] = pcls
        fpred[x0 + dx:x0 + dx + sx, y0 + dy:y0 + dy + sy, 2:2 + sz] = vel + rng.normal(scale=0.4, size=2)
    origins = torch.tensor([[[0.9858, 0.0, 1.8402], [2.4, -0.7, 1.84]]], dtype=torch.float32)
    return pred, gt, fpred, fgt, origins


# ---- input / output formats ---------------------------------------------------------------------
PIPELINE_CASES = {
    # the shipped configs' normalisation (bevformer_base_occ.py:14-15) on 6 small "camera" frames: 70 x 100 -> 96 x 128
    'base_norm': dict(seed=31, n=6, hw=(70, 100), mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False),
    # a general normalisation (ImageNet statistics, BGR -> RGB): exercises the reciprocal-std and channel-flip paths
    'rgb_norm': dict(seed=32, n=2, hw=(64, 96), mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
}


def pipeline_images(case):
    import numpy as np
    rng = np.random.default_rng(case['seed'])
    return [rng.integers(0, 256, case['hw'] + (3,), dtype=np.uint8) for _ in range(case['n'])]


def dataset_infos(occ_dir='/nonexistent/openocc_v2'):
    """Three keyframes of one synthetic scene in the nuScenes `infos` layout NuSceneOcc reads (nuscenes_occ.py:60-120,
    ego_pose_extractor.py:36-80): 6 cameras of the synthetic rig (sensor2lidar extrinsics as 3x3 matrices), a lidar
    mounted 0.94 m ahead / 1.84 m up with a small yaw, an ego driving forward ~4 m per frame while turning."""
    import math
    import numpy as np
    from occnet_amd import synthetic
    rng = np.random.default_rng(33)
    infos = []
    for i in range(3):
        yaw = 0.03 * i
        cams = {}
        for c, (cam_yaw, t, f) in enumerate(synthetic._RIG):
            psi = math.radians(cam_yaw + rng.normal(0, 0.2))
            R_l2c = np.array([[math.sin(psi), -math.cos(psi), 0.0], [0.0, 0.0, -1.0], [math.cos(psi), math.sin(psi), 0.0]])
            cams[f'CAM_{c}'] = dict(data_path=f'samples/CAM_{c}/{i:04d}.jpg', sensor2lidar_rotation=R_l2c.T,
                                    sensor2lidar_translation=np.asarray(t) + rng.normal(0, 0.01, 3),
                                    cam_intrinsic=np.array([[f, 0.0, 816.0 + c], [0.0, f, 491.0 - c], [0.0, 0.0, 1.0]]))
        infos.append(dict(
            token=f'tok{i}', timestamp=1.5e15 + 5e5 * i, cams=cams,
            occ_path=f'{occ_dir}/scene-0001/tok{i}/labels.npz',
            lidar2ego_translation=[0.94, 0.0, 1.84], lidar2ego_rotation=[math.cos(0.004), 0.0, 0.0, math.sin(0.004)],
            ego2global_translation=[600.0 + 4.1 * i, 1600.0 + 0.3 * i * i, 0.0],
            ego2global_rotation=[math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)]))
    return dict(infos=infos, metadata=dict(version='v1.0-trainval'))
