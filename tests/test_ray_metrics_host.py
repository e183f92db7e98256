"""not-gpu: host logic of the RayIoU metric (ray set, vectorised class statistics) vs the oracle's literal
restatement of the reference loops, and the plain-C ray caster on hand-checkable cases."""
import numpy as np
import pytest
import torch

from occnet_amd.metrics import calc_metrics, generate_lidar_rays
from oracle import ray_metrics_ref as oref


def test_lidar_ray_set():
    rays = generate_lidar_rays()
    assert rays.shape == (39 * 360, 3) and rays.dtype == np.float32     # 14 040 rays
    assert np.allclose(np.linalg.norm(rays, axis=1), 1.0, atol=1e-6)
    assert np.isclose(rays[0, 2], -np.sin(np.pi / 2 - np.arctan(1.0)))  # steepest row: -45 degrees


def test_calc_metrics_matches_reference_loops():
    rng = np.random.default_rng(0)
    preds, gts = [], []
    for n in (500, 1, 2000):
        gt = np.stack([rng.integers(0, 16, n).astype(np.float32), rng.uniform(0, 50, n).astype(np.float32),
                       rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)], 1)
        pr = gt.copy()
        flip = rng.random(n) < 0.3
        pr[flip, 0] = rng.integers(0, 17, int(flip.sum()))
        pr[:, 1] += rng.normal(scale=1.5, size=n).astype(np.float32)
        pr[:, 2:] += rng.normal(scale=0.3, size=(n, 2)).astype(np.float32)
        preds.append(pr)
        gts.append(gt)
    iou_a, ave_a = calc_metrics(preds, gts)
    iou_b, ave_b = oref.calc_metrics(preds, gts)
    for a, b in zip(iou_a, iou_b):
        assert np.allclose(a, b, equal_nan=True, rtol=0, atol=0)
    assert np.array_equal(ave_a, ave_b, equal_nan=True)


def test_c_ray_caster_axis_aligned_and_miss():
    try:
        oref.dvr_lib()
    except FileNotFoundError:
        pytest.skip("oracle/_build/libdvr_ref.so not built (make -C oracle)")
    sigma = torch.zeros(1, 1, 4, 6, 8)          # (N, T, Z, Y, X)
    sigma[0, 0, 1, 2, 5] = 1.0                  # one occupied voxel at x=5, y=2, z=1
    origin = torch.tensor([[[0.5, 2.5, 1.5]]])
    points = torch.tensor([[[7.5, 2.5, 1.5],     # +x through the occupied voxel
                            [0.5, 5.5, 1.5],     # +y, hits nothing: last in-grid voxel (0, 5, 1)
                            [-3.0, 2.5, 1.5]]])  # -x: leaves the grid at once from voxel (0, 2, 1)
    tindex = torch.zeros(1, 3)
    pred, gt, coord = oref.render_forward(sigma, origin, points, tindex)
    assert pred[0, 0].item() == 5.5 and coord[0, 0].tolist() == [5.0, 2.0, 1.0]
    assert pred[0, 1].item() == 3.5 and coord[0, 1].tolist() == [0.0, 5.0, 1.0]
    assert pred[0, 2].item() == 0.5 and coord[0, 2].tolist() == [0.0, 2.0, 1.0]
    assert gt[0].tolist() == [7.0, 3.0, 3.5]
    # origin outside the grid pointing away: never enters -> initial values survive
    origin2 = torch.tensor([[[-2.5, 2.5, 1.5]]])
    pred2, gt2, coord2 = oref.render_forward(sigma, origin2, torch.tensor([[[-5.0, 2.5, 1.5]]]),
                                             torch.zeros(1, 1))
    assert pred2.item() == -1.0 and gt2.item() == -1.0 and coord2.abs().sum().item() == 0.0
    # padded ray (tindex < 0) is skipped
    pred3, _, _ = oref.render_forward(sigma, origin, points, torch.tensor([[0.0, -1.0, 0.0]]))
    assert pred3[0, 1].item() == -1.0 and pred3[0, 0].item() == 5.5
