"""not-gpu: input-side formats (pad / normalise / camera matrices / occupancy GT files), SURVEY.md §8f N4."""
import numpy as np
import torch

from occnet_amd import io as oio
from occnet_amd import synthetic


def test_pad_and_normalize_multiview():
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8) for _ in range(2)]
    mean, std = [103.530, 116.280, 123.675], [1.0, 1.0, 1.0]        # bevformer_base_occ.py:14-15
    norm, cfg = oio.normalize_multiview(imgs, mean, std, to_rgb=False)
    pad, meta = oio.pad_multiview(norm, size_divisor=32)
    assert meta['ori_shape'][0] == (900, 1600, 3) and meta['img_shape'][0] == (928, 1600, 3)
    assert pad[0].dtype == np.float32
    assert np.array_equal(pad[0][:900], imgs[0].astype(np.float32) - np.float32(mean).reshape(1, 1, 3))
    assert float(np.abs(pad[0][900:]).max()) == 0.0                  # bottom rows are padding
    rgb, _ = oio.normalize_multiview(imgs[:1], [0, 0, 0], [2, 2, 2], to_rgb=True)
    assert np.array_equal(rgb[0], imgs[0][..., ::-1].astype(np.float32) / 2)
    batch = oio.to_batch(pad)
    assert batch.shape == (1, 2, 3, 928, 1600) and batch.dtype == torch.float32
    assert np.array_equal(batch[0, 1, 2].numpy(), pad[1][..., 2])


def test_quaternion_and_transform_matrix():
    q = [0.7071067811865476, 0.0, 0.0, 0.7071067811865476]          # 90 degrees about z
    R = oio.quaternion_rotation_matrix(q)
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-12)
    t = [1.0, 2.0, 3.0]
    fwd = oio.transform_matrix(t, q)
    inv = oio.transform_matrix(t, q, inverse=True)
    assert np.allclose(fwd @ inv, np.eye(4), atol=1e-12)
    assert np.allclose(oio.transform_matrix(t, R), fwd)


def test_camera_matrices_project_like_the_rig():
    """lidar2img built from sensor2lidar extrinsics + intrinsics (reference formula) projects a lidar point
    to the same pixel as explicit pinhole geometry, and equals synthetic.camera_matrix for the same rig."""
    yaw, t, f = -55.0, (1.55, -0.49, 1.50), 1266.0
    psi = np.radians(yaw)
    R_l2c = np.array([[np.sin(psi), -np.cos(psi), 0.0], [0.0, 0.0, -1.0], [np.cos(psi), np.sin(psi), 0.0]])
    K = np.array([[f, 0, 816.0], [0, f, 491.0], [0, 0, 1.0]])
    cam = dict(sensor2lidar_rotation=R_l2c.T, sensor2lidar_translation=np.asarray(t), cam_intrinsic=K)
    l2i, intr, l2c = oio.camera_matrices([cam])
    assert np.allclose(l2i[0], synthetic.camera_matrix(yaw, t, f), atol=1e-4)   # intrinsics pass through f32
    p = np.array([12.0, -9.0, 0.3, 1.0])
    uvw = l2i[0] @ p
    pc = R_l2c @ (p[:3] - np.asarray(t))
    assert np.allclose(uvw[:2] / uvw[2], (K @ pc)[:2] / pc[2], rtol=1e-6)
    # quaternion form of the same rotation (LightwheelOcc branch, nuscenes_occ.py:97-98)
    from math import cos, sin
    meta = oio.make_img_meta([cam], [0.94, 0.0, 1.84], [cos(0.01), 0, 0, sin(0.01)], [(928, 1600, 3)])
    assert set(('lidar2img', 'ego2lidar', 'img_shape', 'can_bus')) <= set(meta)
    assert np.allclose(meta['ego2lidar'] @ oio.transform_matrix([0.94, 0.0, 1.84], [cos(0.01), 0, 0, sin(0.01)]),
                       np.eye(4), atol=1e-12)


def test_occ_gt_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    sem = rng.integers(0, 18, (200, 200, 16)).astype(np.uint8)
    flow = rng.normal(size=(200, 200, 16, 2)).astype(np.float32)
    path = str(tmp_path / 'labels.npz')
    oio.save_occ_gt(path, sem, flow)
    s2, f2 = oio.load_occ_gt(path)
    assert s2.dtype == np.uint8 and np.array_equal(s2, sem) and np.array_equal(f2, flow)
    s0, f0 = oio.load_occ_gt(str(tmp_path / 'missing.npz'))
    assert s0.shape == (200, 200, 16) and not s0.any() and f0.shape == (200, 200, 16, 2)
