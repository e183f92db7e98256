"""Data formats either side of the hot path: what the reference's dataset / pipeline
classes hand to the model and what its evaluation writes — restated on numpy/torch without mmcv, pyquaternion
or the nuScenes devkit.  Host-side formatting only (no model compute lives here).

  pad_multiview / normalize_multiview / to_batch   P/datasets/pipelines/transform_3d.py:12-101
                                                   (PadMultiViewImage size_divisor=32, NormalizeMultiviewImage)
  quaternion_rotation_matrix, transform_matrix     pyquaternion / nuscenes.utils.geometry_utils, as used at
                                                   P/datasets/nuscenes_occ.py:82-86
  camera_matrices                                  P/datasets/nuscenes_occ.py:87-120 (lidar2img = K_pad @ lidar2cam)
  load_occ_gt / save_occ_gt                        P/datasets/pipelines/loading.py:21-33 (.npz: semantics, flow)
  make_img_meta                                    the img_metas keys the path consumes (encoder.py:94-101,133-134)
  lidar_origins                                    tools/ray_iou/ego_pose_extractor.py:84-121 (origins the rays start from)
  format_submission / read_submission              P/datasets/nuscenes_occ.py:189-257 (submission.gz)
"""
import gzip
import os
import pickle

import numpy as np
import torch


# ----------------------------------------------------------------------------- images
def pad_multiview(imgs, size_divisor=32, size=None, pad_val=0):
    """imgs: list of (H, W, C) arrays -> (padded list, meta dict).  Pads bottom/right to `size` (h, w) or to
    the next multiple of `size_divisor` (mmcv.impad / impad_to_multiple semantics): 900x1600 -> 928x1600."""
    assert (size is None) != (size_divisor is None)
    out = []
    for img in imgs:
        h, w = img.shape[:2]
        if size is not None:
            ph, pw = size
        else:
            ph = int(np.ceil(h / size_divisor)) * size_divisor
            pw = int(np.ceil(w / size_divisor)) * size_divisor
        assert ph >= h and pw >= w
        pad = np.full((ph, pw) + img.shape[2:], pad_val, dtype=img.dtype)
        pad[:h, :w] = img
        out.append(pad)
    meta = dict(ori_shape=[i.shape for i in imgs], img_shape=[i.shape for i in out],
                pad_shape=[i.shape for i in out], pad_fixed_size=size, pad_size_divisor=size_divisor)
    return out, meta


def normalize_multiview(imgs, mean, std, to_rgb=True):
    """((img[..., ::-1] if to_rgb) - mean) * (1 / std) in float32: mmcv.imnormalize subtracts the mean and multiplies
    by the reciprocal of std (formed in float64), it does not divide (NormalizeMultiviewImage, transform_3d.py:82-94)."""
    mean = np.asarray(mean, dtype=np.float32).reshape(1, 1, -1)
    std = np.asarray(std, dtype=np.float32).reshape(1, 1, -1)
    stdinv = (1 / np.float64(std)).astype(np.float32)
    out = []
    for img in imgs:
        x = img.astype(np.float32)
        if to_rgb:
            x = x[..., ::-1]
        out.append((x - mean) * stdinv)
    return out, dict(mean=mean.reshape(-1), std=std.reshape(-1), to_rgb=to_rgb)


def to_batch(imgs):
    """list of (H, W, 3) float arrays -> (1, N, 3, H, W) float32 tensor (DefaultFormatBundle3D + collate)."""
    x = np.stack([np.ascontiguousarray(i.transpose(2, 0, 1)) for i in imgs], 0)
    return torch.from_numpy(x.astype(np.float32))[None]


# ----------------------------------------------------------------------------- geometry
def quaternion_rotation_matrix(q):
    """Unit quaternion (w, x, y, z) -> 3x3 rotation matrix (pyquaternion.Quaternion.rotation_matrix)."""
    w, x, y, z = (np.asarray(q, dtype=np.float64) / np.linalg.norm(q)).tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def transform_matrix(translation, rotation, inverse=False):
    """4x4 homogeneous transform from a translation and a (w, x, y, z) quaternion or 3x3 matrix
    (nuscenes.utils.geometry_utils.transform_matrix)."""
    R = np.asarray(rotation, dtype=np.float64)
    if R.shape != (3, 3):
        R = quaternion_rotation_matrix(R)
    t = np.asarray(translation, dtype=np.float64)
    tm = np.eye(4)
    if inverse:
        tm[:3, :3] = R.T
        tm[:3, 3] = R.T @ (-t)
    else:
        tm[:3, :3] = R
        tm[:3, 3] = t
    return tm


def camera_matrices(cams):
    """cams: iterable of dicts with `sensor2lidar_rotation` (3x3 matrix or quaternion),
    `sensor2lidar_translation` (3,), `cam_intrinsic` (3x3) -> (lidar2img, cam_intrinsic, lidar2cam) lists of
    4x4 arrays, built exactly as the reference does (nuscenes_occ.py:96-113)."""
    lidar2img, intrinsics, lidar2cam = [], [], []
    for cam in cams:
        rot = np.asarray(cam['sensor2lidar_rotation'], dtype=np.float64)
        if rot.shape != (3, 3):
            rot = quaternion_rotation_matrix(rot)
        lidar2cam_r = np.linalg.inv(rot)
        lidar2cam_t = np.asarray(cam['sensor2lidar_translation'], dtype=np.float64) @ lidar2cam_r.T
        rt = np.eye(4)
        rt[:3, :3] = lidar2cam_r.T
        rt[3, :3] = -lidar2cam_t
        intrinsic = np.array(cam['cam_intrinsic'], dtype=np.float32)
        viewpad = np.eye(4)
        viewpad[:intrinsic.shape[0], :intrinsic.shape[1]] = intrinsic
        lidar2img.append(viewpad @ rt.T)
        intrinsics.append(viewpad)
        lidar2cam.append(rt.T)
    return lidar2img, intrinsics, lidar2cam


def make_img_meta(cams, lidar2ego_translation, lidar2ego_rotation, img_shapes, can_bus=None,
                  prev_bev_exists=False, **extra):
    """The img_meta dict of one sample with the keys the hot path reads."""
    lidar2img, intr, l2c = camera_matrices(cams)
    meta = dict(lidar2img=lidar2img, cam_intrinsic=intr, lidar2cam=l2c,
                ego2lidar=transform_matrix(lidar2ego_translation, lidar2ego_rotation, inverse=True),
                img_shape=list(img_shapes), can_bus=np.zeros(18) if can_bus is None else np.asarray(can_bus),
                prev_bev_exists=prev_bev_exists)
    meta.update(extra)
    return meta


PSEUDO_LIDAR2EGO = np.array([[0., 1., 0., 0.94], [-1., 0., 0., 0.], [0., 0., 1., 1.84], [0., 0., 0., 1.]])


_SCENE_FRAMES = {}      # id(data_infos) -> (list object, (len, dataset_type), {scene: [frames]})


def lidar_origins(data_infos, index, dataset_type='openocc_v2', max_origins=8, xy_range=39.0):
    """Lidar origins the ray metric / the submission cast from for sample `index`: the lidar position of EVERY frame
    of the sample's scene expressed in the sample's ego frame, kept when |x|, |y| < 39 m, thinned to 8 evenly spaced
    ones (reference tools/ray_iou/ego_pose_extractor.py:84-121, driven by nuscenes_occ.py:142-166,196-224).
    data_infos: the nuScenes info dicts (`token`, `lidar2ego_translation/rotation`, `ego2global_translation/rotation`,
    `scene_token` or `occ_path`).  -> (token, float tensor (1, T, 3)) — the batch a DataLoader(batch_size=1) yields."""
    def scene_of(info):
        if dataset_type == 'openocc_v2' and 'scene_token' not in info:
            return info['occ_path'].split('openocc_v2/')[-1].split('/')[0]
        return info['scene_token']

    def ego_from_lidar(info):
        if dataset_type == 'lightwheelocc':
            return PSEUDO_LIDAR2EGO
        return transform_matrix(info['lidar2ego_translation'], info['lidar2ego_rotation'])

    def global_from_lidar(info):
        return transform_matrix(info['ego2global_translation'], info['ego2global_rotation']).dot(ego_from_lidar(info))
    info = data_infos[index]
    # scene -> frame list, built once per data_infos list (a rescan per call is O(N^2) over a dataset: ADVICE r3)
    cache = _SCENE_FRAMES.get(id(data_infos))
    if cache is None or cache[0] is not data_infos or cache[1] != (len(data_infos), dataset_type):
        by_scene = {}
        for f in data_infos:
            by_scene.setdefault(scene_of(f), []).append(f)
        cache = (data_infos, (len(data_infos), dataset_type), by_scene)
        _SCENE_FRAMES.clear()
        _SCENE_FRAMES[id(data_infos)] = cache
    frames = cache[2][scene_of(info)]
    ref_index = next(i for i, f in enumerate(frames) if f is info)
    ref_lidar_from_global = np.linalg.inv(global_from_lidar(info))
    ref_ego_from_lidar = ego_from_lidar(info)
    origins = []
    for i, frame in enumerate(frames):
        if i == ref_index:
            o = np.array([0.0, 0.0, 0.0], dtype=np.float32)
        else:
            o = np.array(ref_lidar_from_global.dot(global_from_lidar(frame))[:3, 3], dtype=np.float32)
        pad = np.ones([4])
        pad[:3] = o
        o = np.dot(ref_ego_from_lidar[:3], pad.T).T
        if np.abs(o[0]) < xy_range and np.abs(o[1]) < xy_range:
            origins.append(o)
    if len(origins) > max_origins:
        sel = np.round(np.linspace(0, len(origins) - 1, max_origins)).astype(np.int64)
        origins = [origins[i] for i in sel]
    return info['token'], torch.from_numpy(np.stack(origins))[None]


# ----------------------------------------------------------------------------- occupancy ground truth
OCC_SHAPE = (200, 200, 16)


def load_occ_gt(path, shape=OCC_SHAPE):
    """-> (semantics uint8 shape, flow float32 shape+(2,)); zeros when the file is absent, as the reference."""
    if path is not None and os.path.exists(path):
        z = np.load(path)
        return z['semantics'], z['flow']
    return np.zeros(shape, dtype=np.uint8), np.zeros(tuple(shape) + (2,), dtype=np.float32)


def save_occ_gt(path, semantics, flow):
    np.savez_compressed(path, semantics=np.asarray(semantics, dtype=np.uint8),
                        flow=np.asarray(flow, dtype=np.float32))


# ----------------------------------------------------------------------------- submission
SUBMISSION_HEADER = {
    'method': 'XXXXX (Your method name)', 'team': 'XXXXX (Your team name)', 'authors': "XXXXX (Authors)",
    'e-mail': "XXXXX (Your email)", 'institution / company': "XXXXXXXXXX (Your affiliation)",
    'country / region': "XXXXXXX (Your country/region)"}


def format_submission(samples, submission_prefix, header=None, device='cuda'):
    """samples: iterable of (token, sem_pred (200,200,16) ints, flow_pred (200,200,16,2), lidar origins
    (1, T, 3) tensor in ego metres).  Casts the 14 040 lidar rays per origin through every prediction (HIP
    DVR kernel) and writes `<prefix>/submission.gz` = gzip(pickle({... 'results': {token: {pcd_cls int8,
    pcd_dist float16, pcd_flow float16}}}), mtime=0).  -> path."""
    from .metrics import generate_lidar_rays, process_one_sample
    os.makedirs(submission_prefix, exist_ok=True)
    rays = torch.from_numpy(generate_lidar_rays())
    results = {}
    for token, sem_pred, flow_pred, origins in samples:
        sem_pred = np.reshape(np.asarray(sem_pred), OCC_SHAPE)
        flow_pred = np.reshape(np.asarray(flow_pred), OCC_SHAPE + (2,))
        pcd = process_one_sample(sem_pred, rays, origins, flow_pred, device)
        results[token] = {'pcd_cls': pcd[:, 0].astype(np.int8), 'pcd_dist': pcd[:, 1].astype(np.float16),
                          'pcd_flow': pcd[:, 2:4].astype(np.float16)}
    final = dict(SUBMISSION_HEADER if header is None else header)
    final['results'] = results
    path = os.path.join(submission_prefix, 'submission.gz')
    with open(path, 'wb') as f:
        f.write(gzip.compress(pickle.dumps(final), mtime=0))
    return path


def read_submission(path):
    with open(path, 'rb') as f:
        return pickle.loads(gzip.decompress(f.read()))
