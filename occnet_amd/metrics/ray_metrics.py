"""RayIoU / mAVE / OccScore — host-side mirror of the reference's
projects/mmdet3d_plugin/datasets/ray_metrics.py (same function names, arguments and return values), with
the ray casting done by the gfx950 kernel (`occnet_amd.ext.dvr_render_forward`) instead of the JIT-built
CUDA extension the reference loads at import time (ray_metrics.py:12).  Nothing is compiled on import.

  generate_lidar_rays()   :63-86    39 pitch rows x 360 azimuths of unit directions
  process_one_sample()    :89-143   binarise -> cast rays from every lidar origin -> per-ray
                                    (class, depth [m], flow x, flow y) at the first occupied voxel
  calc_metrics()          :146-197  per-class IoU at depth thresholds 1/2/4 m, AVE at 2 m
  main()                  :200-257  loop over samples, keep rays whose GT is not free, OccScore
"""
import math

import numpy as np
import torch

_pc_range = [-40, -40, -1.0, 40, 40, 5.4]
_voxel_size = 0.4
_occ_size = [200, 200, 16]

occ_class_names = [
    'car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle', 'pedestrian',
    'traffic_cone', 'barrier', 'driveable_surface', 'other_flat', 'sidewalk', 'terrain', 'manmade',
    'vegetation', 'free']
flow_class_names = ['car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle',
                    'pedestrian']


def generate_lidar_rays():
    """(14040, 3) float32 unit vectors: pitch rows from -(pi/2 - atan(k+1)), k = 0..9, continued with the
    last spacing until the nuScenes upper field of view (0.21 rad); azimuth 0..359 degrees."""
    pitch_angles = [-(math.pi / 2 - math.atan(k + 1)) for k in range(10)]
    while pitch_angles[-1] < 0.21:
        pitch_angles.append(pitch_angles[-1] + (pitch_angles[-1] - pitch_angles[-2]))
    az = np.deg2rad(np.arange(0, 360, 1))
    rays = []
    for pitch in pitch_angles:
        rays.append(np.stack([np.cos(pitch) * np.cos(az), np.cos(pitch) * np.sin(az),
                              np.full_like(az, np.sin(pitch))], -1))
    return np.concatenate(rays, 0).astype(np.float32)


def _render(occ_pred, origin_vox, points_vox, tindex, device):
    from .. import ext
    return ext.dvr_render_forward(occ_pred.to(device), origin_vox.to(device), points_vox.to(device),
                                  tindex.to(device), [1] + list(occ_pred.shape[2:]), "test")


def process_one_sample(sem_pred, lidar_rays, output_origin, flow_pred, device='cuda'):
    """sem_pred (200,200,16) int class ids, lidar_rays (R,3) tensor, output_origin (1,T,3) tensor of lidar
    origins in ego metres, flow_pred (200,200,16,2) -> (T*R, 4) float32 rows (label, depth, flow_x, flow_y)."""
    T = output_origin.shape[1]
    free_id = len(occ_class_names) - 1
    occ = np.where(np.asarray(sem_pred) == free_id, 0, 1)
    # voxel tensor order of the caster is (z, y, x); the prediction is indexed (x, y, z)
    occ_pred = torch.from_numpy(occ).permute(2, 1, 0)[None, None].contiguous().float()
    offset = torch.tensor(_pc_range[:3], dtype=torch.float32)[None, None, :]
    scaler = torch.tensor([_voxel_size] * 3, dtype=torch.float32)[None, None, :]
    lidar_tindex = torch.zeros([1, lidar_rays.shape[0]])
    sem_pred = np.asarray(sem_pred)
    flow_pred = np.asarray(flow_pred)
    out = []
    for t in range(T):
        lidar_origin = output_origin[:, t:t + 1, :]
        lidar_endpts = lidar_rays[None] + lidar_origin
        origin_render = ((lidar_origin - offset) / scaler).float()
        points_render = ((lidar_endpts - offset) / scaler).float()
        with torch.no_grad():
            pred_dist, _, coord_index = _render(occ_pred, origin_render, points_render, lidar_tindex,
                                                device)
            pred_dist = pred_dist * _voxel_size
        coord_index = coord_index[0].int().cpu().numpy()
        pred_flow = torch.from_numpy(flow_pred[coord_index[:, 0], coord_index[:, 1], coord_index[:, 2]])
        pred_label = torch.from_numpy(
            sem_pred[coord_index[:, 0], coord_index[:, 1], coord_index[:, 2]])[:, None]
        out.append(torch.cat([pred_label.float(), pred_dist[0, :, None].cpu(), pred_flow.float()], -1))
    return torch.cat(out, 0).numpy()


def calc_metrics(pcd_pred_list, pcd_gt_list):
    """-> (iou_list: 3 arrays (16,) for depth thresholds 1/2/4 m, ave_list (16,) at 2 m).  Vectorised
    over classes (bincount) for the counts; the flow-error sums keep the reference's per-class float32
    np.sum so the result is bit-identical."""
    thresholds = [1, 2, 4]
    ncls = len(occ_class_names)
    is_flow = np.array([c in flow_class_names for c in occ_class_names])
    gt_cnt = np.zeros(ncls)
    pred_cnt = np.zeros(ncls)
    tp_cnt = np.zeros((len(thresholds), ncls))
    ave = np.where(is_flow, 0.0, np.nan)[None].repeat(len(thresholds), 0)
    ave_count = np.zeros((len(thresholds), ncls))
    for pcd_pred, pcd_gt in zip(pcd_pred_list, pcd_gt_list):
        lp, lg = pcd_pred[:, 0], pcd_gt[:, 0]
        in_p = (lp >= 0) & (lp < ncls) & (lp == np.floor(lp))
        in_g = (lg >= 0) & (lg < ncls) & (lg == np.floor(lg))
        gt_cnt += np.bincount(lg[in_g].astype(np.int64), minlength=ncls)
        pred_cnt += np.bincount(lp[in_p].astype(np.int64), minlength=ncls)
        l1 = np.abs(pcd_pred[:, 1] - pcd_gt[:, 1])
        same = (lp == lg) & in_g
        flow_err = np.linalg.norm(pcd_gt[:, 2:4] - pcd_pred[:, 2:4], axis=1)
        for j, thr in enumerate(thresholds):
            tp = same & (l1 < thr)
            cls = lg[tp].astype(np.int64)
            tp_cnt[j] += np.bincount(cls, minlength=ncls)
            for i in np.nonzero(is_flow)[0]:          # float32 sums per class, as the reference's np.sum
                m = tp & (lg == i)
                if m.any():
                    ave[j][i] += np.sum(flow_err[m])
                    ave_count[j][i] += int(m.sum())
    with np.errstate(divide='ignore', invalid='ignore'):
        iou_list = [(tp_cnt[j] / (gt_cnt + pred_cnt - tp_cnt[j]))[:-1] for j in range(len(thresholds))]
        ave_list = ave[1][:-1] / ave_count[1][:-1]
    return iou_list, ave_list


def main(sem_pred_list, sem_gt_list, flow_pred_list, flow_gt_list, lidar_origin_list, device='cuda',
         verbose=True):
    """-> dict(miou, mave, occ_score, iou_list, ave_list); prints the per-class table like the reference."""
    lidar_rays = torch.from_numpy(generate_lidar_rays())
    pcd_pred_list, pcd_gt_list = [], []
    for sem_pred, sem_gt, flow_pred, flow_gt, lidar_origins in zip(
            sem_pred_list, sem_gt_list, flow_pred_list, flow_gt_list, lidar_origin_list):
        sem_pred = np.reshape(sem_pred, _occ_size)
        sem_gt = np.reshape(sem_gt, _occ_size)
        flow_pred = np.reshape(flow_pred, _occ_size + [2])
        flow_gt = np.reshape(flow_gt, _occ_size + [2])
        pcd_pred = process_one_sample(sem_pred, lidar_rays, lidar_origins, flow_pred, device)
        pcd_gt = process_one_sample(sem_gt, lidar_rays, lidar_origins, flow_gt, device)
        valid = pcd_gt[:, 0].astype(np.int32) != len(occ_class_names) - 1   # non-free GT rays only
        pcd_pred_list.append(pcd_pred[valid])
        pcd_gt_list.append(pcd_gt[valid])
    iou_list, ave_list = calc_metrics(pcd_pred_list, pcd_gt_list)
    miou = float(np.nanmean(iou_list))
    mave = float(np.nanmean(ave_list))
    occ_score = miou * 0.9 + max(1 - mave, 0.0) * 0.1
    if verbose:
        print(f"{'Class Names':22s} {'IoU@1':>7s} {'IoU@2':>7s} {'IoU@4':>7s} {'AVE':>7s}")
        for i in range(len(occ_class_names) - 1):
            print(f"{occ_class_names[i]:22s} {iou_list[0][i]:7.3f} {iou_list[1][i]:7.3f} "
                  f"{iou_list[2][i]:7.3f} {ave_list[i]:7.3f}")
        print(f"{'MEAN':22s} {np.nanmean(iou_list[0]):7.3f} {np.nanmean(iou_list[1]):7.3f} "
              f"{np.nanmean(iou_list[2]):7.3f} {np.nanmean(ave_list):7.3f}")
        print(' --- Occ score:', occ_score)
    return dict(miou=miou, mave=mave, occ_score=occ_score, iou_list=iou_list, ave_list=ave_list)
