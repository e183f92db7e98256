"""Oracle for the RayIoU metric glue (test infrastructure; see oracle/__init__.py).

Literal restatement of projects/mmdet3d_plugin/datasets/ray_metrics.py: `process_one_sample` (:89-143)
driving the plain-C ray caster (oracle/dvr_ref.c) and `calc_metrics` (:146-197) with the reference's
per-class Python loops.  PINNED: tests/golden/ray_metrics.npz holds the outputs of the reference's own file run in
place (oracle/refshim.install_metrics + oracle/build_ref.py) — tests/test_ray_metrics_golden.py checks this
restatement against them bit for bit, and oracle/dvr_ref.c against the reference's kernel body."""
import ctypes
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_pc_range = [-40, -40, -1.0, 40, 40, 5.4]
_voxel_size = 0.4
occ_class_names = ['car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle',
                   'pedestrian', 'traffic_cone', 'barrier', 'driveable_surface', 'other_flat', 'sidewalk',
                   'terrain', 'manmade', 'vegetation', 'free']
flow_class_names = ['car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle',
                    'pedestrian']


def dvr_lib():
    so = os.path.join(HERE, '_build', 'libdvr_ref.so')
    if not os.path.exists(so):
        raise FileNotFoundError(f"{so} not built (make -C oracle)")
    return ctypes.CDLL(so)


def render_forward(sigma, origin, points, tindex, phase_name="test"):
    """CPU float32 tensors, shapes as dvr.render_forward -> (pred_dist, gt_dist, coord_index)."""
    sigma, origin, points, tindex = (t.contiguous().float() for t in (sigma, origin, points, tindex))
    N, T, Z, Y, X = sigma.shape
    M = points.shape[1]
    pred = torch.empty(N, M)
    gt = torch.empty(N, M)
    coord = torch.empty(N, M, 3)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    dvr_lib().dvr_render_forward_ref(p(sigma), p(origin), p(points), p(tindex), p(pred), p(gt), p(coord),
                                     N, T, Z, Y, X, M, points.shape[2], 1 if phase_name == "train" else 0)
    return pred, gt, coord


def process_one_sample(sem_pred, lidar_rays, output_origin, flow_pred):
    T = output_origin.shape[1]
    pred_pcds_t = []
    free_id = len(occ_class_names) - 1
    occ_pred = np.array(sem_pred, copy=True)
    occ_pred[sem_pred < free_id] = 1
    occ_pred[sem_pred == free_id] = 0
    occ_pred = torch.from_numpy(occ_pred).permute(2, 1, 0)
    occ_pred = occ_pred[None, None, :].contiguous().float()
    offset = torch.Tensor(_pc_range[:3])[None, None, :]
    scaler = torch.Tensor([_voxel_size] * 3)[None, None, :]
    lidar_tindex = torch.zeros([1, lidar_rays.shape[0]])
    for t in range(T):
        lidar_origin = output_origin[:, t:t + 1, :]
        lidar_endpts = lidar_rays[None] + lidar_origin
        output_origin_render = ((lidar_origin - offset) / scaler).float()
        output_points_render = ((lidar_endpts - offset) / scaler).float()
        pred_dist, _, coord_index = render_forward(occ_pred, output_origin_render, output_points_render,
                                                   lidar_tindex, "test")
        pred_dist *= _voxel_size
        coord_index = coord_index[0, :, :].int()
        pred_flow = torch.from_numpy(flow_pred[coord_index[:, 0], coord_index[:, 1], coord_index[:, 2]])
        pred_label = torch.from_numpy(sem_pred[coord_index[:, 0], coord_index[:, 1], coord_index[:, 2]])[:, None]
        pred_dist = pred_dist[0, :, None]
        pred_pcds_t.append(torch.cat([pred_label.float(), pred_dist, pred_flow.float()], dim=-1))
    return torch.cat(pred_pcds_t, dim=0).numpy()


def calc_metrics(pcd_pred_list, pcd_gt_list):
    thresholds = [1, 2, 4]
    gt_cnt = np.zeros([len(occ_class_names)])
    pred_cnt = np.zeros([len(occ_class_names)])
    tp_cnt = np.zeros([len(thresholds), len(occ_class_names)])
    ave = np.zeros([len(thresholds), len(occ_class_names)])
    for i, cls in enumerate(occ_class_names):
        if cls not in flow_class_names:
            ave[:, i] = np.nan
    ave_count = np.zeros([len(thresholds), len(occ_class_names)])
    for pcd_pred, pcd_gt in zip(pcd_pred_list, pcd_gt_list):
        for j, threshold in enumerate(thresholds):
            l1_error = np.abs(pcd_pred[:, 1] - pcd_gt[:, 1])
            tp_dist_mask = (l1_error < threshold)
            for i, cls in enumerate(occ_class_names):
                cls_mask_pred = (pcd_pred[:, 0] == i)
                cls_mask_gt = (pcd_gt[:, 0] == i)
                if j == 0:
                    gt_cnt[i] += cls_mask_gt.sum()
                    pred_cnt[i] += cls_mask_pred.sum()
                tp_mask = np.logical_and(cls_mask_gt & cls_mask_pred, tp_dist_mask)
                tp_cnt[j][i] += tp_mask.sum()
                if cls in flow_class_names and tp_mask.sum() > 0:
                    flow_error = np.linalg.norm(pcd_gt[tp_mask, 2:4] - pcd_pred[tp_mask, 2:4], axis=1)
                    ave[j][i] += np.sum(flow_error)
                    ave_count[j][i] += flow_error.shape[0]
    with np.errstate(divide='ignore', invalid='ignore'):
        iou_list = [(tp_cnt[j] / (gt_cnt + pred_cnt - tp_cnt[j]))[:-1] for j in range(len(thresholds))]
        ave_list = ave[1][:-1] / ave_count[1][:-1]
    return iou_list, ave_list
