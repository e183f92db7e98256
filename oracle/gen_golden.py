#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE'S OWN module files
(from /root/reference, under the stubs of oracle/refshim.py) on seeded synthetic inputs — TEST
INFRASTRUCTURE.  Run in the build container only (the GPU box has no /root/reference):

    python -m oracle.gen_golden

For each case: weights = tests.util.randomize(seed) of the shared state_dict layout, inputs =
occnet_amd.synthetic (seeded), outputs of the reference head / of its individual attention modules
are stored as float32 .npz together with checksums of the seeded weights and inputs (so a change of
the RNG stream is detected instead of silently comparing different problems).
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from tests.golden_cases import (CASES, FULL_CASES, FULL_KEYS, METRIC_SEEDS, case_inputs, checksum, digest,  # noqa: E402
                                full_case_inputs, metric_scene)


def metric_golden():
    """RayIoU / mAVE / OccScore fixtures from the reference's own ray_metrics.py (process_one_sample :89-143,
    calc_metrics :146-197, main :200-257) and tools/ray_iou/metric.py (calc_metrics :6-81), both executed in place
    with the reference's ray-casting kernel compiled for the host (oracle/build_ref.py)."""
    import contextlib
    import io
    from oracle import refshim
    ns = refshim.install_metrics()
    print('reference metric files executed:')
    for f in ns.files:
        print('  ', f)
    rm = ns.ray_metrics
    scenes = [metric_scene(s) for s in METRIC_SEEDS]
    lidar_rays = torch.from_numpy(rm.generate_lidar_rays())
    arrays = {'lidar_rays': lidar_rays.numpy()}
    pcd_pred_list, pcd_gt_list = [], []
    with ns.on_host():
        for i, (sp, sg, fp, fg, org) in enumerate(scenes):
            pcd_pred = rm.process_one_sample(sp, lidar_rays, org, fp)          # unmasked: every ray of every origin
            pcd_gt = rm.process_one_sample(sg, lidar_rays, org, fg)
            arrays[f'pcd_pred_{i}'], arrays[f'pcd_gt_{i}'] = pcd_pred, pcd_gt
            valid = pcd_gt[:, 0].astype(np.int32) != len(rm.occ_class_names) - 1
            pcd_pred_list.append(pcd_pred[valid])
            pcd_gt_list.append(pcd_gt[valid])
        with np.errstate(divide='ignore', invalid='ignore'):
            iou_list, ave_list = rm.calc_metrics(pcd_pred_list, pcd_gt_list)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                rm.main([s[0].reshape(-1) for s in scenes], [s[1].reshape(-1) for s in scenes],
                        [s[2].reshape(-1) for s in scenes], [s[3].reshape(-1) for s in scenes],
                        [s[4] for s in scenes])
            score_line = [l for l in buf.getvalue().splitlines() if 'Occ score' in l][-1]
            occ_score = float(score_line.split(':')[-1])
            # tools/ray_iou/metric.py on the per-ray (class, distance, flow) lists of a submission file
            m_iou, m_ave = ns.metric.calc_metrics(
                [p[:, 0] for p in arrays_lists(arrays, 'pcd_pred')], [p[:, 1] for p in arrays_lists(arrays, 'pcd_pred')],
                [p[:, 2:4] for p in arrays_lists(arrays, 'pcd_pred')], [p[:, 0] for p in arrays_lists(arrays, 'pcd_gt')],
                [p[:, 1] for p in arrays_lists(arrays, 'pcd_gt')], [p[:, 2:4] for p in arrays_lists(arrays, 'pcd_gt')])
    arrays.update(iou=np.stack(iou_list), ave=np.asarray(ave_list), occ_score=np.float64(occ_score),
                  miou=np.float64(np.nanmean(iou_list)), mave=np.float64(np.nanmean(ave_list)),
                  metric_py_iou=np.stack(m_iou), metric_py_ave=np.asarray(m_ave))
    arrays.pop('lidar_rays')
    path = os.path.join(OUT, 'ray_metrics.npz')
    np.savez_compressed(path, **arrays)
    print(f'ray_metrics: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); RayIoU {arrays["miou"]:.4f} '
          f'mAVE {arrays["mave"]:.4f} OccScore {occ_score:.4f}')


def arrays_lists(arrays, prefix):
    return [arrays[f'{prefix}_{i}'] for i in range(len(METRIC_SEEDS))]


def main():
    from oracle import refshim
    ref = refshim.install()
    print('reference files executed:')
    for f in ref.files:
        print('  ', f)
    from tests.util import head_cfg, randomize
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    for name, case in CASES.items():
        g = case['geometry']
        cfg = head_cfg(g)
        head = ref.build_head(copy.deepcopy(cfg))
        assert type(head).__module__.startswith('projects.mmdet3d_plugin'), type(head).__module__
        randomize(head, case['seed'])
        head.eval()
        feats, metas, prev_bev = case_inputs(case)
        taps = {}

        def tap(label):
            def hook(mod, args, out):
                taps[label] = out.detach().clone()
            return hook
        layer0 = head.transformer.encoder.layers[0]
        hs = [layer0.attentions[0].register_forward_hook(tap('layer0_tsa_out')),
              layer0.attentions[1].register_forward_hook(tap('layer0_sca_out'))]
        with torch.no_grad():
            out = head(feats, metas, prev_bev=None if prev_bev is None else prev_bev.clone())
        for h in hs:
            h.remove()
        arrays = dict(
            bev_embed=out['bev_embed'].numpy(), occ=out['occ'].numpy(), flow=out['flow'].numpy(),
            weights_checksum=np.float64(checksum(head.state_dict().values())),
            inputs_checksum=np.float64(checksum(feats)),
            **{k: v.numpy() for k, v in taps.items()})
        path = os.path.join(OUT, f'{name}.npz')
        np.savez_compressed(path, **arrays)
        print(f'{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); '
              f'occ {arrays["occ"].shape} mean|occ| {np.abs(arrays["occ"]).mean():.4f}')


def fullsize_golden():
    """The reference's own module files at the BENCHMARKED geometry (BASELINE.json configs[1] / configs[2]): 40 000
    BEV queries, 6 cameras x 30 825 keys, max_len ~ 9 900 padded rows per camera (spatial_cross_attention.py:136-173),
    one encoder layer, without and with a rotated history BEV (transformer_occ.py:189-205,
    temporal_self_attention.py:177-204).  Stored: tests.golden_cases.digest of every output (strided subsample +
    float64 sums), a few MB of RAM and ~1 minute of CPU per case."""
    import time
    from oracle import refshim
    ref = refshim.install()
    from tests.util import head_cfg, randomize
    torch.set_num_threads(os.cpu_count() or 1)
    only = [a for a in sys.argv[1:] if a in FULL_CASES]          # `--full-only name ...`: just these cases
    for name, case in FULL_CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        cfg = head_cfg(case['geometry'])
        head = ref.build_head(copy.deepcopy(cfg))
        assert type(head).__module__.startswith('projects.mmdet3d_plugin'), type(head).__module__
        randomize(head, case['seed'])
        head.eval()
        feats, metas, prev_bev = full_case_inputs(case)
        taps = {}
        layer0 = head.transformer.encoder.layers[0]
        hs = [layer0.attentions[0].register_forward_hook(
                  lambda m, a, o: taps.__setitem__('layer0_tsa_out', o.detach().clone())),
              layer0.attentions[1].register_forward_hook(
                  lambda m, a, o: taps.__setitem__('layer0_sca_out', o.detach().clone()))]
        with torch.no_grad():
            out = head(feats, metas, prev_bev=None if prev_bev is None else prev_bev.clone())
        for h in hs:
            h.remove()
        out = dict(out, **taps)
        arrays = dict(weights_checksum=np.float64(checksum(head.state_dict().values())),
                      inputs_checksum=np.float64(checksum(feats)))
        for k in FULL_KEYS:
            arrays.update({f'{k}_{kk}': v for kk, v in digest(out[k]).items()})
            arrays[f'{k}_shape'] = np.asarray(out[k].shape, np.int64)
        path = os.path.join(OUT, f'{name}.npz')
        np.savez_compressed(path, **arrays)
        print(f'{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB) in {time.time() - t0:.0f} s; '
              f'occ {tuple(out["occ"].shape)} mean|occ| {float(out["occ"].abs().mean()):.4f}')


def datasets_golden():
    """Input / output FORMAT fixtures from the reference's own dataset / pipeline files run in place
    (refshim.install_datasets): NormalizeMultiviewImage + PadMultiViewImage on seeded uint8 frames
    (transform_3d.py:31-45,82-94), LoadOccGTFromFile (loading.py:21-33), NuSceneOcc.get_data_info
    (nuscenes_occ.py:49-126: lidar2img / lidar2cam / cam_intrinsic / ego2lidar), EgoPoseDataset origins and
    NuSceneOcc.format_results (nuscenes_occ.py:189-257: the submission.gz content)."""
    import pickle
    import tempfile
    from oracle import refshim
    from occnet_amd import io as oio
    from tests.golden_cases import PIPELINE_CASES, dataset_infos, metric_scene, pipeline_images
    ns = refshim.install_datasets()
    print('reference dataset / pipeline files executed:')
    for f in ns.files:
        print('  ', f)
    arrays = {}
    for name, case in PIPELINE_CASES.items():
        results = dict(img=[im.astype(np.float32) for im in pipeline_images(case)])   # LoadMultiViewImageFromFiles(to_float32)
        results = ns.NormalizeMultiviewImage(mean=case['mean'], std=case['std'], to_rgb=case['to_rgb'])(results)
        results = ns.PadMultiViewImage(size_divisor=32)(results)
        arrays[f'{name}_img'] = np.stack(results['img'])
        arrays[f'{name}_img_shape'] = np.asarray(results['img_shape'])
        arrays[f'{name}_ori_shape'] = np.asarray(results['ori_shape'])
        arrays[f'{name}_pad_shape'] = np.asarray(results['pad_shape'])
    with tempfile.TemporaryDirectory() as tmp:
        # occupancy ground-truth file, read by the reference's loader (present and missing)
        sp, sg, fp, fg, _ = metric_scene(43)
        gt_path = os.path.join(tmp, 'labels.npz')
        oio.save_occ_gt(gt_path, sg, fg)
        res = ns.LoadOccGTFromFile()(dict(occ_path=gt_path))
        arrays['gt_semantics_sum'] = np.int64(res['voxel_semantics'].astype(np.int64).sum())
        arrays['gt_flow_abs_sum'] = np.float64(np.abs(res['voxel_flow'].astype(np.float64)).sum())
        assert np.array_equal(res['voxel_semantics'], sg) and np.array_equal(res['voxel_flow'], fg)
        miss = ns.LoadOccGTFromFile()(dict(occ_path=os.path.join(tmp, 'missing.npz')))
        arrays['gt_missing_semantics_shape'] = np.asarray(miss['voxel_semantics'].shape)
        arrays['gt_missing_flow_shape'] = np.asarray(miss['voxel_flow'].shape)
        arrays['gt_missing_dtypes'] = np.asarray([str(miss['voxel_semantics'].dtype), str(miss['voxel_flow'].dtype)])
        # dataset: annotation file -> get_data_info -> camera matrices; format_results -> submission.gz
        ann = os.path.join(tmp, 'infos.pkl')
        with open(ann, 'wb') as f:
            pickle.dump(dataset_infos(), f)
        ds = ns.NuSceneOcc(ann_file=ann, data_root=tmp, modality=dict(use_camera=True), test_mode=True)
        for i in range(len(ds.data_infos)):
            d = ds.get_data_info(i)
            arrays[f'info{i}_lidar2img'] = np.stack(d['lidar2img'])
            arrays[f'info{i}_lidar2cam'] = np.stack(d['lidar2cam'])
            arrays[f'info{i}_cam_intrinsic'] = np.stack(d['cam_intrinsic'])
            arrays[f'info{i}_ego2lidar'] = np.asarray(d['ego2lidar'])
            tok, org = ns.EgoPoseDataset(ds.data_infos, dataset_type='openocc_v2')[i]
            arrays[f'info{i}_origins'] = org.numpy()
        occ_results = []
        for i in range(len(ds.data_infos)):
            sp, _, fp, _, _ = metric_scene(44 + i)
            occ_results.append(dict(occ_results=torch.from_numpy(sp.astype(np.int64)).reshape(1, 200, 200, 16),
                                    flow_results=torch.from_numpy(fp).reshape(1, 200, 200, 16, 2)))
        # num_workers=8 in the reference's DataLoader: forked workers inherit the stub modules
        with ns.on_host():
            ds.format_results(occ_results, os.path.join(tmp, 'sub'))
        sub = oio.read_submission(os.path.join(tmp, 'sub', 'submission.gz'))
        arrays['sub_header_keys'] = np.asarray(sorted(k for k in sub if k != 'results'))
        arrays['sub_tokens'] = np.asarray(list(sub['results']))
        for tok, r in sub['results'].items():
            for k, v in r.items():
                arrays[f'sub_{tok}_{k}'] = v
    path = os.path.join(OUT, 'datasets.npz')
    np.savez_compressed(path, **arrays)
    print(f'datasets: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    if '--datasets-only' in sys.argv:
        datasets_golden()
        sys.exit(0)
    if '--full-only' in sys.argv:
        fullsize_golden()
        sys.exit(0)
    if '--metrics-only' not in sys.argv:
        main()
        fullsize_golden()
        datasets_golden()
    metric_golden()
