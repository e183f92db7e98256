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
    for name, case in FULL_CASES.items():
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


if __name__ == '__main__':
    if '--full-only' in sys.argv:
        fullsize_golden()
        sys.exit(0)
    if '--metrics-only' not in sys.argv:
        main()
        fullsize_golden()
    metric_golden()
