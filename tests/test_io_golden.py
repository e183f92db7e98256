"""Input / output formats pinned to the reference (VERDICT r2 missing #1).

tests/golden/datasets.npz was written by oracle/gen_golden.py::datasets_golden running the reference's OWN
  projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py   NormalizeMultiviewImage :65-101, PadMultiViewImage :12-62
  projects/mmdet3d_plugin/datasets/pipelines/loading.py        LoadOccGTFromFile :7-38
  projects/mmdet3d_plugin/datasets/nuscenes_occ.py             NuSceneOcc.get_data_info :49-126, format_results :189-257
  tools/ray_iou/ego_pose_extractor.py                          EgoPoseDataset.__getitem__ :84-121
in place (oracle/refshim.install_datasets; third-party leaves restated there: mmcv.impad_to_multiple / imnormalize,
pyquaternion, nuscenes transform_matrix) on the seeded inputs of tests/golden_cases.py.  Checked against it:
  not gpu   occnet_amd/io.py: normalise + pad (bit-exact), img_shape / ori_shape / pad_shape metas, camera matrices
            and ego2lidar (bit-exact), lidar origins, occupancy GT loader; the submission content through the C-oracle
            ray caster (bit-exact per-ray class / distance / flow in the file's int8 / float16 types);
  gpu       the uint8 stem kernel's fused normalise + pad == the stem run on the REFERENCE pipeline's float frames
            (bit-exact), and io.format_submission on the HIP ray caster == the reference's submission content.
"""
import os

import numpy as np
import pytest
import torch

from occnet_amd import io as oio
from tests.golden_cases import PIPELINE_CASES, dataset_infos, metric_scene, pipeline_images

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'datasets.npz')


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(GOLD))


@pytest.mark.parametrize('name', sorted(PIPELINE_CASES))
def test_normalize_pad_match_reference_pipeline(gold, name):
    case = PIPELINE_CASES[name]
    raw = pipeline_images(case)
    normed, cfg = oio.normalize_multiview([r.astype(np.float32) for r in raw], case['mean'], case['std'],
                                          to_rgb=case['to_rgb'])
    padded, meta = oio.pad_multiview(normed, size_divisor=32)
    want = gold[f'{name}_img']
    got = np.stack(padded)
    assert got.dtype == want.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    for key in ('img_shape', 'ori_shape', 'pad_shape'):
        assert np.array_equal(np.asarray(meta[key]), gold[f'{name}_{key}']), key
    # uint8 frames straight in (what the device path does) give the same floats
    normed_u8, _ = oio.normalize_multiview(raw, case['mean'], case['std'], to_rgb=case['to_rgb'])
    assert np.array_equal(np.stack(oio.pad_multiview(normed_u8, size_divisor=32)[0]), want)


def test_camera_matrices_match_reference_dataset(gold):
    infos = dataset_infos()['infos']
    for i, info in enumerate(infos):
        meta = oio.make_img_meta(list(info['cams'].values()), info['lidar2ego_translation'],
                                 info['lidar2ego_rotation'], [(928, 1600, 3)] * 6)
        for key in ('lidar2img', 'lidar2cam', 'cam_intrinsic'):
            got, want = np.stack(meta[key]), gold[f'info{i}_{key}']
            assert got.dtype == want.dtype and np.array_equal(got, want), (i, key, float(np.abs(got - want).max()))
        assert np.array_equal(meta['ego2lidar'], gold[f'info{i}_ego2lidar'])
        tok, org = oio.lidar_origins(infos, i)
        assert tok == info['token']
        want = gold[f'info{i}_origins']
        assert tuple(org.shape) == (1,) + want.shape and org.dtype == torch.float64
        assert np.array_equal(org[0].numpy(), want), float(np.abs(org[0].numpy() - want).max())


def test_occ_gt_loader_matches_reference(gold, tmp_path):
    sp, sg, fp, fg, _ = metric_scene(43)
    path = str(tmp_path / 'labels.npz')
    oio.save_occ_gt(path, sg, fg)
    sem, flow = oio.load_occ_gt(path)
    assert int(sem.astype(np.int64).sum()) == int(gold['gt_semantics_sum'])
    assert float(np.abs(flow.astype(np.float64)).sum()) == float(gold['gt_flow_abs_sum'])
    s0, f0 = oio.load_occ_gt(str(tmp_path / 'missing.npz'))
    assert np.array_equal(np.asarray(s0.shape), gold['gt_missing_semantics_shape'])
    assert np.array_equal(np.asarray(f0.shape), gold['gt_missing_flow_shape'])
    assert [str(s0.dtype), str(f0.dtype)] == list(gold['gt_missing_dtypes'])


def _check_submission(gold, sub):
    assert sorted(k for k in sub if k != 'results') == list(gold['sub_header_keys'])
    assert list(sub['results']) == list(gold['sub_tokens'])
    for tok, r in sub['results'].items():
        assert sorted(r) == ['pcd_cls', 'pcd_dist', 'pcd_flow']
        for k, v in r.items():
            want = gold[f'sub_{tok}_{k}']
            assert v.dtype == want.dtype and v.shape == want.shape, (tok, k, v.dtype, want.dtype)
            assert np.array_equal(v, want), (tok, k, int((v != want).sum()))


def _samples():
    infos = dataset_infos()['infos']
    for i, info in enumerate(infos):
        sp, _, fp, _, _ = metric_scene(44 + i)
        tok, org = oio.lidar_origins(infos, i)
        yield tok, sp.astype(np.int64), fp, org


def test_submission_content_matches_reference_with_oracle_caster(gold, tmp_path, monkeypatch):
    """io.format_submission's layout / casts / origins with the C-oracle ray caster standing in for the HIP kernel
    (the device caster itself is held to the same bits in the gpu test below and in tests/test_gpu_dvr.py)."""
    from oracle import ray_metrics_ref as oref
    try:
        oref.dvr_lib()
    except FileNotFoundError:
        pytest.skip("oracle/_build/libdvr_ref.so not built (make -C oracle)")
    from occnet_amd import metrics
    monkeypatch.setattr(metrics.ray_metrics, '_render',
                        lambda occ, org, pts, tidx, device: oref.render_forward(occ, org, pts, tidx, "test"))
    path = oio.format_submission(_samples(), str(tmp_path / 'sub'), device='cpu')
    _check_submission(gold, oio.read_submission(path))


@pytest.mark.gpu
def test_submission_matches_reference_on_hip_caster(gold, tmp_path):
    path = oio.format_submission(_samples(), str(tmp_path / 'sub'))
    _check_submission(gold, oio.read_submission(path))


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(PIPELINE_CASES))
def test_u8_stem_equals_stem_on_reference_pipeline_frames(gold, name):
    """The stem kernel fed with the RAW uint8 frames (normalise + pad fused into its tile staging) is BIT-IDENTICAL to
    the float stem kernel fed with the frames the reference's own NormalizeMultiviewImage + PadMultiViewImage
    produced (the golden) in DefaultFormatBundle3D's (N, 3, H, W) layout."""
    from occnet_amd import ext
    case = PIPELINE_CASES[name]
    raw = pipeline_images(case)
    ref_frames = gold[f'{name}_img']                                           # (N, H, W, 3) float32, reference-made
    x = torch.from_numpy(np.ascontiguousarray(ref_frames.transpose(0, 3, 1, 2))).cuda()
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).cuda()
    b = torch.randn(64, generator=g).cuda()
    frag = ext.stem_pack_weight(w)
    want = ext.stem_conv7x7_pool(x, frag, b)
    got, hw = ext.stem_conv7x7_pool_u8(torch.from_numpy(np.stack(raw)).cuda(), frag, b, case['mean'], case['std'],
                                       to_rgb=case['to_rgb'])
    assert hw == tuple(x.shape[2:]) and got.shape == want.shape
    assert torch.equal(got, want)
