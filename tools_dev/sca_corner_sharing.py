#!/usr/bin/env python
"""How often do the 128 bilinear corner rows of one (camera-row, head) of the SCA gather coincide?  (VERDICT r2 weak #2:
"the one lever the builder names — corner-row sharing — has no measurement of how often it occurs".)

Host-side count on the bench's own inputs (base config, bench.py's weights: reference init + N(0, 0.02) on the
offset / weight Linears, seed 0 features): the oracle's SCA call is intercepted (TEST INFRASTRUCTURE, CPU) and for every
visible (camera, query) row and head the in-map corner rows (level, y, x) of its 32 samples x 4 corners are counted:
total in-map corners, distinct rows, distinct 128-byte lines (= rows: one head's 32 channels are one line), and
distinct (level, y) x-PAIRS (a horizontally adjacent corner pair = 2 lines 256 B apart in the (S, heads, 32) layout).
    python tools_dev/sca_corner_sharing.py [layers]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import oracle.model as om
    from occnet_amd import synthetic
    from occnet_amd.plugin import Config
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'occ_base_200x200x16.py'))
    hc = json.loads(json.dumps(dict(cfg.model.pts_bbox_head)))
    hc.pop('type')
    n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    hc['transformer']['encoder']['num_layers'] = n_layers
    hc['transformer']['encoder']['transformerlayers']['operation_order'] = tuple(
        hc['transformer']['encoder']['transformerlayers']['operation_order'])
    torch.manual_seed(0)
    ora = om.BEVFormerOccHead(**hc).eval()
    ora.init_weights()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in ora.named_parameters():
            if n.endswith('sampling_offsets.weight') or n.endswith('attention_weights.weight'):
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
    geo = dict(synthetic.BASE)
    feats = synthetic.make_features(geo, batch=1, seed=0)
    metas = synthetic.make_img_metas(geo, batch=1, seed=0)
    calls = []
    orig = om.multi_scale_deformable_attn_pytorch
    om.multi_scale_deformable_attn_pytorch = lambda *a: (calls.append(a), orig(*a))[1]
    masks = []
    orig_ps = om.point_sampling
    om.point_sampling = lambda *a, **k: (lambda r: (masks.append(r[1]), r)[1])(orig_ps(*a, **k))
    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        ora(feats, metas, only_bev=True)
    om.multi_scale_deformable_attn_pytorch = orig
    om.point_sampling = orig_ps
    bev_mask = masks[0]                                           # (cams, bs, Q, Z)
    vis = [m[0].sum(-1).nonzero().squeeze(-1) for m in bev_mask]  # visible queries per camera
    print(f"visible rows per camera {[len(v) for v in vis]}, R = {sum(len(v) for v in vis)}")
    for li in range(n_layers):
        value, shapes_t, loc, attn = calls[2 * li + 1]            # call 0 = TSA, 1 = SCA of each layer
        shapes = [tuple(int(v) for v in r) for r in shapes_t.tolist()]
        tot_in = tot_rows = tot_pairs = tot_quads = 0
        n_rh = 0
        hist = np.zeros(129, np.int64)
        for c in range(loc.shape[0]):
            L = loc[c, :len(vis[c])].numpy().astype(np.float64)   # (rows, heads, levels, points, 2) valid rows only
            rows, M, NL, P, _ = L.shape
            ids = []                                              # per level: (rows, M, P*4) corner ids or -1
            for l, (H, W) in enumerate(shapes):
                x = L[:, :, l, :, 0] * W - 0.5
                y = L[:, :, l, :, 1] * H - 0.5
                x0, y0 = np.floor(x), np.floor(y)
                ok = (x > -1) & (y > -1) & (x < W) & (y < H)      # mmcv admission test
                per = []
                for dy in (0, 1):
                    for dx in (0, 1):
                        xx, yy = x0 + dx, y0 + dy
                        inside = ok & (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
                        cid = (l * 100000 + yy * 400 + xx).astype(np.int64)
                        per.append(np.where(inside, cid, -1))
                ids.append(np.stack(per, -1).reshape(rows, M, P * 4))
            ids = np.concatenate(ids, -1)                          # (rows, M, 128)
            ids.sort(-1)
            valid = ids >= 0
            first = np.ones_like(valid)
            first[..., 1:] = ids[..., 1:] != ids[..., :-1]
            distinct = (valid & first).sum(-1)                     # (rows, M)
            # x-pairs: a row id and its right neighbour both present -> one 2-line pair
            pair_key = np.where(valid, ids // 2 * 2 + (ids // 400 % 1), -1)
            tot_in += int(valid.sum())
            tot_rows += int(distinct.sum())
            n_rh += rows * M
            np.add.at(hist, distinct.reshape(-1), 1)
        print(f"layer {li}: in-map corners {tot_in} ({tot_in / n_rh:.1f} per (row, head) of 128), distinct rows "
              f"{tot_rows} ({tot_rows / n_rh:.1f}); duplicates = {1 - tot_rows / tot_in:.3%} of the in-map corner loads")
        q = np.cumsum(hist) / hist.sum()
        print("   distinct rows per (row, head): median", int(np.searchsorted(q, 0.5)), " p10", int(np.searchsorted(q, 0.1)),
              " p90", int(np.searchsorted(q, 0.9)))


if __name__ == '__main__':
    main()
