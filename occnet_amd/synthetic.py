"""Seeded synthetic inputs of nuScenes shape: camera rig, img_metas, FPN features.

No dataset or checkpoint is reachable, so the bench and the parity tests use a nuScenes-nominal
6-camera pinhole rig (realistic per-camera visibility of the BEV grid) and N(0,1) feature maps.
img_metas carries exactly the keys the hot path consumes (reference:
projects/mmdet3d_plugin/bevformer/modules/encoder.py:94-101,133-134): `lidar2img` (6 x 4x4),
`ego2lidar` (4x4), `img_shape` [(H, W, 3)] * num_cams, and `can_bus` (only read with a history BEV).
"""
import math

import numpy as np

# yaw (deg), mount position (m), focal length (px) per camera: front, front-right, front-left,
# back, back-left, back-right
_RIG = (
    (0.0, (1.70, 0.02, 1.51), 1266.0),
    (-55.0, (1.55, -0.49, 1.50), 1266.0),
    (55.0, (1.52, 0.49, 1.51), 1266.0),
    (180.0, (0.03, 0.00, 1.58), 809.0),
    (108.0, (1.04, 0.48, 1.59), 1266.0),
    (-110.0, (1.01, -0.48, 1.57), 1266.0),
)
_CX, _CY = 816.0, 491.0

BASE = dict(name="bevformer_base_occ", num_cams=6, img_h=928, img_w=1600,
            feat_shapes=((116, 200), (58, 100), (29, 50), (15, 25)), bev_h=200, bev_w=200,
            pillar_h=16, num_points_in_pillar=8, embed_dims=256,
            pc_range=(-40.0, -40.0, -1.0, 40.0, 40.0, 5.4))
TINY = dict(name="bevformer_tiny_occ", num_cams=1, img_h=256, img_w=256,
            feat_shapes=((32, 32), (16, 16), (8, 8), (4, 4)), bev_h=50, bev_w=50,
            pillar_h=4, num_points_in_pillar=4, embed_dims=256,
            pc_range=(-40.0, -40.0, -1.0, 40.0, 40.0, 5.4))
HIRES = dict(BASE, name="bevformer_hires_occ", bev_h=400, bev_w=400, pillar_h=32)


def camera_matrix(yaw_deg, t, f, cx=_CX, cy=_CY):
    """lidar2img = K [R | -R t] for a pinhole camera looking along yaw in the ego x-y plane."""
    psi = math.radians(yaw_deg)
    R = np.array([[math.sin(psi), -math.cos(psi), 0.0],   # x: right
                  [0.0, 0.0, -1.0],                       # y: down
                  [math.cos(psi), math.sin(psi), 0.0]])   # z: forward
    K = np.array([[f, 0.0, cx], [0.0, f, cy], [0.0, 0.0, 1.0]])
    M = np.eye(4)
    M[:3, :3] = K @ R
    M[:3, 3] = -(K @ R) @ np.asarray(t, dtype=np.float64)
    return M


def make_img_metas(cfg=BASE, batch=1, seed=0, jitter=0.0):
    """One img_meta dict per batch element.  jitter > 0 perturbs the mounts per sample (seeded)."""
    rng = np.random.default_rng(seed)
    metas = []
    for _ in range(batch):
        mats = []
        for c in range(cfg["num_cams"]):
            yaw, t, f = _RIG[c % len(_RIG)]
            if cfg["num_cams"] == 1:   # tiny config: one square front camera
                f, cx, cy = 0.8 * cfg["img_w"], cfg["img_w"] / 2.0, cfg["img_h"] / 2.0
            else:
                cx, cy = _CX, _CY
            if jitter:
                yaw = yaw + rng.normal(0.0, jitter)
                t = tuple(np.asarray(t) + rng.normal(0.0, 0.01 * jitter, 3))
            mats.append(camera_matrix(yaw, t, f, cx, cy))
        metas.append(dict(
            lidar2img=[m.copy() for m in mats],
            ego2lidar=np.eye(4),
            img_shape=[(cfg["img_h"], cfg["img_w"], 3)] * cfg["num_cams"],
            can_bus=np.zeros(18),
            prev_bev_exists=False,
        ))
    return metas


def make_features(cfg=BASE, batch=1, seed=0, device="cpu", dtype=None):
    """FPN outputs: list of L tensors (B, num_cams, C, h, w) ~ N(0,1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    feats = []
    for (h, w) in cfg["feat_shapes"]:
        x = torch.randn((batch, cfg["num_cams"], cfg["embed_dims"], h, w), generator=g,
                        dtype=torch.float32)
        feats.append(x.to(device=device, dtype=dtype or torch.float32))
    return feats


def make_images(cfg=BASE, batch=1, seed=0, device="cpu"):
    """(B, num_cams, 3, H, W) ~ N(0,1)*57 — mean-subtracted BGR at img_norm_cfg std=1."""
    import torch
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.randn((batch, cfg["num_cams"], 3, cfg["img_h"], cfg["img_w"]), generator=g) * 57.0
    return x.to(device)


def bev_tile_order(bev_h, bev_w, tile_h=8, tile_w=8, n_xcd=8, waves_per_block=4, patch=None):
    """Processing order of the BEV queries for the gather kernels (a permutation of arange(H*W)).

    patch = (ph, pw): inside a full tile the queries are emitted patch by patch (ph * pw consecutive entries = one patch).
    Queries are visited in tile_h x tile_w tiles (neighbouring pillars project to neighbouring
    pixels, so a tile's samples share cache lines).  The hardware dispatches block b to XCD b % 8,
    each XCD with a private L2: the tile sequence is dealt so that every XCD walks one contiguous
    1/8 of the tile list instead of every 8th tile.  Only locality depends on this, never results.
    """
    q = np.arange(bev_h * bev_w, dtype=np.int64).reshape(bev_h, bev_w)
    tiles = []
    for y0 in range(0, bev_h, tile_h):
        xs = range(0, bev_w, tile_w)
        if (y0 // tile_h) % 2:
            xs = reversed(list(xs))        # boustrophedon: consecutive tiles stay adjacent
        for x0 in xs:
            tile = q[y0:y0 + tile_h, x0:x0 + tile_w]
            if patch is not None and tile.shape == (tile_h, tile_w):
                # the tile as (ph x pw) patches of 8 queries: a patch is what ONE wave of the head-major SCA gather owns
                ph, pw = patch
                tile = tile.reshape(tile_h // ph, ph, tile_w // pw, pw).transpose(0, 2, 1, 3)
            tiles.append(tile.reshape(-1))
    flat = np.concatenate(tiles)
    n = flat.size
    nblk = (n + waves_per_block - 1) // waves_per_block
    pad = nblk * waves_per_block - n
    blocks = np.concatenate([flat, -np.ones(pad, np.int64)]).reshape(nblk, waves_per_block)
    # logical block j (spatial order) -> hardware block id: XCD x gets logical blocks
    # [x*per, (x+1)*per).  hardware id b runs on XCD b % n_xcd as its (b // n_xcd)-th block.
    per = (nblk + n_xcd - 1) // n_xcd
    hw = -np.ones((per * n_xcd, waves_per_block), np.int64)
    j = np.arange(nblk)
    hw_id = (j % per) * n_xcd + (j // per)
    hw[hw_id] = blocks
    out = hw.reshape(-1)
    out = out[out >= 0]
    if out.size != n:                       # holes moved valid entries past n: fall back
        out = flat
    return out.astype(np.int32)
