"""not-gpu: host-side logic of the path — query processing order, visibility packing, sample sharding
and the world_size-2 gloo run of the data-parallel plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from occnet_amd import synthetic
from occnet_amd.dist import shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('hw', [(200, 200), (50, 50), (400, 400), (12, 12), (7, 13)])
def test_bev_tile_order_is_a_permutation(hw):
    o = synthetic.bev_tile_order(*hw)
    assert o.dtype == np.int32 and np.array_equal(np.sort(o), np.arange(hw[0] * hw[1]))


def test_bev_tile_order_xcd_contiguity():
    """Hardware block b runs on XCD b % 8: the queries an XCD processes form one contiguous 1/8 of the
    tile sequence, i.e. a compact band of the BEV plane."""
    o = synthetic.bev_tile_order(200, 200).reshape(-1, 4)            # 4 waves (queries) per block
    rows = o // 200
    for x in range(8):
        r = rows[x::8]
        assert r.max() - r.min() < 200 // 8 + 8                      # one band of ~25 rows (+ tile)


def test_pack_vis_bits():
    from occnet_amd.plugin.spatial_cross_attention import pack_vis_bits
    g = torch.Generator().manual_seed(0)
    m = torch.rand(6, 2, 50, 8, generator=g) > 0.9
    bits = pack_vis_bits(m)
    assert bits.shape == (2, 50) and bits.dtype == torch.int32
    for c in range(6):
        assert torch.equal(((bits >> c) & 1).bool(), m[c].any(-1))


def test_rig_visibility_is_realistic():
    import oracle.model as om
    g = synthetic.BASE
    ref3d = om.get_reference_points(200, 200, 6.4, 8, '3d', 1)
    _, mask = om.point_sampling(ref3d, list(g['pc_range']), synthetic.make_img_metas(g))
    rows = [int(mask[c, 0].any(-1).sum()) for c in range(6)]
    assert 40000 < sum(rows) < 50000 and max(rows) < 11000           # R ~ 44.5 k


def test_shard_indices():
    assert shard_indices(10, 0, 4) == [0, 4, 8] and shard_indices(10, 3, 4) == [3, 7, 1]
    assert shard_indices(10, 1, 4, drop_last=True) == [1, 5]
    allidx = sorted(i for r in range(8) for i in shard_indices(64, r, 8))
    assert allidx == list(range(64))
    assert shard_indices(0, 0, 2) == []
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from occnet_amd import dist as od
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert od.env_world() == (rank, rank, world)
        mine = od.shard_indices(7, rank, world)
        # every rank "processes" its samples; no data-path collective — only timing is reduced
        elapsed = 1.0 + rank
        od.barrier()
        tmax = od.max_over_ranks(elapsed)
        thr = od.throughput(len(mine), elapsed)
        # DDP-style gradient all-reduce(mean): the one collective training needs
        g = torch.full((5,), float(rank + 1))
        dist.all_reduce(g)
        g /= world
        q.put((rank, mine, tmax, thr, g.tolist()))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 0]
    for r in res:
        assert r[2] == 2.0                      # max over ranks
        assert abs(r[3] - 2 * 4 / 2.0) < 1e-9   # all ranks' units / max time
        assert r[4] == [1.5] * 5


def test_folded_backbone_plan_on_cpu_matches_modules():
    """The BatchNorm fold and the plan's control flow (stem, bottlenecks with/without projection, FPN top-down,
    extra levels) on stock torch ops: fp32 on the CPU the folded plan equals the modules to rounding."""
    from occnet_amd.plugin.backbone import FPN, FusedInferenceBackbone, ResNet
    torch.manual_seed(0)
    bb = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_eval=True).eval()
    bb.init_weights()
    g = torch.Generator().manual_seed(1)
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    nk = FPN(in_channels=[512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_output',
             num_outs=4, relu_before_extra_convs=True).eval()
    x = torch.randn(1, 3, 64, 96, generator=g)
    with torch.no_grad():
        ref = nk(bb(x))
        plan = FusedInferenceBackbone(bb, nk, dtype=torch.float32, hip_tail=True)   # fp32: hip_tail switches off
        assert not plan.hip_tail and not plan._bneck and not plan._stem_fused
        out = plan(x)
    assert len(out) == len(ref) == 4
    for a, b in zip(ref, out):
        assert a.shape == b.shape
        rel = float((a - b).abs().max() / a.abs().max())
        assert rel < 2e-5, rel


def test_lazy_features_only_for_backbone_format_inputs():
    from occnet_amd.plugin.transformer_occ import LazyFeatures
    f32 = [torch.zeros(1, 6, 256, 4, 5)]
    bf16 = [torch.zeros(1, 6, 256, 4, 5, dtype=torch.bfloat16)]
    with torch.no_grad():
        assert not LazyFeatures.eligible(f32) and not LazyFeatures.eligible(bf16)     # host tensors: never
    assert not LazyFeatures.eligible([])
