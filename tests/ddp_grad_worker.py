"""Worker of tests/test_gpu_training.py::test_ddp_two_ranks_gradients_identical — launched by
torch.distributed.run with 2 ranks that SHARE cuda:0 (gloo process group: RCCL refuses two ranks on one device).

Each rank: the base model (ResNet-50 + FPN + 4 encoder layers) on small images and a 24x24x16 grid, DDP exactly
as occnet_amd.train.wrap_ddp configures it, its own sample (seed = rank), one train_step with and one without
the bf16 backbone autocast.  After each step every rank checksums its gradients; rank 0 prints one JSON line
with both ranks' checksums and the rank-local (un-reduced) gradient checksum for comparison."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from occnet_amd import synthetic
    from occnet_amd.plugin import Config, build_model, import_plugin
    from occnet_amd.train import make_optimizer, synthetic_targets, train_step, wrap_ddp
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "occ_base_200x200x16.py"))
    cfg.merge_from_dict({"model.pts_bbox_head.bev_h": 24, "model.pts_bbox_head.bev_w": 24,
                         "model.pts_bbox_head.positional_encoding.row_num_embed": 24,
                         "model.pts_bbox_head.positional_encoding.col_num_embed": 24,
                         "model.pts_bbox_head.transformer.rotate_center": [12, 12],
                         "model.pts_bbox_head.transformer.encoder.num_layers": 2})
    import_plugin(cfg)
    torch.manual_seed(0)                       # identical initial weights on both ranks
    model = build_model(cfg.model)
    model.init_weights()
    model = model.to(device).train()
    ddp = wrap_ddp(model, device)
    assert ddp is not model
    opt = make_optimizer(ddp, lr=0.0)          # gradients are what is checked; parameters stay put
    geo = dict(synthetic.BASE, img_h=96, img_w=160)
    img = synthetic.make_images(geo, batch=1, seed=rank, device=device)          # a different sample per rank
    metas = synthetic.make_img_metas(geo, batch=1, seed=rank, jitter=1.0)
    head = model.pts_bbox_head
    sem, flow, mask = synthetic_targets(head.bev_h, head.bev_w, head.transformer.pillar_h,
                                        num_classes=head.num_classes, seed=rank, device=device)
    out = {}
    import numpy as np

    def reseed():       # the same dropout masks (torch RNG) and GridMask draw (numpy RNG) in both passes
        torch.manual_seed(1000 + rank)
        torch.cuda.manual_seed(1000 + rank)
        np.random.seed(1000 + rank)
    for autocast in (False, True):
        reseed()
        train_step(ddp, opt, img, metas, sem, flow, mask, max_norm=1e9, autocast_backbone=autocast)
        g = torch.cat([p.grad.detach().float().reshape(-1) for p in model.parameters() if p.grad is not None])
        digest = torch.stack([g.double().sum(), g.double().abs().sum(), (g.double() ** 2).sum()])
        # the same sample WITHOUT the wrapper: the rank-local gradient
        model.zero_grad(set_to_none=True)
        model.backbone_autocast_dtype = torch.bfloat16 if autocast else None
        reseed()
        losses = model(return_loss=True, img_metas=metas, img=img, voxel_semantics=sem, voxel_flow=flow,
                       mask_camera=mask)
        sum(losses.values()).backward()
        gl = torch.cat([p.grad.detach().float().reshape(-1) for p in model.parameters() if p.grad is not None])
        local = torch.stack([gl.double().sum(), gl.double().abs().sum(), (gl.double() ** 2).sum()])
        both = [torch.zeros(6, dtype=torch.float64) for _ in range(2)]
        dist.all_gather(both, torch.cat([digest, local]).cpu())
        # mean of the two rank-local gradients, element-wise (what the all-reduce must have produced)
        mean_local = gl.clone()
        dist.all_reduce(mean_local)
        mean_local /= 2
        err = float((g - mean_local).abs().max() / mean_local.abs().max().clamp_min(1e-30))
        out["autocast" if autocast else "fp32"] = dict(
            ddp=[b[:3].tolist() for b in both], local=[b[3:].tolist() for b in both], n_grad=int(g.numel()),
            rel_err_vs_mean_of_local=err)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
