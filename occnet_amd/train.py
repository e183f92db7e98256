"""Data-parallel training step of the occ model on the MI355X path.

Mirrors what the reference's driver does per iteration (P/bevformer/apis/mmdet_train.py:71-126 with
C/bevformer/bevformer_base_occ.py:214-223): DDP over one process per GPU (`broadcast_buffers=False,
find_unused_parameters=False`), AdamW lr 2e-4 / weight decay 0.01 with the image backbone at lr x0.1,
gradient clipping at max_norm 35, loss = loss_occ + loss_flow.  The ONLY collective is DDP's bucketed
gradient all-reduce (backend string 'nccl' = RCCL over xGMI); the reference's scalar-loss logging
all-reduce (mmdet `_parse_losses`) is dropped: losses are logged rank-locally.

Autograd through the hot path takes the reference-shaped decomposition with the deformable attention
going through MultiScaleDeformableAttnFunction_fp32 — the HIP forward and backward kernels
(occ_ms_deform_attn_forward_f32 / occ_ms_deform_attn_backward_ws_f32) — and, around it, this repository's
training kernels behind autograd Functions: encoder / head Linears on ext.LinearX3Function (forward + dx on
linear_bf16x3, dW/db on linear_wgrad), the SCA rebatch / scatter-back on ext.RowsGatherSumFunction, the
norm_eval backbone as folded convolutions (plugin/backbone.py::conv_bn_folded) with its frozen stages on the
inference-plan kernels.  DESIGN.md §9 lists what each of these bought on MI355X (137 -> 52.7 ms per sample).
"""
import torch
import torch.distributed as dist


def make_optimizer(model, lr=2e-4, weight_decay=0.01, backbone_lr_mult=0.1):
    """AdamW with paramwise_cfg custom_keys {'img_backbone': lr_mult 0.1} (bevformer_base_occ.py:214-221)."""
    backbone, rest = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (backbone if 'img_backbone' in name else rest).append(p)
    groups = [dict(params=rest, lr=lr)]
    if backbone:
        groups.append(dict(params=backbone, lr=lr * backbone_lr_mult))
    # fused: one multi-tensor kernel per parameter group instead of a dozen foreach launches per step
    fused = all(p.is_cuda for g in groups for p in g['params'])
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=fused)


def wrap_ddp(model, device):
    """DDP exactly as the reference configures it (mmdet_train.py:71-79)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if not (dist.is_available() and dist.is_initialized()):
        return model
    ddp = DDP(model, device_ids=[device.index] if device.type == 'cuda' else None, broadcast_buffers=False,
              find_unused_parameters=False)
    # the constructor broadcasts rank 0's parameters into the others through .data (no _version bump)
    from . import invalidate_caches
    invalidate_caches()
    return ddp


def synthetic_targets(bev_h, bev_w, pillar_h, num_classes=18, batch=1, seed=0, device='cpu'):
    """Occupancy ground truth of the on-disk shapes (loading.py:21-33): semantics (B, W, H, Z) uint8
    class ids, flow (B, W, H, Z, 2) float32, camera mask (B, W, H, Z) bool."""
    g = torch.Generator().manual_seed(seed + 31)
    sem = torch.randint(0, num_classes, (batch, bev_w, bev_h, pillar_h), generator=g, dtype=torch.int64)
    flow = torch.randn((batch, bev_w, bev_h, pillar_h, 2), generator=g)
    mask = torch.rand((batch, bev_w, bev_h, pillar_h), generator=g) > 0.3
    return sem.to(torch.uint8).to(device), flow.to(device), mask.to(device)


def train_step(model, optimizer, img, img_metas, voxel_semantics, voxel_flow, mask_camera,
               max_norm=35.0, autocast_backbone=False):
    """One optimisation step; returns the rank-local loss dict (python floats are NOT synchronised:
    call .item() on the values only when logging)."""
    net = model.module if hasattr(model, 'module') else model
    optimizer.zero_grad(set_to_none=True)
    # the stock MIOpen backbone may run in bf16 (the hand-written hot path stays fp32): a detector attribute,
    # so that the WHOLE step goes through model(...) — i.e. through DDP.forward, which arms the reducer; a
    # forward that bypasses the wrapper leaves the gradients un-reduced (each rank would diverge silently)
    net.backbone_autocast_dtype = torch.bfloat16 if autocast_backbone else None
    losses = model(return_loss=True, img_metas=img_metas, img=img, voxel_semantics=voxel_semantics,
                   voxel_flow=voxel_flow, mask_camera=mask_camera)
    loss = sum(losses.values())
    loss.backward()          # DDP all-reduces the gradient buckets here (overlapped with backward)
    params = [p for p in net.parameters() if p.grad is not None]
    torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=2)
    optimizer.step()
    return losses
