#!/usr/bin/env python
"""Dev probe: fwd+bwd time of the decoder's Conv3d layers (16->32->32, k3, 16x200x200) under different
dtypes / memory formats on MIOpen — informs the training path (occnet_amd/train.py)."""
import time
import torch
import torch.nn as nn

torch.manual_seed(0)
dev = "cuda"
for dtype, fmt in ((torch.float32, "contig"), (torch.float32, "cl3d"), (torch.bfloat16, "contig"),
                   (torch.bfloat16, "cl3d"), (torch.float16, "cl3d")):
    net = nn.Sequential(nn.Conv3d(16, 32, 3, padding=1, bias=False), nn.BatchNorm3d(32), nn.ReLU(),
                        nn.Conv3d(32, 32, 3, padding=1, bias=False), nn.BatchNorm3d(32), nn.ReLU()).to(dev).to(dtype)
    x = torch.randn(1, 16, 16, 200, 200, device=dev, dtype=dtype, requires_grad=True)
    if fmt == "cl3d":
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    try:
        t0 = time.perf_counter()
        for _ in range(2):
            net(x).sum().backward()
        torch.cuda.synchronize()
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(3):
            net(x).sum().backward()
        torch.cuda.synchronize()
        print(f"conv3d fwd+bwd {dtype} {fmt}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms (warmup {warm:.1f} s)", flush=True)
    except Exception as e:
        print(f"conv3d fwd+bwd {dtype} {fmt}: FAILED {e!r}"[:200], flush=True)
