"""Dev probe: what bounds ms_deform_attn_backward?  Same launch (6 cameras x 9000 queries x 8 heads x 4 levels x
8 points, base feature maps) with (a) random sampling locations, (b) smooth locations as the SCA produces them
(neighbouring queries sample neighbouring pixels), (c) every sample on ONE pixel per level (maximal contention),
(d) every sample outside the maps (no atomics at all)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from occnet_amd import ext
from tools_dev.conv_probe import timeit

g = torch.Generator().manual_seed(0)
B, M, D, L, P, Lq = 6, 8, 32, 4, 8, 9000
hw = [(116, 200), (58, 100), (29, 50), (15, 25)]
S = sum(h * w for h, w in hw)
shapes = torch.tensor(hw, dtype=torch.long).cuda()
starts = torch.tensor([0, 23200, 29000, 30450], dtype=torch.long).cuda()
value = torch.randn(B, S, M, D, generator=g).cuda()
aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P).cuda()
go = torch.randn(B, Lq, M * D, generator=g).cuda()
qx = (torch.arange(Lq) % 200).float() / 200.0
qy = (torch.arange(Lq) // 200).float() / 45.0
smooth = torch.stack([qx, qy], -1).view(1, Lq, 1, 1, 1, 2).expand(B, Lq, M, L, P, 2)
cases = {
    "random": torch.rand(B, Lq, M, L, P, 2, generator=g),
    "smooth (+-2% jitter)": (smooth + 0.02 * torch.randn(B, Lq, M, L, P, 2, generator=g)).clamp(0.01, 0.99),
    "one pixel": torch.full((B, Lq, M, L, P, 2), 0.5),
    "outside": torch.full((B, Lq, M, L, P, 2), 3.0),
}
gv, gl, ga = torch.zeros_like(value), torch.zeros(B, Lq, M, L, P, 2).cuda(), torch.zeros_like(aw)
for name, loc in cases.items():
    loc = loc.contiguous().cuda()
    us = timeit(lambda: ext.ms_deform_attn_backward(value, shapes, starts, loc, aw, go, gv, gl, ga, im2col_step=64), iters=5)
    n_at = B * Lq * M * L * P * 4 * D
    print(f"{name:22s}: {us / 1e3:7.2f} ms   ({n_at / us * 1e-3:6.1f} G dword atomics/s if all corners were inside)")
us = timeit(lambda: ext.ms_deform_attn_forward(value, shapes, starts, cases['random'].cuda(), aw, im2col_step=64), iters=5)
print(f"forward (random)      : {us / 1e3:7.2f} ms")
