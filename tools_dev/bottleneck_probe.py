"""Dev probe: time the whole-bottleneck kernel on the ResNet-50 layer1 shape (6 x 232 x 400) against the
HBM floor (x read once + out written once)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from occnet_amd import ext

def run(cin, ds, iters=20):
    g = torch.Generator().manual_seed(0)
    N, H, W = 6, 232, 400
    x = torch.randn(N, cin, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    mk = lambda *s: torch.randn(*s, generator=g).cuda() * 0.05
    pack = ext.bottleneck64_pack(mk(64, cin, 1, 1), mk(64), mk(64, 64, 3, 3), mk(64), mk(256, 64, 1, 1), mk(256),
                                 mk(256, cin, 1, 1) if ds else None, mk(256) if ds else None)
    for _ in range(3):
        ext.bottleneck64_nhwc(x, pack)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        ext.bottleneck64_nhwc(x, pack)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    mb = N * H * W * (cin + 256) * 2 / 1e6
    print(f"bottleneck64 cin={cin} ds={ds}: {us:.1f} us  ({mb:.0f} MB min traffic -> {mb / us:.2f} TB/s)")

if __name__ == '__main__':
    run(256, False); run(64, True)
