#!/usr/bin/env python
"""Time the encoder's Linear shapes (M = 40 000) on both bf16x3 kernels.  usage: python tools_dev/linear_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext   # noqa: E402

dev = torch.device("cuda", 0)
M = int(os.environ.get("LIN_M", "40000"))
SHAPES = [  # name, K1, K2, N, act, residual+ln
    ("tsa_value_proj 256->256", 256, 0, 256, None, False),
    ("tsa_query 512->192 (+pos)", 256, 256, 192, None, False),
    ("out_proj 256->256 +res+LN", 256, 0, 256, None, True),
    ("sca_query 256->768", 256, 0, 768, None, False),
    ("ffn1 256->512 relu", 256, 0, 512, 'relu', False),
    ("ffn2 512->256 +res+LN", 512, 0, 256, None, True),
]
g = torch.Generator().manual_seed(0)
for name, K1, K2, N, act, resln in SHAPES:
    a = torch.randn(M, K1, generator=g).to(dev)
    a2 = torch.randn(M, K2, generator=g).to(dev) if K2 else None
    add = torch.randn(M, K2, generator=g).to(dev) if K2 else None
    w = (torch.randn(N, K1 + K2, generator=g) * (K1 + K2) ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if resln else None
    ln = (torch.ones(N, device=dev), torch.zeros(N, device=dev), 1e-5) if resln else None
    line = f"{name:30s}"
    outs = {}
    for kern in ("x3", "x3s"):
        ext.LINEAR_KERNEL = kern
        for _ in range(5):
            o = ext.linear(a, w, b, a2=a2, a2_add=add, act=act, residual=res, ln=ln)
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = ext.linear(a, w, b, a2=a2, a2_add=add, act=act, residual=res, ln=ln); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in evs)
        outs[kern] = o
        fl = 2.0 * M * N * (K1 + K2)
        line += f"  {kern}: {ms[len(ms) // 2] * 1e3:7.1f} us ({fl / ms[len(ms) // 2] / 1e9:6.1f} TF f32-eq)"
    line += f"  maxdiff {float((outs['x3'] - outs['x3s']).abs().max()):.2e}"
    print(line, flush=True)

# fused FFN vs the two launches it replaces
w1 = (torch.randn(512, 256, generator=g) / 16).to(dev); b1 = torch.randn(512, generator=g).to(dev)
w2 = (torch.randn(256, 512, generator=g) / 22).to(dev); b2 = torch.randn(256, generator=g).to(dev)
xx = torch.randn(M, 256, generator=g).to(dev)
ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5)
ext.LINEAR_KERNEL = "x3"
def two():
    return ext.linear(ext.linear(xx, w1, b1, act='relu'), w2, b2, residual=xx, ln=ln)
def one():
    return ext.ffn_fused(xx, w1, b1, w2, b2, ln=ln)
for name, fn in (("ffn two launches", two), ("ffn fused", one)):
    for _ in range(5):
        o = fn()
    evs = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name:30s} {ms[len(ms) // 2] * 1e3:7.1f} us", flush=True)
print("maxdiff fused vs two:", float((one() - two()).abs().max()))
