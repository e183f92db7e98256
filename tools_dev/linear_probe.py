#!/usr/bin/env python
"""Time the encoder's Linear shapes (M = 40 000) on the default bf16x3 kernel.  (Round 3 compared it with the
weight-stationary persistent kernel and a one-launch FFN, both now in tools_dev/lab: profiles/r03_linear_probe_ws_v*.txt.)
usage: python tools_dev/linear_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext   # noqa: E402

dev = torch.device("cuda", 0)
M = int(os.environ.get("LIN_M", "40000"))
SHAPES = [  # name, K1, K2, N, act, residual, ln
    ("tsa_value_proj 256->256", 256, 0, 256, None, False, False),
    ("tsa_query 256->192 +res", 256, 0, 192, None, True, False),
    ("out_proj 256->256 +res+LN", 256, 0, 256, None, True, True),
    ("sca_query 256->768", 256, 0, 768, None, False, False),
    ("ffn1 256->512 relu", 256, 0, 512, 'relu', False, False),
    ("ffn2 512->256 +res+LN", 512, 0, 256, None, True, True),
]


def timed(fn, n=30):
    for _ in range(5):
        o = fn()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2], o


g = torch.Generator().manual_seed(0)
for name, K1, K2, N, act, has_res, has_ln in SHAPES:
    a = torch.randn(M, K1, generator=g).to(dev)
    w = (torch.randn(N, K1 + K2, generator=g) * (K1 + K2) ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if has_res else None
    ln = (torch.ones(N, device=dev), torch.zeros(N, device=dev), 1e-5) if has_ln else None
    line = f"{name:30s}"
    outs = {}
    mb = (M * (K1 + K2) + M * N + (M * N if has_res else 0)) * 4 / 1e6
    ms, o = timed(lambda: ext.linear(a, w, b, act=act, residual=res, ln=ln))
    line += f"  {ms * 1e3:7.1f} us ({mb / ms / 1e3:5.2f} TB/s)  [{mb:.0f} MB, floor {mb / 6.3e3 * 1e3:.1f} us @6.3 TB/s]"
    print(line, flush=True)

