#!/usr/bin/env python
"""Time the encoder's Linear shapes (M = 40 000) on the bf16x3 kernels; OCC_LINEAR_RESIDENT=0 in the environment keeps the
tiled kernel (the switch is read once per process: run the script twice for an A/B).
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
    ("tsa_query 256+256->192 (history)", 256, 256, 192, None, False, False),
    ("out_proj 256->256 +res+LN", 256, 0, 256, None, True, True),
    ("sca_query 256->768", 256, 0, 768, None, False, False),
    ("ffn1 256->512 relu", 256, 0, 512, 'relu', False, False),
    ("ffn2 512->256 +res+LN", 512, 0, 256, None, True, True),
]


def timed(fn, n=40):
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


kern = "tiled" if os.environ.get("OCC_LINEAR_RESIDENT", "1") == "0" else "resident"
g = torch.Generator().manual_seed(0)
total = 0.0
for name, K1, K2, N, act, has_res, has_ln in SHAPES:
    a = torch.randn(M, K1, generator=g).to(dev)
    a2 = torch.randn(M, K2, generator=g).to(dev) if K2 else None
    w = (torch.randn(N, K1 + K2, generator=g) * (K1 + K2) ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if has_res else None
    ln = (torch.ones(N, device=dev), torch.zeros(N, device=dev), 1e-5) if has_ln else None
    mb = (M * (K1 + K2) + M * N + (M * N if has_res else 0)) * 4 / 1e6
    ms, o = timed(lambda: ext.linear(a, w, b, a2=a2, act=act, residual=res, ln=ln))
    ref = torch.nn.functional.linear((a if a2 is None else torch.cat([a, a2], 1))[:512].double(), w.double(), b.double())
    if act:
        ref = ref.relu()
    if has_res:
        ref = ref + res[:512].double()
    if has_ln:
        ref = torch.nn.functional.layer_norm(ref, (N,), ln[0].double(), ln[1].double(), 1e-5)
    total += ms
    print(f"{kern:8s} {name:34s} {ms * 1e3:7.1f} us ({mb / ms / 1e3:5.2f} TB/s)  maxdiff {float((o[:512].double() - ref).abs().max()):.2e}"
          f"  [{mb:.0f} MB, floor {mb / 6.3e3 * 1e3:.1f} us @6.3 TB/s]", flush=True)
print(f"{kern:8s} sum {total * 1e3:.1f} us", flush=True)
