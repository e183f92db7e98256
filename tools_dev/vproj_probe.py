#!/usr/bin/env python
"""Time the stacked SCA value projection at the base shape (6 cameras x 4 FPN levels = 184 950 rows, K = 256, four
layers' 256 columns each, fp16 or fp32 output), one launch; OCC_VPROJ_RESIDENT=0 in the environment selects the tiled
kernel (the switch is read once per process: run the script twice for an A/B).
usage: python tools_dev/vproj_probe.py [iters]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
cams, K, N, P = 6, 256, 256, 4
hw = [(116, 200), (58, 100), (29, 50), (15, 25)]
rpg = [h * w for h, w in hw]
starts = [0]
for r in rpg[:-1]:
    starts.append(starts[-1] + r)
total = sum(rpg)
total += total & 1            # fp16 maps are written in pixel pairs: an even block per camera
g = torch.Generator().manual_seed(0)
a_list = [torch.randn(cams * r, K, generator=g).to(dev).to(torch.bfloat16) for r in rpg]
ws = [(torch.randn(N, K, generator=g) / 16).to(dev) for _ in range(P)]
gbs = [torch.randn(len(rpg), cams, N, generator=g).to(dev) for _ in range(P)]
for dt in (torch.float16, torch.float32):
    out = torch.empty((P, cams * total, N), device=dev, dtype=dt)
    fn = lambda: ext.value_proj_bf16_planes(a_list, ws, gbs, out, rows_per_group=rpg, out_group_rows=total, out_row0=starts)
    for _ in range(5):
        fn()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    ref = (a_list[3].double() @ ws[2].double().t()).view(cams, rpg[3], N) + gbs[2][3].double()[:, None, :]
    rows = ext.sca_unpair_layout(out[2].view(cams, total, N // 32, 32)).reshape(cams, total, N) if dt == torch.float16 \
        else out[2].view(cams, total, N)
    got = rows[:, starts[3]:starts[3] + rpg[3]].double()
    flops = 2.0 * cams * total * K * N * P
    mb = (cams * total * K * 2 + out.numel() * out.element_size()) / 1e6
    print(json.dumps(dict(kernel="resident" if os.environ.get("OCC_VPROJ_RESIDENT", "1") != "0" else "tiled",
                          out=str(dt), median_ms=ms[len(ms) // 2], min_ms=ms[0], maxdiff_f64=float((got - ref).abs().max()),
                          mfma_tflops_2term=2 * flops / ms[len(ms) // 2] / 1e9, hbm_gbps=mb / ms[len(ms) // 2])), flush=True)
