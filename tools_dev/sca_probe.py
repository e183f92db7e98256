#!/usr/bin/env python
"""Time the SCA gather kernel variants on the REAL inputs of the base config (captured from one forward of the
bench model's first layer) and check them against each other.  usage: python tools_dev/sca_probe.py [iters]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                      # noqa: E402
from occnet_amd import ext        # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
variants = [int(v) for v in os.environ.get("SCA_PROBE_VARIANTS", "0,1,2,3").split(",")]
dev = torch.device("cuda", 0)
cfg, model, geo = bench.build(os.path.join(bench.ROOT, "configs", "occ_base_200x200x16.py"), dev)
st = bench.Stepper(model, geo, "hotpath", "bf16", dev, seed=0, plan="folded", hot_feat_format="backbone")
captured = []
orig = ext.sca_fused_forward
ext.sca_fused_forward = lambda *a, **k: (captured.append((a, k)), orig(*a, **k))[1]
st()
torch.cuda.synchronize()
ext.sca_fused_forward = orig
a, k = captured[-1]              # last layer: realistic, query-dependent offsets
k = {kk: v for kk, v in k.items() if kk not in ("kernel", "stats", "order")}
enc = model.pts_bbox_head.transformer.encoder
head = model.pts_bbox_head
orders = {False: enc._bev_order(head.bev_h, head.bev_w, dev), True: enc._bev_order(head.bev_h, head.bev_w, dev, flat=True)}
ref = None
res = {}
for var in variants:
    stats = torch.zeros(2, dtype=torch.int64, device=dev)
    k['order'] = orders[var != 0]
    out = orig(*a, **k, kernel=var, stats=stats)
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    for _ in range(5):
        orig(*a, **k, kernel=var)
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(*a, **k, kernel=var); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    res[var] = dict(name=ext.sca_variant_name(var), median_ms=ms[len(ms) // 2], min_ms=ms[0], max_ms=ms[-1],
                    maxdiff_vs_first=float((out - ref).abs().max()), rows=int(stats[0]), n_in=int(stats[1]))
    print(var, json.dumps(res[var]), flush=True)
