#!/usr/bin/env python
"""Time the SCA gather kernels on the REAL inputs of the base config (captured from one forward of the bench model's
last layer) and check them against each other: the fp32-value kernel (sca_fused_kernel) against the fp16-value kernel
(sca_fused_h_kernel, 4 waves/SIMD + 2-sample rolling window; the other (waves, window) pairs were measured in round 3
through a since-removed development switch: profiles/r03_sca_probe_fp16_rows.txt), plus the value projection with
fp32 / fp16 output.   usage: python tools_dev/sca_probe.py [iters]"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                      # noqa: E402
from occnet_amd import ext        # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
cfg, model, geo = bench.build(os.path.join(bench.ROOT, "configs", "occ_base_200x200x16.py"), dev)
st = bench.Stepper(model, geo, "hotpath", "bf16", dev, seed=0, plan="folded", hot_feat_format="backbone")
captured = []
orig = ext.sca_fused_forward
ext.sca_fused_forward = lambda *a, **k: (captured.append((a, k)), orig(*a, **k))[1]
ext.SCA_VALUES = "f32"
st()
torch.cuda.synchronize()
ext.sca_fused_forward = orig
a, k = captured[-1]              # last layer: realistic, query-dependent offsets
k = {kk: v for kk, v in k.items() if kk not in ("kernel", "stats")}
v32 = a[0]
v16 = v32.half()
print(f"value {tuple(v32.shape)}; fp16 rounding of the values: max abs {float((v16.float() - v32).abs().max()):.3e}", flush=True)


def timed(fn):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2], ms[0]


ref = orig(v32, *a[1:], **k)
ref16 = None
v16 = ext.sca_pair_layout(v16)          # the kernel's operand order (what the value projection writes)
for name, val in (("f32 values (sca_fused_kernel)", v32), ("f16 values (sca_fused_h_kernel, 4 waves, window 2)", v16)):
    stats = torch.zeros(2, dtype=torch.int64, device=dev)
    kk = dict(k, value_layout="pairs" if val is v16 else "rows")
    out = orig(val, *a[1:], **kk, stats=stats)
    torch.cuda.synchronize()
    med, mn = timed(lambda: orig(val, *a[1:], **kk))
    if val is v16 and ref16 is None:
        ref16 = out
    print(json.dumps(dict(kernel=name, median_ms=med, min_ms=mn, maxdiff_vs_f32_kernel=float((out - ref).abs().max()),
                          maxdiff_vs_first_f16=None if ref16 is None else float((out - ref16).abs().max()),
                          rows=int(stats[0]), n_in=int(stats[1]))), flush=True)

# the value projection that feeds it: fp32 vs fp16 output (LazyFeatures.project of the last layer's value_proj)
tr = model.pts_bbox_head.transformer
from occnet_amd.plugin.transformer_occ import LazyFeatures   # noqa: E402
lf = LazyFeatures(tr, st.feats)
vp = tr.encoder.layers[-1].attentions[1].deformable_attention.value_proj
for mode in ("f32", "f16"):
    ext.SCA_VALUES = mode
    med, mn = timed(lambda: lf.project(vp))
    print(json.dumps(dict(kernel=f"value_proj_bf16 -> {mode}", median_ms=med, min_ms=mn)), flush=True)
ext.SCA_VALUES = "f32"

# Where do the gather's operands live when it runs?  In the step each layer's value plane was written ~1 GB of traffic
# earlier (cold), the query Linear outputs right before (warm).  SCA_COLD=1: time single launches after (a) nothing (hot:
# the loop above), (b) a 1 GB flush (everything cold), (c) the flush, then a streaming READ of the value plane (value
# warm in the memory-side cache, the rest cold), (d) the flush, then reads of value AND the query Linear outputs.
if os.environ.get("SCA_COLD", "0") == "1":
    kk = dict(k, value_layout="pairs")
    junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)          # 1 GB
    offs_t, logit_t = a[3], a[4]            # sca_fused_forward(value, shapes, start, offsets, logits, ...)
    def once(prep):
        ts = []
        for _ in range(12):
            junk.fill_(1.0)
            prep()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); orig(v16, *a[1:], **kk); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]
    t_stream, _ = timed(lambda: v16.sum())
    print(json.dumps(dict(experiment="sca gather, operand residency",
                          hot_ms=timed(lambda: orig(v16, *a[1:], **kk))[0],
                          all_cold_ms=once(lambda: None),
                          value_streamed_ms=once(lambda: v16.sum()),
                          value_and_linear_streamed_ms=once(lambda: (v16.sum(), offs_t.sum(), logit_t.sum())),
                          linear_streamed_only_ms=once(lambda: (offs_t.sum(), logit_t.sum())),
                          stream_read_of_value_ms=t_stream)), flush=True)
