"""Development: failure RATE of the default hot path under co-scheduled load, per runtime / kernel
switch.  Un-instrumented steps (only references to the ext.* outputs are kept: no extra launches); every repetition is
compared bit for bit with the solo run; for a failing repetition the first differing launch is named.
usage: python tools_dev/hazard_matrix.py <reps> [label]   (switches: environment, inherited)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext, synthetic                              # noqa: E402
from tests.util import build_pair                                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
label = sys.argv[2] if len(sys.argv) > 2 else ""
g = dict(synthetic.BASE, num_points=8, num_layers=4)
prod, _ = build_pair(g, seed=12)


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


x = [nhwc(f.to(torch.bfloat16)) for f in synthetic.make_features(g, seed=12)]
metas = synthetic.make_img_metas(g)
OPS = [o for o in ("linear_pair_chain", "tsa_fused_forward", "linear_ln_chain", "value_range_scale", "value_proj_bf16_planes",
                   "sca_fused_forward", "encoder_ffn_chain", "conv3d_bn_relu", "conv3d_heads_decode") if hasattr(ext, o)]
trace = []
real = {o: getattr(ext, o) for o in OPS}


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for e in o for t in tensors(e)]
    return []


for o in OPS:
    setattr(ext, o, (lambda name: lambda *a, **k: (lambda out: (trace.append((name, [t.detach() for t in tensors(out)])), out)[1])(
        real[name](*a, **k)))(o))


def run():
    del trace[:]
    with torch.no_grad():
        prod(x, metas)
    torch.cuda.synchronize()
    return list(trace)


run()
solo = run()
load = torch.cuda.Stream()
a = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
big = torch.empty(256 << 20, device='cuda', dtype=torch.float32)
dst = torch.empty_like(big)
with torch.cuda.stream(load):          # the load's own first-use costs are paid before the measured repetitions
    dst.copy_(big)
    c = a @ b
torch.cuda.synchronize()
fails, first_ops = 0, {}
t0 = time.time()
for rep in range(reps):
    with torch.cuda.stream(load):
        for _ in range(6):
            dst.copy_(big)
            c = a @ b
    got = run()
    bad = None
    for i, ((n1, o1), (n2, o2)) in enumerate(zip(solo, got)):
        for j, (t1, t2) in enumerate(zip(o1, o2)):
            if not torch.equal(t1, t2):
                d = (t1.float() - t2.float()).abs()
                rows = int((d.reshape(-1, d.shape[-1]).amax(-1) > 0).sum()) if d.dim() > 1 else int((d > 0).sum())
                bad = (i, n1, j, rows, float(d.max()))
                break
        if bad:
            break
    if bad:
        fails += 1
        first_ops[bad[1]] = first_ops.get(bad[1], 0) + 1
        if fails <= 4:
            print(f"   rep {rep}: first differing launch #{bad[0]} {bad[1]} output {bad[2]}: {bad[3]} rows, max {bad[4]:.2e}", flush=True)
env = {k: v for k, v in os.environ.items() if k.startswith(("OCC_", "AMD_", "GPU_", "HIP_FORCE", "HSA_EN", "ROC"))}
print(f"HAZARD {label}: {fails} of {reps} repetitions differ from the solo run; first differing launch by kernel {first_ops}; "
      f"{time.time() - t0:.1f} s; env {env}", flush=True)
