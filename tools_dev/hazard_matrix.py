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
# (value_proj_bf16_planes is not compared: its buffer carries one never-written padding pixel per camera — the gathers that
# read the planes are compared instead)
OPS = [o for o in ("linear_pair_chain", "tsa_fused_forward", "linear_ln_chain", "value_range_scale",
                   "sca_fused_forward", "encoder_ffn_chain", "conv3d_bn_relu", "conv3d_heads_decode") if hasattr(ext, o)]
trace = []
real = {o: getattr(ext, o) for o in OPS}


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for e in o for t in tensors(e)]
    return []


# HZ_SLEEP_BEFORE=<op>[,<op>]: spin the GPU for HZ_SLEEP_CYCLES (default 100 000 ~ 45 us) on the step's stream in front of
# the named launches — does a pause between a producer kernel and its consumer change the failure rate?
_sleep_ops = set(filter(None, os.environ.get("HZ_SLEEP_BEFORE", "").split(",")))
_sleep_cycles = int(os.environ.get("HZ_SLEEP_CYCLES", "100000"))
for o in _sleep_ops:
    real[o] = (lambda fn: lambda *a, **k: (torch.cuda._sleep(_sleep_cycles), fn(*a, **k))[1])(real[o])
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
kinds = set(os.environ.get("HZ_LOAD", "copy,gemm").split(","))
# load kinds "vproj" / "range": the library's OWN side-stream kernels as an INDEPENDENT load (own outputs, no event or
# stream dependency on the step) — separates "these kernels next to the step" from "the side stream's event plumbing"
map_rows = [f.permute(0, 1, 3, 4, 2).reshape(-1, f.shape[2]) for f in x]
hw = [f.shape[3] * f.shape[4] for f in x]
starts = [sum(hw[:i]) for i in range(len(hw))]
total = sum(hw) + (sum(hw) & 1)
gl = torch.Generator().manual_seed(3)
ws = [((torch.rand(256, 256, generator=gl) * 2 - 1) * 0.1).cuda() for _ in range(4)]
gbs = [torch.randn(4, 6, 256, generator=gl).cuda() for _ in range(4)]
planes = torch.empty(4, 6 * total, 256, dtype=torch.float32 if os.environ.get("HZ_PLANES") == "f32" else torch.float16, device='cuda')


ca_attn = torch.randn(1, 40000, 256, device='cuda')
ca_q = torch.randn(1, 40000, 256, device='cuda')
ca_w1 = ((torch.rand(256, 256, generator=gl) * 2 - 1) * 0.06).cuda()
ca_b1 = torch.zeros(256, device='cuda')
ca_ln = torch.nn.LayerNorm(256).cuda()
ca_w2 = ((torch.rand(768, 256, generator=gl) * 2 - 1) * 0.06).cuda()
ca_b2 = torch.zeros(768, device='cuda')


_micro = None
_sink = torch.zeros(4, device='cuda')


def spin(kind, iters):
    """synthetic aggressors of tools_dev/hazard_micro.hip (spinN in HZ_LOAD): 0 = MFMAs only, 1 = MFMAs + LDS reads,
    2 = LDS reads only, 3 = packed-fp32 VALU only"""
    global _micro
    if _micro is None:
        import ctypes
        _micro = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libhazard_micro.so"))
    import ctypes
    _micro.hz_spin(ctypes.c_void_p(_sink.data_ptr()), kind, 512, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def own_kernels():
    for kd in range(4):
        if f"spin{kd}" in kinds:
            spin(kd, int(os.environ.get("HZ_SPIN_ITERS", "20000")))
    if "chainA" in kinds:          # chain program A (output_proj + LN + the SCA's query Linears) on own buffers
        real["linear_ln_chain"](ca_attn, ca_q, ca_w1, ca_b1, ca_ln, ca_w2, ca_b2)
    if "range" in kinds:
        real["value_range_scale"](map_rows, [20.0] * 4, [5.0] * 4)
    if "vproj" in kinds:
        ext.value_proj_bf16_planes(map_rows, ws, gbs, planes, rows_per_group=hw, out_group_rows=total, out_row0=starts)


with torch.cuda.stream(load):          # the load's own first-use costs are paid before the measured repetitions
    dst.copy_(big)
    c = a @ b
    own_kernels()
torch.cuda.synchronize()
fails, first_ops = 0, {}
t0 = time.time()
for rep in range(reps):
    with torch.cuda.stream(load):
        for _ in range(6):
            if "copy" in kinds:
                dst.copy_(big)
            if "gemm" in kinds:
                c = a @ b
            own_kernels()
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
        if fails <= 2:
            # forensics on the wrong rows: where are they, what do they hold?
            t_ok, t_bad = solo[bad[0]][1][bad[2]], got[bad[0]][1][bad[2]]
            ok2, bad2 = t_ok.reshape(-1, t_ok.shape[-1]), t_bad.reshape(-1, t_bad.shape[-1])
            idx = ((ok2 != bad2).any(-1)).nonzero().flatten()
            print(f"      rows {idx.tolist()[:24]}; (y, x) at 200 wide {[(int(i) // 200, int(i) % 200) for i in idx[:12]]}")
            wrong = bad2[idx]
            print(f"      all zero {bool((wrong == 0).all())}; finite {bool(torch.isfinite(wrong.float()).all())}; elements that differ "
                  f"per row {[(int((ok2[i] != bad2[i]).sum())) for i in idx[:12]]} of {ok2.shape[1]}")
            # equal to ANOTHER correct row of the same tensor, or to the same row of the same kernel's other launches?
            for r_i in idx[:6]:
                m = (ok2 == bad2[r_i]).all(-1).nonzero().flatten().tolist()
                others = [k for k, (n, o) in enumerate(solo) if n == bad[1] and k != bad[0] and o[bad[2]].shape == t_ok.shape
                          and bool((o[bad[2]].reshape(-1, t_ok.shape[-1])[r_i] == bad2[r_i]).all())]
                print(f"      wrong row {int(r_i)}: equals correct row(s) {m[:4]} of this launch; equals the same row of launch(es) {others}")
env = {k: v for k, v in os.environ.items() if k.startswith(("OCC_", "AMD_", "GPU_", "HIP_FORCE", "HSA_EN", "ROC"))}
print(f"HAZARD {label}: {fails} of {reps} repetitions differ from the solo run; first differing launch by kernel {first_ops}; "
      f"{time.time() - t0:.1f} s; env {env}", flush=True)
