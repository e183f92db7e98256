"""Development: the DEFAULT hot path under co-scheduled load — which launch first produces a wrong
row, and what do the wrong rows hold?  Every ext.* launch of the step is wrapped: its outputs are cloned (same stream)
and compared, op by op, with the solo run's.  For the first op that differs the script prints the differing rows and tests
them against (a) zero, (b) the SAME op's correct rows one layer earlier (stale inputs / stale kernel arguments), and re-runs
the op alone on its recorded inputs.
usage: python tools_dev/concurrency_probe.py [load kinds: copy,gemm,gather,sort (default all)] [reps]
env: OCC_VPROJ_OVERLAP=0 (no side stream), HIP_FORCE_DEV_KERNARG, GPU_MAX_HW_QUEUES, ... are simply inherited."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext, synthetic                              # noqa: E402
from tests.util import build_pair                                  # noqa: E402

kinds = set((sys.argv[1] if len(sys.argv) > 1 else "copy,gemm,gather,sort").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print("load:", sorted(kinds), "env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "GPU_", "HIP_", "AMD_", "OCC_", "ROC"))},
      flush=True)
g = dict(synthetic.BASE, num_points=8, num_layers=4)
prod, _ = build_pair(g, seed=12)


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


x = [nhwc(f.to(torch.bfloat16)) for f in synthetic.make_features(g, seed=12)]
metas = synthetic.make_img_metas(g)
OPS = ["linear_pair_chain", "tsa_fused_forward", "linear_ln_chain", "value_range_scale", "value_proj_bf16_planes",
       "value_proj_bf16", "sca_fused_forward", "encoder_ffn_chain", "linear", "conv3d_bn_relu", "conv3d_heads_decode", "occ_heads",
       "point_sampling"]
OPS = [o for o in OPS if hasattr(ext, o)]
trace = []
real = {o: getattr(ext, o) for o in OPS}


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for e in o for t in tensors(e)]
    return []


def wrap(name):
    def f(*a, **k):
        out = real[name](*a, **k)
        # OCC_PROBE_CLONE=1: clone the outputs (extra kernels on the stream: perturbs the timing); default: keep references —
        # the path's outputs are fresh tensors that nothing overwrites, and a reference costs no launch
        keepref = os.environ.get("OCC_PROBE_CLONE") != "1"
        trace.append((name, [t.detach() if keepref else t.detach().clone() for t in tensors(out)], (a, k)))
        return out
    return f


for o in OPS:
    setattr(ext, o, wrap(o))


def run():
    del trace[:]
    with torch.no_grad():
        out = prod(x, metas)
    torch.cuda.synchronize()
    return {k: out[k].clone() for k in ('bev_embed', 'occ', 'flow')}, list(trace)


solo_out, solo = run()
solo_out, solo = run()
out2, tr2 = run()
print("ops per step:", [n for n, _, _ in solo])
print("solo twice bit-identical:", all(torch.equal(a, b) for (_, o1, _), (_, o2, _) in zip(solo, tr2) for a, b in zip(o1, o2)), flush=True)

load = torch.cuda.Stream()
a = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
big = torch.empty(256 << 20, device='cuda', dtype=torch.float32)
dst = torch.empty_like(big)
idx = torch.randint(0, 1 << 20, (1 << 22,), device='cuda')
tab = torch.randn(1 << 20, 64, device='cuda')
for rep in range(reps):
    with torch.cuda.stream(load):
        for _ in range(6):
            if "copy" in kinds:
                dst.copy_(big)
            if "gemm" in kinds:
                c = a @ b
            if "gather" in kinds:
                s = tab[idx[(rep % 4) << 20:((rep % 4) + 1) << 20]].sum(0)
            if "sort" in kinds:
                st = torch.sort(a[:256].float().view(-1))[0]
        if "fresh" in kinds:
            # operations this process has NOT run before (new code objects, library initialisation, first-use workspaces):
            # a different set in every repetition — call 1's one failure was in the first loaded repetition only
            xf = a[:1024, :1024].float()
            fresh = [lambda: torch.fft.rfft(xf), lambda: torch.cumsum(xf, 1), lambda: torch.topk(xf, 7, 1),
                     lambda: torch.erf(xf) + torch.lgamma(xf.abs() + 1), lambda: torch.linalg.norm(xf, dim=1),
                     lambda: torch.nn.functional.conv2d(xf[None, None], xf[None, None, :5, :5]),
                     lambda: torch.bmm(xf.view(16, 64, 1024), xf.view(16, 1024, 64)),
                     lambda: torch.unique(idx[:100000]), lambda: torch.nn.functional.softmax(xf, 1).multinomial(3),
                     lambda: torch.median(xf, 1), lambda: torch.argsort(xf, 1), lambda: xf.double() @ xf.double().t(),
                     lambda: torch.nn.functional.max_pool2d(xf[None, None], 3), lambda: torch.kthvalue(xf, 5, 1),
                     lambda: torch.nn.functional.layer_norm(xf, (1024,)), lambda: torch.cdist(xf[:256], xf[:256])]
            for fn in fresh[(2 * rep) % len(fresh):][:2]:
                try:
                    fn()
                except Exception as e:
                    print("   fresh op failed:", repr(e)[:100])
    got_out, got = run()
    rep_t = getattr(prod.transformer, "value_range_report", None)
    if rep_t is not None:
        print(f"rep {rep}: range report {[round(v, 6) for v in rep_t.tolist()[:5]]}", flush=True)
    first = None
    for i, ((n1, o1, io1), (n2, o2, io2)) in enumerate(zip(solo, got)):
        assert n1 == n2
        for j, (t1, t2) in enumerate(zip(o1, o2)):
            if not torch.equal(t1, t2):
                first = (i, n1, j)
                break
        if first:
            break
    final_bad = {k: int((got_out[k] != solo_out[k]).sum()) for k in got_out}
    print(f"rep {rep}: first differing op = {first}; final outputs differing elements {final_bad}", flush=True)
    if first is None:
        continue
    i, name, j = first
    t_ok, t_bad = solo[i][1][j], got[i][1][j]
    d = (t_ok.float() - t_bad.float()).abs()
    d2 = d.reshape(-1, d.shape[-1]) if d.dim() > 1 else d.reshape(-1, 1)
    rows = (d2.amax(-1) > 0).nonzero().flatten()
    print(f"   op #{i} {name} output {j} shape {tuple(t_ok.shape)}: {rows.numel()} rows differ, max {float(d.max()):.3e}; "
          f"rows {rows[:10].tolist()} ... {rows[-4:].tolist()}", flush=True)
    bad_rows = t_bad.reshape(-1, t_bad.shape[-1])[rows] if t_bad.dim() > 1 else t_bad.reshape(-1, 1)[rows]
    print(f"   wrong rows all zero: {bool((bad_rows == 0).all())}; finite: {bool(torch.isfinite(bad_rows.float()).all())}; "
          f"mean|wrong| {float(bad_rows.float().abs().mean()):.3e} vs mean|right| "
          f"{float(t_ok.reshape(-1, t_ok.shape[-1])[rows].float().abs().mean()):.3e}")
    # the same op one layer earlier / later: do the wrong rows hold ITS correct rows?
    same = [k for k, (n, _, _) in enumerate(solo) if n == name]
    for k in same:
        if k != i and solo[k][1][j].shape == t_ok.shape:
            other = solo[k][1][j].reshape(-1, t_ok.shape[-1])[rows]
            eq = int((other == bad_rows).all(-1).sum())
            print(f"   wrong rows equal to the correct rows of op #{k} ({name}): {eq} of {rows.numel()}")
    # the op alone, on the inputs it was given in the loaded run
    torch.cuda.synchronize()
    args, kw = got[i][2]
    alone = tensors(real[name](*args, **kw))[j]
    torch.cuda.synchronize()
    print(f"   re-run alone on the recorded inputs: equal to the solo result = {bool(torch.equal(alone, t_ok))}, "
          f"equal to the loaded result = {bool(torch.equal(alone, t_bad))}", flush=True)
