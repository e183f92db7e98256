"""Development (round 5): the REAL TSA gather kernel alone (not the step) next to the library's value projection on another
stream, on CRAFTED inputs that say what went wrong in a wrong row:
  case "random": random offsets / logits / values — compared bit for bit with the solo run;
  case "ones":   value map = 1.0 everywhere, random offsets / logits: every interior output must be 1 (softmax and bilinear
                 weights sum to one) — a deviation means wrong WEIGHTS;
  case "centre": logits = 0, offsets = 0: every sample sits on the query's own pixel, output = the query's own value row — a
                 deviation means a wrong row FETCH (address or data).
With the library as it is now (the gathers compute their quotients with occ::fdiv) every case reports 0; the commit before
"Gather kernels without the IEEE division expansion" reproduces the hazard (99 of 100 runs), and OCC_TSA_VARIANT there selects
the compile-time variants of profiles/r05_c17_tsa_standalone_variants.log.  tools_dev/hazard_repro.py is the self-contained
successor (a copy of that kernel in hazard_micro.hip, no torch): it fails with fdiv as well and is cured by a set-up without
scalar lane masks.
usage: python tools_dev/hazard_tsa_standalone.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from occnet_amd import ext, synthetic                              # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(7)
bh = bw = 200
Nq, M, D, P = bh * bw, 8, 32, 4
ys, xs = torch.meshgrid(torch.arange(bh), torch.arange(bw), indexing='ij')
ref = torch.stack([(xs.flatten() + 0.5) / bw, (ys.flatten() + 0.5) / bh], -1)[None, :, None, :].expand(2, Nq, 1, 2).contiguous().cuda()
value_r = torch.randn(1, Nq, M, D, generator=g).cuda()
offs_r = (torch.randn(1, Nq, M * 2 * P * 2, generator=g) * 1.5).cuda()
logits_r = torch.randn(1, Nq, M * 2 * P, generator=g).cuda()
cases = {
    "random": (value_r, offs_r, logits_r),
    "ones": (torch.ones_like(value_r), offs_r, logits_r),
    "centre": (value_r, torch.zeros_like(offs_r), torch.zeros_like(logits_r)),
}
feats = synthetic.make_features(dict(synthetic.BASE), seed=12)
maps = [f.reshape(-1, 256, f.shape[3], f.shape[4]).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for f in feats]
map_rows = [m.permute(0, 2, 3, 1).reshape(-1, 256) for m in maps]
hw = [m.shape[2] * m.shape[3] for m in maps]
starts = [sum(hw[:i]) for i in range(len(hw))]
total = sum(hw) + (sum(hw) & 1)
ws = [((torch.rand(256, 256, generator=g) * 2 - 1) * 0.1).cuda() for _ in range(4)]
gbs = [torch.randn(4, 6, 256, generator=g).cuda() for _ in range(4)]
planes = torch.empty(4, 6 * total, 256, dtype=torch.float16, device='cuda')
load = torch.cuda.Stream()


def vproj():
    ext.value_proj_bf16_planes(map_rows, ws, gbs, planes, rows_per_group=hw, out_group_rows=total, out_row0=starts)


with torch.cuda.stream(load):
    vproj()
torch.cuda.synchronize()
interior = ((ys >= 8) & (ys < bh - 8) & (xs >= 8) & (xs < bw - 8)).flatten().cuda()
for name, (value, offs, logits) in cases.items():
    run = lambda: ext.tsa_fused_forward(value, offs, logits, ref, bh, bw, M, P, shared_queue=True)
    solo = run()
    torch.cuda.synchronize()
    assert torch.equal(run(), solo)
    bad_reps, shown = 0, 0
    for rep in range(reps):
        with torch.cuda.stream(load):
            for _ in range(3):
                vproj()
        outs = [run() for _ in range(4)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, solo):
                bad_reps += 1
                if shown < 3:
                    shown += 1
                    d = (o - solo).abs()[0]
                    rows = (d.amax(-1) > 0).nonzero().flatten()
                    r = int(rows[0])
                    cols = (d[r] > 0).nonzero().flatten()
                    print(f"   {name} rep {rep}: rows {rows.tolist()[:8]}; row {r}: {cols.numel()} wrong channels "
                          f"[{int(cols[0])}..{int(cols[-1])}]; right {solo[0, r, cols[:4]].tolist()} wrong {o[0, r, cols[:4]].tolist()}")
                    if name == "centre":
                        # does the wrong row equal ANOTHER pixel's value row (a wrong address)?
                        h0 = int(cols[0]) // 32
                        v = value[0, :, h0, :]                      # (Nq, 32) of that head
                        m = (v == o[0, r, h0 * 32:(h0 + 1) * 32]).all(-1).nonzero().flatten().tolist()
                        print(f"      head {h0}: the wrong 32 channels equal the value row of pixel(s) {m[:4]} (query {r} = pixel {r})")
                break
    extra = ""
    if name == "ones":
        extra = f"; solo interior max|out - 1| = {float((solo[0][interior] - 1).abs().max()):.2e}"
    print(f"TSA-STANDALONE case {name:7s}: {bad_reps} of {reps} repetitions (4 launches each) differ from the solo run{extra}", flush=True)
