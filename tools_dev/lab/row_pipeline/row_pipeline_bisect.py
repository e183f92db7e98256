"""Development: WHICH launch of the row pipeline first produces a wrong row under OCC_ROW_PIPELINE_SERIAL=1?
Every stage of every band keeps its outputs (OCC_ROW_PIPELINE_DEBUG_KEEP=1): T -> attn, A -> x1 / lin, S -> slots,
B -> out rows / next layer's zq / zv rows.  The bands on ONE stream are the reference (same kernels, same launches, no
concurrency); the first tensor of the serial multi-stream run that differs from it names the stage.  That stage is then
re-run ALONE on the serial run's own recorded inputs: equal to the reference -> its inputs were fine and the launch computed
something else while other kernels ran; equal to the bad output -> its inputs were bad when it ran.
usage: python tools_dev/row_pipeline_bisect.py [K] [reps]"""
import os
import sys

import torch

os.environ["OCC_ROW_PIPELINE_DEBUG_KEEP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import synthetic                                   # noqa: E402
from occnet_amd.plugin import encoder as enc_mod                   # noqa: E402
from tests.util import build_pair                                  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = dict(synthetic.BASE, num_points=8, num_layers=4)
prod, _ = build_pair(g, seed=12)
feats = [f.to(torch.bfloat16) for f in synthetic.make_features(g, seed=12)]


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


metas = synthetic.make_img_metas(g)
x = [nhwc(f) for f in feats]
encoder = prod.transformer.encoder
print("env:", {k_: v for k_, v in os.environ.items() if k_.startswith(("HSA_", "GPU_", "HIP_", "AMD_", "OCC_"))}, flush=True)


def snapshot():
    """{(layer, band, name): tensor clone} of the last pipelined call."""
    d = encoder._row_debug
    out = {}
    for l, st in enumerate(d['keep']):
        for i, s in enumerate(st):
            b = d['bands'][i]
            for name in ('attn', 'x1', 'lin', 'slots'):
                out[(l, i, name)] = s[name].clone()
            out[(l, i, 'out')] = d['outs'][l][0, b['m0']:b['m1']].clone()
            if l < len(d['tails']):
                out[(l, i, 'nzq')] = d['tails'][l][1][0, b['m0']:b['m1']].clone()
                out[(l, i, 'nzv')] = d['tails'][l][2][0, b['m0']:b['m1']].clone()
    return out


ORDER = ['attn', 'x1', 'lin', 'slots', 'out', 'nzq', 'nzv']
STAGE = dict(attn='T (TSA gather)', x1='A (chain program A)', lin='A (chain program A)', slots='S (SCA gather)',
             out='B (chain program B)', nzq='B (chain program B: tail)', nzv='B (chain program B: tail)')


def run(mode):
    os.environ["OCC_ROW_PIPELINE_STREAMS"] = "0" if mode == "one stream" else "1"
    os.environ["OCC_ROW_PIPELINE_DEBUG_SYNC"] = "0"
    enc_mod._ROW_PIPELINE_SERIAL = "serial" in mode
    os.environ["OCC_ROW_PIPELINE_DUMMY"] = "1" if "dummy" in mode else "0"
    encoder._row_plan = None
    enc_mod._ROW_PIPELINE = k
    encoder._row_debug = None
    prod(x, metas)
    torch.cuda.synchronize()
    if encoder._row_debug is None:          # the first call after a change takes the standard path
        prod(x, metas)
        torch.cuda.synchronize()
    snap = snapshot()
    enc_mod._ROW_PIPELINE = 0
    return snap


with torch.no_grad():
    prod(x, metas)
    torch.cuda.synchronize()
    ref = run("one stream")
    again = run("one stream")
    same = all(torch.equal(ref[key], again[key]) for key in ref)
    print(f"one stream twice: bit-identical = {same}", flush=True)
    for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("default", "serial")):
        for rep in range(reps):
            got = run(mode)
            first = None
            report = []
            for l in range(4):
                for name in ORDER:
                    for i in range(k):
                        key = (l, i, name)
                        if key not in ref:
                            continue
                        d = (got[key] - ref[key]).abs()
                        bad_rows = int((d.reshape(d.shape[-2] if d.dim() > 1 else -1, -1).amax(-1) > 1e-3).sum()) if d.numel() else 0
                        if bad_rows and first is None:
                            first = key
                        if bad_rows:
                            report.append(f"L{l} band{i} {name}: {bad_rows} rows, max {float(d.max()):.2e}")
            print(f"K={k} mode={mode} rep {rep}: first bad tensor = {first}"
                  + (f" -> stage {STAGE[first[2]]}" if first else "") + " | " + "; ".join(report[:6]), flush=True)
            if first is not None:
                l, i, name = first
                d = (got[first] - ref[first]).abs()
                rows = (d.reshape(d.shape[-2], -1).amax(-1) > 1e-3).nonzero().flatten()
                print(f"   bad rows (band-local) {rows[:12].tolist()} ... {rows[-4:].tolist()}; "
                      f"row // 64 tiles {sorted(set((rows // 64).tolist()))[:16]}", flush=True)
                # inputs of that stage in the serial run: were THEY equal to the reference's?
                deps = dict(attn=[(l - 1, None, 'nzq'), (l - 1, None, 'nzv')], x1=[(l, i, 'attn'), (l - 1, i, 'out')],
                            lin=[(l, i, 'attn'), (l - 1, i, 'out')], slots=[(l, i, 'lin')],
                            out=[(l, i, 'slots'), (l, i, 'x1')], nzq=[(l, i, 'slots'), (l, i, 'x1')],
                            nzv=[(l, i, 'slots'), (l, i, 'x1')])[name]
                for (dl, di, dn) in deps:
                    for bi in (range(k) if di is None else [di]):
                        kk = (dl, bi, dn)
                        if kk in ref:
                            dd = float((got[kk] - ref[kk]).abs().max())
                            print(f"   input {kk}: final contents differ from the reference by {dd:.2e}", flush=True)
