"""Development: where does the row pipeline (OCC_ENCODER_ROW_PIPELINE) leave the standard chain path?  Full base
geometry, 4 layers, bf16 NHWC maps; per-layer max|pipeline - standard| with the bands on ONE stream (plumbing /
kernel contracts) and on their own streams (ordering)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import synthetic                                   # noqa: E402
from occnet_amd.plugin import encoder as enc_mod                   # noqa: E402
from tests.util import build_pair, maxdiff                         # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = dict(synthetic.BASE, num_points=8, num_layers=4)
prod, _ = build_pair(g, seed=12)
feats = [f.to(torch.bfloat16) for f in synthetic.make_features(g, seed=12)]


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


metas = synthetic.make_img_metas(g)
x = [nhwc(f) for f in feats]
encoder = prod.transformer.encoder
std = []
orig_fc = enc_mod.BEVFormerLayer.forward_chain


def rec_fc(self, *a, **kw):
    out = orig_fc(self, *a, **kw)
    std.append(out[0])
    return out


enc_mod.BEVFormerLayer.forward_chain = rec_fc
piped = []
orig_rp = encoder._forward_row_pipeline


def rec_rp(*a, **kw):
    r = orig_rp(*a, **kw)
    if r is not None:
        piped.append(r)
    return r


encoder._forward_row_pipeline = rec_rp
with torch.no_grad():
    prod(x, metas)
    torch.cuda.synchronize()
    want = list(std)
    # (profiles/r04_rowpipe_debug3.log was written by an earlier cut of this list: its "plain" = "serial" here, its
    # "noserial" = "default")
    for one in ("default", "one stream", "serial", "serial+sync", "serial+xbarrier"):
        os.environ["OCC_ROW_PIPELINE_STREAMS"] = "0" if one == "one stream" else "1"
        os.environ["OCC_ROW_PIPELINE_DEBUG_SYNC"] = "1" if "sync" in one else "2" if "xbarrier" in one else "0"
        enc_mod._ROW_PIPELINE_SERIAL = "serial" in one
        encoder._row_plan = None
        enc_mod._ROW_PIPELINE = k
        for rep in range(3):
            del piped[:]
            prod(x, metas)
            torch.cuda.synchronize()
            if piped:
                for li, (a, b) in enumerate(zip(piped[0], want)):
                    d = (a - b).abs().amax(-1)[0].view(200, 200)
                    bad = (d > 1e-3)
                    ys = bad.any(1).nonzero().flatten().tolist()
                    if li == 1:
                        print(f"   layer {li}: {int(bad.sum())} bad queries, BEV rows {ys[:4]}..{ys[-4:]} ({len(ys)} rows), "
                              f"band 0: {int(bad[:104].sum())}, band 1: {int(bad[104:].sum())}")
                print(f"K={k} mode={one} rep {rep}: per-layer max|diff| =",
                      " ".join(f"{maxdiff(a, b):.2e}" for a, b in zip(piped[0], want)),
                      "| bands", [f"{maxdiff(piped[0][0][:, m0:m1], want[0][:, m0:m1]):.1e}"
                                  for m0, m1, _ in enc_mod.row_bands(200, 200, k)], flush=True)
        enc_mod._ROW_PIPELINE = 0
