#!/usr/bin/env python
"""Localise the full-size history-chain mismatch (tests/test_gpu_fullsize.py::test_base_geometry_with_history)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import synthetic
from tests.util import build_pair, maxdiff

torch.set_num_threads(32)
nl = int(os.environ.get("NL", "2"))
g = dict(synthetic.BASE, num_points=8, num_layers=nl)
prod, ora = build_pair(g, seed=17)
frames = [synthetic.make_features(g, seed=170 + i) for i in range(4)]
metas = []
for i in range(4):
    m = synthetic.make_img_metas(g, seed=i)[0]
    m['prev_bev_exists'] = i != 1
    m['can_bus'][-1] = 2.5 * i - 3.0
    metas.append(m)


def taps(head, store):
    hs = []
    for li, layer in enumerate(head.transformer.encoder.layers):
        hs.append(layer.attentions[0].register_forward_hook(lambda m, a, o, li=li: store.__setitem__(f'L{li}.tsa', o.detach().clone())))
        hs.append(layer.attentions[1].register_forward_hook(lambda m, a, o, li=li: store.__setitem__(f'L{li}.sca', o.detach().clone())))
        hs.append(layer.register_forward_hook(lambda m, a, o, li=li: store.__setitem__(f'L{li}.out', o.detach().clone())))
    return hs


with torch.no_grad():
    pp = po = None
    for i in range(3):
        if not metas[i]['prev_bev_exists']:
            pp = po = None
        pp = prod([f.cuda() for f in frames[i]], [metas[i]], pp, only_bev=True)
        po = ora(frames[i], [metas[i]], po, only_bev=True)
        print(f"chain frame {i}: bev diff {maxdiff(pp, po):.3e}", flush=True)
    for name, kw in (("final only_bev", dict(only_bev=True)), ("final full", dict())):
        sp, so = {}, {}
        for layer in prod.transformer.encoder.layers:
            layer.use_fused = os.environ.get("UNFUSED", "0") != "1"
        hp, ho = taps(prod, sp), taps(ora, so)
        op = prod([f.cuda() for f in frames[3]], [metas[3]], pp.clone(), **kw)
        oo = ora(frames[3], [metas[3]], po.clone(), **kw)
        for h in hp + ho:
            h.remove()
        bp = op if kw else op['bev_embed']
        bo = oo if kw else oo['bev_embed']
        print(f"{name}: bev diff {maxdiff(bp, bo):.3e}; taps product {sorted(sp)} oracle {sorted(so)}")
        for k in sorted(so):
            if k in sp:
                d = (sp[k].cpu() - so[k]).abs()
                print(f"   {k}: max diff {float(d.max()):.3e}; rows differing > 1e-3: {int((d.amax(-1) > 1e-3).sum())} of {d.shape[-2]}", flush=True)
    # same final call with the history passed at angle 0 / angle 2
    for ang in (0.0, 2.0, 4.5):
        m = dict(metas[3]); m['can_bus'] = m['can_bus'].copy(); m['can_bus'][-1] = ang
        bp = prod([f.cuda() for f in frames[3]], [m], pp.clone(), only_bev=True)
        bo = ora(frames[3], [m], po.clone(), only_bev=True)
        d = (bp.cpu() - bo).abs()
        print(f"angle {ang}: bev diff {float(d.max()):.3e}, queries differing > 1e-3: {int((d.amax(-1) > 1e-3).sum())}", flush=True)
    # rotation alone
    from occnet_amd.plugin.transformer_occ import rotate_bev_nearest
    import oracle.model as om
    x = po[0].reshape(200, 200, 256).permute(2, 0, 1).contiguous()
    for ang in (2.0, 4.5):
        a = rotate_bev_nearest(x.cuda(), ang, [100, 100]).cpu()
        b = om.rotate_nearest(x.clone(), ang, [100, 100])
        print(f"rotation {ang}: pixels differing {int((a != b).any(0).sum())}")
