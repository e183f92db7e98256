"""Host-side bookkeeping of the encoder's row pipeline (OCC_ENCODER_ROW_PIPELINE, plugin/encoder.py) WITHOUT a GPU.

The pipeline launches the same kernels as BEVFormerLayer.forward_chain, band by band on several streams; what can go
wrong on the host is the plumbing — band boundaries, band-local query orders, which rows of which buffer a launch
reads / writes, how the layers hand their results on.  Here every `ext` entry point the pipeline calls is replaced by a
plain torch function with the SAME indexing contract as the kernel it stands for (csrc/tsa_fused.hip, sca_fused.hip,
linear_chain_x3.hip: a gather launch indexes offs / logits / ref / vis / out by band-local query and the value map by
pixel; the chain kernels are row-local), and the streams / events by inert stand-ins.  The banded result must equal the
un-banded walk through the same stand-ins, for several band counts.  The arithmetic of the kernels is NOT under test
here (tests/test_gpu_*.py), and the opt-in GPU test of the real pipeline is tests/test_gpu_row_pipeline.py."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from occnet_amd import ext                                              # noqa: E402
from occnet_amd.plugin import Config, build_model, encoder as enc_mod   # noqa: E402
from occnet_amd.plugin.encoder import row_bands                         # noqa: E402
from occnet_amd.synthetic import bev_tile_order                         # noqa: E402


@pytest.mark.parametrize("h,w,k", [(200, 200, 2), (200, 200, 3), (50, 50, 2), (400, 400, 4), (6, 5, 3), (16, 8, 2)])
def test_row_bands_cover_the_grid_on_tile_boundaries(h, w, k):
    bands = row_bands(h, w, k)
    assert 1 <= len(bands) <= k
    assert bands[0][0] == 0 and bands[-1][1] == h * w
    for (a0, a1, ah), (b0, _, _) in zip(bands, bands[1:]):
        assert a1 == b0 and (a1 // w) % 8 == 0            # contiguous, cut on a tile row
    for m0, m1, bh in bands:
        assert m1 - m0 == bh * w and bh > 0
        order = bev_tile_order(bh, w)                      # the band-local order the gathers walk
        assert sorted(order.tolist()) == list(range(m1 - m0))


class _Stream:
    def wait_event(self, e):
        assert e is None or e.recorded, "waiting for an event that was never recorded"

    def wait_stream(self, s):
        pass


class _Event:
    def __init__(self, *a, **k):
        self.recorded = False

    def record(self, s=None):
        self.recorded = True


def _fake_ops(monkeypatch, bev_h, bev_w, log):
    """torch stand-ins with the kernels' indexing contracts."""
    nq = bev_h * bev_w

    def linear(a, w, b=None, residual=None, **kw):
        out = F.linear(a, w, b)
        return out if residual is None else out + residual

    def linear_pair_chain(a, wq, q_term, wv, bv):
        zq = F.linear(a, wq)
        return (zq if q_term is None else zq + q_term), F.linear(a, wv, bv)

    def tsa_fused_forward(value, offs, logits, ref_2d, bh, bw, heads, points, shared_queue=False, order=None,
                          value_rows=None):
        B, n = offs.shape[:2]
        assert (bh, bw) == (bev_h, bev_w) and value.shape[1] == nq and (value_rows in (None, nq))
        assert tuple(ref_2d.shape) == (2, n, 1, 2) and ref_2d.is_contiguous()
        assert order is not None and sorted(order.tolist()) == list(range(n))
        assert offs.stride(0) == n * offs.stride(1) and logits.stride(0) == n * logits.stride(1)
        # the kernel samples the value MAP around the query's reference point: here, the pixel under it and its right
        # / lower neighbours (clamped) — a wrong band of reference points or a banded value map changes the result
        x = (ref_2d[1, :, 0, 0] * bw).floor().long().clamp(0, bw - 1)
        y = (ref_2d[1, :, 0, 1] * bh).floor().long().clamp(0, bh - 1)
        v = value.reshape(nq, -1)
        pix = y * bw + x
        g = v[pix] + 0.5 * v[(y * bw + (x + 1).clamp(max=bw - 1))] + 0.25 * v[((y + 1).clamp(max=bh - 1) * bw + x)]
        log.append(('T', n))
        return (g * torch.tanh(logits.mean(-1, keepdim=True)) + offs.mean(-1, keepdim=True)).view(1, n, -1)

    def linear_ln_chain(a, residual, w1, b1, ln, w2, b2, act2=None):
        y = ln(F.linear(a, w1, b1) + residual)
        log.append(('A', a.shape[1]))
        return y, F.linear(y, w2, b2)

    def sca_fused_forward(value, shapes, lsi, offs, logits, ref_cam, vis_bits, heads, levels, points, order=None,
                          stats=None, value_layout="rows", value_scale=None):
        NC, B, n, Z, _ = ref_cam.shape
        assert ref_cam.is_contiguous() and tuple(vis_bits.shape) == (1, n) and vis_bits.is_contiguous()
        assert tuple(offs.shape[:2]) == (1, n) and offs.stride(0) == n * offs.stride(1)
        assert order is not None and sorted(order.tolist()) == list(range(n))
        # per query: its own offsets / logits, its own anchor points in every camera, its own visibility word, and
        # the (un-banded) camera planes
        cam = ref_cam[:, 0].reshape(NC, n, -1).sum((0, 2))                                   # (n,)
        c = value.shape[-1] * value.shape[-2]
        o = offs[0, :, :c] * cam[:, None] + logits[0, :, :c] + vis_bits[0].float()[:, None] + value.float().mean()
        assert value_scale is not None and value_scale.numel() == 1          # fp16 planes travel with their range scale
        o = o + value_scale.float()                                          # the LAYER's own scale reaches its gather
        log.append(('S', n))
        return o.view(1, n, c)

    def encoder_ffn_chain(a, residual, wo, bo, ln1, w1, b1, w2, b2, ln2, tail=None, out=None):
        x2 = ln1(F.linear(a, wo, bo) + residual)
        y = ln2(F.linear(F.relu(F.linear(x2, w1, b1)), w2, b2) + x2)
        zq = zv = None
        if tail is not None:
            wq, q_term, wv, bv = tail
            zq = F.linear(y, wq)
            if q_term is not None:
                assert q_term.shape[1] == a.shape[1]
                zq = zq + q_term
            zv = F.linear(y, wv, bv)
        log.append(('B', a.shape[1]))
        if out is not None:
            out[0].copy_(y)
            if tail is not None:
                out[1].copy_(zq)
                out[2].copy_(zv)
            return out
        return y, zq, zv

    def encoder_bands_forward(q0, zq0, zv0, layers, bands, spatial_shapes, level_start_index, vis_bits, bh, bw,
                              num_levels, num_points, tsa_points, flags=0):
        """csrc/encoder_bands.hip's launch sequence, on the stand-ins above (same pointer arithmetic, as slices)."""
        assert (bh, bw) == (bev_h, bev_w) and [b['m0'] for b in bands] == [0] + [b['m1'] for b in bands[:-1]]
        assert bands[-1]['m1'] == nq and len({id(b['stream']) for b in bands}) == len(bands)
        n_lin = 8 * num_levels * num_points * 3
        tsa_noff = 8 * 2 * tsa_points * 2
        q_prev, zq, zv = q0, zq0, zv0
        for l, y in enumerate(layers):
            assert (y.get('tail') is not None) == (l + 1 < len(layers))
            w1, b1, ln, w2, b2 = y['a']
            wo, bo, ln1, f1, fb1, f2, fb2, ln2 = y['b']
            for b in bands:
                m0, m1 = b['m0'], b['m1']
                n = m1 - m0
                assert tuple(b['ref_cam'].shape[:3]) == (6, 1, n) and b['lin'].shape == (1, n, n_lin)
                lin = zq[:, m0:m1]
                b['attn'].copy_(tsa_fused_forward(zv.view(1, nq, 8, -1), lin[0, :, :tsa_noff].unsqueeze(0),
                                                  lin[0, :, tsa_noff:].unsqueeze(0), b['ref_2d'], bh, bw, 8, tsa_points,
                                                  shared_queue=True, order=b['order'], value_rows=nq))
            for b in bands:
                x1, z = linear_ln_chain(b['attn'], q_prev[:, b['m0']:b['m1']], w1, b1, ln, w2, b2)
                b['x1'].copy_(x1)
                b['lin'].copy_(z)
            for b in bands:
                b['slots'].copy_(sca_fused_forward(y['plane'], spatial_shapes, level_start_index,
                                                   b['lin'][..., :n_lin // 3 * 2], b['lin'][..., n_lin // 3 * 2:],
                                                   b['ref_cam'], vis_bits[:, b['m0']:b['m1']], 8, num_levels, num_points,
                                                   order=b['order'], stats=y['stats'], value_layout="pairs",
                                                   value_scale=y['plane_scale']))
            for b in bands:
                m0, m1 = b['m0'], b['m1']
                tail = y.get('tail')
                if tail is not None:
                    tail = (tail[0], None if tail[1] is None else tail[1][:, m0:m1], tail[2], tail[3])
                encoder_ffn_chain(b['slots'], b['x1'], wo, bo, ln1, f1, fb1, f2, fb2, ln2, tail=tail,
                                  out=(y['out'][:, m0:m1], None if tail is None else y['zq'][:, m0:m1],
                                       None if tail is None else y['zv'][:, m0:m1]))
            q_prev = y['out']
            if l + 1 < len(layers):
                zq, zv = y['zq'], y['zv']

    for name, fn in dict(linear=linear, linear_pair_chain=linear_pair_chain, tsa_fused_forward=tsa_fused_forward,
                         encoder_bands_forward=encoder_bands_forward,
                         linear_ln_chain=linear_ln_chain, sca_fused_forward=sca_fused_forward,
                         encoder_ffn_chain=encoder_ffn_chain).items():
        monkeypatch.setattr(ext, name, fn)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None)


class _Planes:
    """LazyFeatures stand-in: one fp16 'plane' per value_proj module."""

    def __init__(self, encoder):
        g = torch.Generator().manual_seed(5)
        self.planes = {id(l.attentions[1].deformable_attention.value_proj):
                       torch.randn(6, 64, 256, generator=g).half() for l in encoder.layers}
        self.scales = {k: torch.tensor([2.0 ** (i - 3)]) for i, k in enumerate(self.planes)}     # one per layer, all different
        self.asked = []

    def value_scale(self, vp):
        return self.scales[id(vp)]

    def project_on(self, vp, streams):
        self.asked.append(id(vp))
        return self.planes[id(vp)]

    def take_on(self, vp, streams):
        return self.project_on(vp, streams), None


def _reference_walk(encoder, q, planes, bev_pos, ref_2d, bev_h, bev_w, ref_cam, vis_bits, order):
    """forward_chain's sequence on all rows, on the same stand-ins."""
    layers = encoder.layers
    tsa0 = layers[0].attentions[0]
    zq, zv = ext.linear_pair_chain(q, *tsa0.chain_tail(bev_pos))
    outs = []
    for lid, layer in enumerate(layers):
        tsa, sca = layer.attentions
        n_off = tsa.sampling_offsets.out_features
        attn = ext.tsa_fused_forward(zv.view(1, -1, tsa.num_heads, 256 // tsa.num_heads), zq[..., :n_off], zq[..., n_off:],
                                     ref_2d, bev_h, bev_w, tsa.num_heads, tsa.num_points, shared_queue=True, order=order)
        wq, bq = sca.query_linear_operands()
        x1, lin = ext.linear_ln_chain(attn, q, tsa.output_proj.weight, tsa.output_proj.bias, layer.norms[0], wq, bq)
        slots = sca.gather_projected(lin, planes.planes[id(sca.deformable_attention.value_proj)], ref_cam, vis_bits,
                                     None, None, order, None,
                                     value_scale=planes.value_scale(sca.deformable_attention.value_proj))
        ffn = layer.ffns[0]
        tail = layers[lid + 1].attentions[0].chain_tail(bev_pos) if lid + 1 < len(layers) else None
        q, zq, zv = ext.encoder_ffn_chain(slots, x1, sca.output_proj.weight, sca.output_proj.bias, layer.norms[1],
                                          ffn.layers[0][0].weight, ffn.layers[0][0].bias, ffn.layers[1].weight,
                                          ffn.layers[1].bias, layer.norms[2], tail=tail)
        outs.append(q)
    return outs


@pytest.mark.parametrize("k,native", [(2, False), (3, False), (1, True), (2, True), (3, True)])
def test_row_pipeline_plumbing_equals_the_unbanded_walk(monkeypatch, k, native):
    monkeypatch.setattr(enc_mod, "_ROW_PIPELINE_NATIVE", native)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'occ_base_200x200x16.py'))
    torch.manual_seed(0)
    model = build_model(cfg.model)
    model.init_weights()
    encoder = model.pts_bbox_head.transformer.encoder
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in encoder.named_parameters():
            if n.endswith("sampling_offsets.weight") or n.endswith("attention_weights.weight"):
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
    bev_h, bev_w = 24, 20                       # three tile rows: bands of 16 + 8 (k = 2) or 8 + 8 + 8 (k = 3) rows
    nq = bev_h * bev_w
    log = []
    _fake_ops(monkeypatch, bev_h, bev_w, log)
    q = torch.randn(1, nq, 256, generator=g)
    bev_pos = torch.randn(1, nq, 256, generator=g)
    _, ref_2d, hybrid, _, _ = encoder._reference_grids(bev_h, bev_w, 1, q.device, q.dtype)
    ref_cam = torch.rand(6, 1, nq, 4, 2, generator=g)
    vis_bits = torch.randint(0, 64, (1, nq), generator=g, dtype=torch.int32)
    order = torch.from_numpy(bev_tile_order(bev_h, bev_w))
    planes = _Planes(encoder)
    with torch.no_grad():
        want = _reference_walk(encoder, q, planes, bev_pos, hybrid, bev_h, bev_w, ref_cam, vis_bits, order)
        log.clear()
        run = lambda: encoder._forward_row_pipeline(k, q.permute(1, 0, 2).permute(1, 0, 2), planes, bev_pos, hybrid, bev_h,
                                                    bev_w, ref_cam, None, None, vis_bits, None)
        assert run() is None, "the first call after a weight / cache change must take the standard path"
        assert not log
        got = run()
    assert got is not None and len(got) == len(want) == len(encoder.layers)
    for a, b in zip(got, want):
        assert a.shape == b.shape == (1, nq, 256)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    # every layer asked for its own plane once; every stage ran once per band on the band's rows
    assert planes.asked == [id(l.attentions[1].deformable_attention.value_proj) for l in encoder.layers]
    sizes = [m1 - m0 for m0, m1, _ in row_bands(bev_h, bev_w, k)]
    assert len(sizes) == k
    per_layer = [(s, n) for s in 'TASB' for n in sizes]
    assert log == per_layer * len(encoder.layers)
    assert log.count(('B', sizes[0])) == len(encoder.layers) * sizes.count(sizes[0])
    # a weight update sends the next call down the standard path again
    with torch.no_grad():
        next(encoder.parameters()).add_(0.0)
        assert run() is None


def test_row_pipeline_switch_is_off_by_default():
    assert os.environ.get("OCC_ENCODER_ROW_PIPELINE") or enc_mod._ROW_PIPELINE == 0
