#!/usr/bin/env python
"""Time the row-local Linear chains of one encoder layer (M = 40 000): the chain kernels (csrc/linear_chain_x3.hip)
against the one-launch-per-Linear sequence they replace, same operands, ABAB in one process.
usage: python tools_dev/chain_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnet_amd import ext   # noqa: E402

dev = torch.device("cuda", 0)
M = int(os.environ.get("LIN_M", "40000"))
g = torch.Generator().manual_seed(0)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
s256, s512 = 256 ** -0.5, 512 ** -0.5
a, res = R(M, 256), R(M, 256)
wo, bo = R(256, 256, sc=s256), R(256, sc=0.1)
ln1 = (torch.rand(256, generator=g).to(dev) + 0.5, R(256, sc=0.1), 1e-5)
ln2 = (torch.rand(256, generator=g).to(dev) + 0.5, R(256, sc=0.1), 1e-5)
wq, bq = R(768, 256, sc=s256), R(768, sc=0.1)
w1, b1 = R(512, 256, sc=s256), R(512, sc=0.1)
w2, b2 = R(256, 512, sc=s512), R(256, sc=0.1)
wt, qt = R(192, 256, sc=s256), R(M, 192)
wv, bv = R(256, 256, sc=s256), R(256, sc=0.1)


def timed(fn, n=30):
    for _ in range(4):
        o = fn()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2] * 1e3, o


def sep_a():
    y = ext.linear(a, wo, bo, residual=res, ln=ln1)
    return y, ext.linear(y, wq, bq)


def sep_b(tail=True):
    x2 = ext.linear(a, wo, bo, residual=res, ln=ln1)
    h = ext.linear(x2, w1, b1, act='relu')
    y = ext.linear(h, w2, b2, residual=x2, ln=ln2)
    if not tail:
        return y, None, None
    return y, ext.linear(y, wt, None, residual=qt), ext.linear(y, wv, bv)


def md(x, y):
    return float((x - y).abs().max())


# block-count sweep: 16 384 rows = 256 blocks (one per CU, alone), 32 768 = 512 (two per CU, one full round),
# 40 000 = 625 (1.22 rounds: the encoder's shape)
if os.environ.get("CHAIN_SWEEP", "1") == "1":
    for m in (16384, 32768, 40000, 65536):
        aa, rr, qq = a[:m] if m <= M else R(m, 256), res[:m] if m <= M else R(m, 256), qt[:m] if m <= M else R(m, 192)
        ta, _ = timed(lambda: ext.linear_ln_chain(aa, rr, wo, bo, ln1, wq, bq))
        tb, _ = timed(lambda: ext.encoder_ffn_chain(aa, rr, wo, bo, ln1, w1, b1, w2, b2, ln2, tail=(wt, qq, wv, bv)))
        print(f"rows {m:6d} ({(m + 63) // 64:5d} blocks): program A {ta:7.1f} us ({ta / m * 1e3:6.3f} ns/row)   program B {tb:7.1f} us "
              f"({tb / m * 1e3:6.3f} ns/row)", flush=True)

# memory floor of program A's traffic with stock streaming kernels: y = a + res (read 2 x 41 MB, write 41 MB), z = three
# copies of y (read 41 MB, write 123 MB) — 246 MB like the chain (which reads y from LDS, not from memory)
if os.environ.get("CHAIN_FLOOR", "1") == "1":
    zbuf = torch.empty(M, 768, device=dev)
    def floor_a():
        y = torch.add(a, res)
        zbuf.view(M, 3, 256).copy_(y.view(M, 1, 256).expand(M, 3, 256))
        return y
    t_f, _ = timed(floor_a)
    t_add, _ = timed(lambda: torch.add(a, res))
    print(f"memory floor, program A traffic (torch add + 3-way copy): {t_f:7.1f} us   (add alone {t_add:6.1f} us = "
          f"{3 * M * 1024 / t_add / 1e6:5.2f} TB/s)", flush=True)

for rnd in range(int(os.environ.get("CHAIN_ROUNDS", "2"))):
    t_sa, o_sa = timed(sep_a)
    t_ca, o_ca = timed(lambda: ext.linear_ln_chain(a, res, wo, bo, ln1, wq, bq))
    t_sb, o_sb = timed(sep_b)
    t_cb, o_cb = timed(lambda: ext.encoder_ffn_chain(a, res, wo, bo, ln1, w1, b1, w2, b2, ln2, tail=(wt, qt, wv, bv)))
    t_sb0, o_sb0 = timed(lambda: sep_b(False))
    t_cb0, o_cb0 = timed(lambda: ext.encoder_ffn_chain(a, res, wo, bo, ln1, w1, b1, w2, b2, ln2))
    fa = 2.0 * M * 256 * (256 + 768) * 3
    fb = 2.0 * M * 256 * (256 + 512 + 512 + 192 + 256) * 3
    print(f"round {rnd}: program A  separate (2 launches) {t_sa:7.1f} us   chain {t_ca:7.1f} us ({fa / t_ca / 1e6:6.0f} TFLOP/s bf16 issued)"
          f"   max|dy| {md(o_sa[0], o_ca[0]):.1e} max|dz| {md(o_sa[1], o_ca[1]):.1e}", flush=True)
    print(f"round {rnd}: program B  separate (5 launches) {t_sb:7.1f} us   chain {t_cb:7.1f} us ({fb / t_cb / 1e6:6.0f} TFLOP/s bf16 issued)"
          f"   max|dy| {md(o_sb[0], o_cb[0]):.1e} max|dzq| {md(o_sb[1], o_cb[1]):.1e} max|dzv| {md(o_sb[2], o_cb[2]):.1e}", flush=True)
    print(f"round {rnd}: program B without tail  separate (3 launches) {t_sb0:7.1f} us   chain {t_cb0:7.1f} us   max|dy| {md(o_sb0[0], o_cb0[0]):.1e}",
          flush=True)

# per-block phase timeline (CHAIN_TRACE=1): the stamped build of the kernels writes 16 wall-clock stamps per block
# (100 MHz); reported relative to the launch's first stamp, as medians over the blocks of the first resident round
# (two per CU) and over the tail round
if os.environ.get("CHAIN_TRACE", "0") == "1":
    import numpy as np
    import tempfile
    NAMES = {"A": [(0, 1, "rows in, tile built"), (1, 2, "S1 k-loop"), (2, 4, "LayerNorm (2 barriers)"), (4, 5, "x1 stores issued"),
                   (5, 6, "tile rebuilt"), (6, 3, "block barrier"), (3, 11, "pass 0 k-loop"),
                   (11, 12, "stores 0 + pass 1 k-loop"), (12, 13, "stores 1 + pass 2 k-loop"), (13, 15, "stores 2")],
             "B": [(0, 1, "rows in, tile built"), (1, 2, "S1 k-loop"), (2, 3, "LN + tile"), (3, 4, "S2a k-loop (+ x2 stores)"),
                   (4, 5, "S2b k-loop"), (5, 6, "ha -> tile, x2 reload"), (6, 7, "S3a k-loop"), (7, 8, "hb -> tile"),
                   (8, 9, "S3b k-loop"), (9, 16, "LayerNorm 2 (2 barriers)"), (16, 17, "tile rebuilt"), (17, 10, "block barrier"),
                   (10, 11, "term rows + pass 0 k-loop (+ x3 stores)"),
                   (11, 12, "stores 0 + pass 1 k-loop"), (12, 15, "stores 1")]}
    prefix = os.path.join(tempfile.mkdtemp(), "chain")
    for m in (int(x) for x in os.environ.get("CHAIN_TRACE_ROWS", "32768,40000").split(",")):
        aa, rr, qq = a[:m], res[:m], qt[:m]
        for prog, fn in (("A", lambda: ext.linear_ln_chain(aa, rr, wo, bo, ln1, wq, bq)),
                         ("B", lambda: ext.encoder_ffn_chain(aa, rr, wo, bo, ln1, w1, b1, w2, b2, ln2, tail=(wt, qq, wv, bv)))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            path = f"{prefix}.{prog}.bin"
            if os.path.exists(path):
                os.remove(path)
            os.environ["OCC_CHAIN_TRACE"] = prefix
            fn()
            torch.cuda.synchronize()
            os.environ.pop("OCC_CHAIN_TRACE")
            sw = np.fromfile(path, dtype=np.int64).reshape(-1, 4, 24).astype(np.float64)
            sw = (sw - sw[:, :, 0].min()) * 0.01         # us since the first wave started; [block][wave][stamp]
            st = sw[:, 0, :]                             # the timeline below follows wave 0
            nblk = st.shape[0]
            first = st[:min(nblk, 512)]
            print(f"--- program {prog}, {m} rows, {nblk} blocks: launch span {st[:, 15].max():.1f} us; first-round blocks start "
                  f"{np.median(first[:, 0]):.1f} us (max {first[:, 0].max():.1f}), end {np.median(first[:, 15]):.1f} us "
                  f"(min {first[:, 15].min():.1f}, max {first[:, 15].max():.1f})")
            groups = [("first round", first)] + ([("tail round", st[512:])] if nblk > 512 else [])
            for gname, gs in groups:
                if gname == "tail round":
                    print(f"    tail round: {gs.shape[0]} blocks start {np.median(gs[:, 0]):.1f} us (min {gs[:, 0].min():.1f}, max {gs[:, 0].max():.1f}), "
                          f"end {np.median(gs[:, 15]):.1f} us (max {gs[:, 15].max():.1f})")
                gw = sw[:min(nblk, 512)] if gname == "first round" else sw[512:]
                for i0, i1, label in NAMES[prog]:
                    d = gs[:, i1] - gs[:, i0]
                    skew = gw[:, :, i1].max(axis=1) - gw[:, :, i1].min(axis=1)       # arrival spread of the block's 4 waves
                    print(f"    {gname:11s} {label:28s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}"
                          f"   wave skew at its end: median {np.median(skew):5.2f}  p90 {np.percentile(skew, 90):5.2f}")
