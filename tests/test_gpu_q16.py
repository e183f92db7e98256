"""-m gpu: the q16 value rows of the SCA gather (round 6) — block floating point, 16 bits per element like
the fp16 rows: 8 int16 mantissas per 16-byte piece under one 4-bit exponent (csrc/common.h).  The reference keeps these rows
in fp32 (`spatial_cross_attention.py:75,387-390`).
 * both HIP encoders — occ_sca_rows_encode_q16 and the value projection's q16 epilogue — against the numpy restatement
   (tests/q16_ref.py), bit for bit;
 * the gather over q16 rows against the fp32-row gather over the DECODED values (same sampling arithmetic, fp32 accumulation):
   equal to summation order;
 * end-to-end accuracy against the CPU oracle at the base geometry's feature statistics: tests/test_gpu_value_range.py
   (scales 1 .. 1e7) and tests/test_gpu_modules.py (every kernel shape)."""
import numpy as np
import pytest
import torch

from occnet_amd import ext
from tests import q16_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("BN,S,C,amp", [(2, 37, 256, 1.0), (6, 300, 256, 3e4), (1, 8, 64, 1e-5), (3, 1001, 256, 50.0)])
def test_rows_encode_kernel_equals_the_restatement(BN, S, C, amp):
    g = torch.Generator().manual_seed(S + C)
    v = torch.randn(BN, S, C, generator=g) * amp
    v[:, ::5, 7] *= 40                                              # outliers set their piece's exponent
    v[0, 0, :8] = 0.0                                               # an all-zero piece
    enc, s = ext.q16_range_scaled(v.cuda())
    assert 2.0 ** 14 <= float(v.abs().max()) * float(s) <= 2.0 ** 15
    assert enc.dtype == torch.int16 and tuple(enc.shape) == (BN, S + (S & 1), C)
    want = q16_ref.encode(v.numpy(), float(s), group=8)             # row order; the rows kernel: one exponent per piece
    want = q16_ref.pair_layout(want.reshape(BN, S, C // 32, 32)).reshape(BN, -1, C)
    got = enc.cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())
    # non-finite inputs have no q16 image: they encode to unspecified FINITE mantissas, their neighbours are unaffected
    v2 = v.clone()
    v2[0, 1, 40] = float('inf'); v2[0, 2, C - 7] = float('nan')
    enc2 = ext.sca_rows_encode_q16(v2.cuda(), s).cpu().numpy()
    diff = (enc2 != got).reshape(BN, -1, C // 32, 2, 32)             # [pair][head][pix & 1][32]
    assert diff.sum() > 0 and diff[1:].sum() == 0 and diff[0, 2:].sum() == 0


@pytest.mark.parametrize("amp", [1.0, 3e4])
def test_value_projection_q16_epilogue_equals_encode_of_the_fp32_projection(amp):
    """occ_value_proj_bf16_planes(out_f16 = 2): the resident kernel's q16 epilogue == encode(scale * its own fp32 output)."""
    g = torch.Generator().manual_seed(11)
    hws, G, K, N, P = [(12, 20), (16, 10)], 2, 256, 256, 3
    a_list = [(torch.randn(G * h * w, K, generator=g) * amp).to(torch.bfloat16).cuda() for h, w in hws]
    ws = [((torch.rand(N, K, generator=g) * 2 - 1) * 0.1).cuda() for _ in range(P)]
    gbs = [torch.randn(len(hws), G, N, generator=g).cuda() for _ in range(P)]
    rpg = [h * w for h, w in hws]
    starts = [0, rpg[0]]
    total = sum(rpg)
    l1 = [float(w.abs().sum(1).max()) for w in ws]
    bm = [float(gb.abs().max()) for gb in gbs]
    t = ext.value_range_scale(a_list, l1, bm)
    f32 = torch.empty(P, G * total, N, device='cuda')
    ext.value_proj_bf16_planes(a_list, ws, gbs, f32, rows_per_group=rpg, out_group_rows=total, out_row0=starts)
    q = torch.empty(P, G * total, N, dtype=torch.int16, device='cuda')
    ext.value_proj_bf16_planes(a_list, ws, gbs, q, rows_per_group=rpg, out_group_rows=total, out_row0=starts, out_scale=t[:P])
    torch.cuda.synchronize()
    for p in range(P):
        s = float(t[p])
        want = q16_ref.encode(f32[p].cpu().numpy().reshape(G, total, N), s, group=32)      # one exponent per 64-byte head row
        want = q16_ref.pair_layout(want.reshape(G, total, N // 32, 32)).reshape(G * total, N)
        got = q[p].cpu().numpy()
        assert np.array_equal(got, want), (p, int((got != want).sum()))
        assert int(np.abs(got.astype(np.int64)).max()) >= 1 << 10          # the planes use their range


def test_gather_over_q16_rows_equals_the_fp32_gather_of_the_decoded_values():
    gen = torch.Generator().manual_seed(5)
    B, NC, M, D, L, P, Z = 1, 6, 8, 32, 4, 8, 8
    shapes = torch.tensor([(29, 50), (15, 25), (8, 13), (4, 7)])
    S = int(shapes.prod(1).sum())                                    # 1957: odd -> a pad row
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    Nq = 900
    value = torch.randn(B * NC, S, M * D, generator=gen) * 37.0
    value[:, ::11, 5] *= 30
    offs = torch.randn(B, Nq, M * L * P * 2, generator=gen) * 3
    logits = torch.randn(B, Nq, M * L * P, generator=gen)
    ref_cam = torch.rand(NC, B, Nq, Z, 2, generator=gen) * 1.2 - 0.1
    vis = torch.randint(1, 64, (B, Nq), generator=gen, dtype=torch.int32)
    args = (shapes.cuda(), start.cuda(), offs.cuda(), logits.cuda(), ref_cam.cuda(), vis.cuda(), M, L, P)
    enc, s = ext.q16_range_scaled(value.cuda())
    got = ext.sca_fused_forward(enc.view(B * NC, -1, M, D), *args, value_layout="pairs", value_scale=s)
    # decode on the host (row order), gather those exact values with the fp32-row kernel
    rows = ext.sca_unpair_layout(enc.view(B * NC, -1, M, D), S=S).cpu().numpy()
    # (the exponent bits sit in elements 0, 1 of every 8: decode works on row-ordered pieces)
    dec = torch.from_numpy(q16_ref.decode(rows.reshape(B * NC, S, M * D), float(s)).astype(np.float32))
    want = ext.sca_fused_forward(dec.view(B * NC, S, M, D).cuda(), *args)
    d = float((got - want).abs().max())
    scale = float(want.abs().max())
    print(f"q16 gather vs fp32 gather of the decoded rows: max diff {d:.3e} (scale {scale:.1f}); "
          f"vs the original values: {float((got.cpu() - ext.sca_fused_forward(value.view(B * NC, S, M, D).cuda(), *args).cpu()).abs().max()):.3e}")
    assert d <= 2e-6 * scale
    with pytest.raises(ext.OccAmdError):
        ext.sca_fused_forward(enc.view(B * NC, -1, M, D), *args, value_scale=s)           # q16 rows exist as pairs only
