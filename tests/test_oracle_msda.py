"""not-gpu: the three restatements of the deformable-attention arithmetic agree — the grid_sample
form (mmcv's CPU function), the scalar fp64 re-derivation of the CUDA kernel
(Appendix B.2) and the plain-C port (oracle/msda_ref.c)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import msda

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(seed, B=2, M=3, D=8, shapes=((5, 7), (3, 4), (2, 2)), Lq=9, P=4, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.tensor(shapes)
    start = torch.cat([shapes_t.new_zeros(1), shapes_t.prod(1).cumsum(0)[:-1]])
    S = int(shapes_t.prod(1).sum())
    v = torch.randn(B, S, M, D, generator=g, dtype=dtype)
    loc = torch.rand(B, Lq, M, len(shapes), P, 2, generator=g, dtype=dtype) * 1.6 - 0.3
    H, W = shapes[0]
    loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0])
    loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0])
    if P > 2:
        loc[0, 0, 0, 0, 2] = torch.tensor([0.5 / W, 0.5 / H])
    loc[0, 1] = 1e6
    loc[0, 2] = -1e6
    aw = torch.rand(B, Lq, M, len(shapes), P, generator=g, dtype=dtype)
    return v, shapes_t, start, loc, aw


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_grid_sample_form_equals_scalar_kernel_arithmetic(seed):
    v, shapes, start, loc, aw = _case(seed)
    a = msda.multi_scale_deformable_attn_pytorch(v, shapes, loc, aw).numpy()
    b, n_in = msda.msda_scalar_f64(v.numpy(), shapes.numpy(), start.numpy(), loc.numpy(), aw.numpy())
    assert np.abs(a - b).max() < 1e-12
    assert n_in == msda.count_inbounds_corners(shapes, loc)


def test_empty_query_set_and_single_pixel_map():
    v, shapes, start, loc, aw = _case(3, B=1, M=2, D=4, shapes=((1, 1),), Lq=5, P=2)
    a = msda.multi_scale_deformable_attn_pytorch(v, shapes, loc, aw)
    b, _ = msda.msda_scalar_f64(v.numpy(), shapes.numpy(), start.numpy(), loc.numpy(), aw.numpy())
    assert np.abs(a.numpy() - b).max() < 1e-12
    out = msda.multi_scale_deformable_attn_pytorch(v, shapes, loc[:, :0], aw[:, :0])
    assert out.shape == (1, 0, 8)


def test_c_port_matches():
    so = os.path.join(ROOT, "oracle", "_build", "libmsda_ref.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(so)
    v, shapes, start, loc, aw = _case(4, B=2, M=8, D=32, shapes=((12, 20), (6, 10), (3, 5), (2, 3)),
                                      Lq=50, P=8)
    ref = msda.multi_scale_deformable_attn_pytorch(v, shapes, loc, aw).float()
    vf, lf, af = v.float().contiguous(), loc.float().contiguous(), aw.float().contiguous()
    out = torch.empty(ref.shape)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.msda_forward_ref_f32(p(vf), p(shapes), p(start), p(lf), p(af), p(out), 2, vf.shape[1], 8, 32,
                             4, 50, 8)
    assert float((out - ref).abs().max()) < 2e-5
